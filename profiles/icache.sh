#!/bin/bash
# Instruction-cache counters of the benchmark frame for one or more library variants.
# BENCH_ARGS="--workload <name>" selects another workload than the demo frame.
# usage: profiles/icache.sh <variant name | ""> ...     ("" = the product library)
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
for n in "$@"; do
	[ -n "$n" ] && export PT_CORE_LIB=$ROOT/pathtracer_amd/variant_$n.so || unset PT_CORE_LIB
	OUT=$ROOT/gpurun_out/icache_${n:-product}
	rm -rf "$OUT"
	timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-secondary --no-prewarm ${BENCH_ARGS:-} > "$OUT.log" 2>&1 || { echo "run failed"; tail -5 "$OUT.log"; exit 1; }
	python3 - "$OUT" "${n:-product}" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pt_k_trace" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], " ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(acc.items())))
PY
done
