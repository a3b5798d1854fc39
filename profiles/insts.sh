#!/bin/bash
# Dynamic instruction counts of the trace kernel by type, for one or more library variants.
# BENCH_ARGS="--workload <name>" selects another workload than the demo frame.
# usage: profiles/insts.sh <variant name | ""> ...     ("" = the product library)
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
for n in "$@"; do
	[ -n "$n" ] && export PT_CORE_LIB=$ROOT/pathtracer_amd/variant_$n.so || unset PT_CORE_LIB
	OUT=$ROOT/gpurun_out/insts_${n:-product}
	rm -rf "$OUT"
	timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-secondary --no-prewarm ${BENCH_ARGS:-} > "$OUT.log" 2>&1 || { echo "run failed"; tail -5 "$OUT.log"; exit 1; }
	python3 - "$OUT" "${n:-product}" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pt_k_trace" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
total = sum(v for k, v in m.items() if k.startswith("SQ_INSTS") and k != "SQ_INSTS_BRANCH")
print(sys.argv[2], " ".join(f"{k[3:]}={v:.4g}" for k, v in sorted(m.items())), f"all={total:.4g}")
PY
done
