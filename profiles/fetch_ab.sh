#!/bin/bash
# FETCH_SIZE / wait counters of the trace kernel for one workload under different environment switches.
# usage: profiles/fetch_ab.sh <workload> "<env assignments or empty>" ...
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
W=$1; shift
i=0
for envs in "$@"; do
	i=$((i+1))
	OUT=$ROOT/gpurun_out/fetch_$i
	rm -rf "$OUT"
	for grp in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum"; do
		name=$(echo $grp | cut -d' ' -f1)
		( export $envs DUMMY=1; timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --no-prewarm --workload $W > "$OUT.$name.log" 2>&1 ) || { echo "run failed"; tail -5 "$OUT.$name.log"; exit 1; }
	done
	python3 - "$OUT" "$envs" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pt_k_trace" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("[%s]" % sys.argv[2], " ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(acc.items())))
PY
done
