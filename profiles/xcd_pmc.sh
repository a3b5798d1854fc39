#!/bin/bash
# L2 traffic of the 100 000-entity frame with and without the per-XCD dealing order (PT_TILE_ORDER_XCD): separate counter passes
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
for x in 0 1; do
	if [ $x = 1 ]; then export PT_TILE_ORDER_XCD=1; else unset PT_TILE_ORDER_XCD; fi
	for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
		name=$(echo $grp | cut -d' ' -f1)
		OUT=$ROOT/gpurun_out/xcd_pmc/order${x}_$name
		mkdir -p $OUT
		timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --no-prewarm --workload dense100k-1920x1080-128spp-d8 > $OUT.log 2>&1 || { echo "pass failed"; tail -5 $OUT.log; exit 1; }
		python3 - $OUT $x <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pt_k_trace" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("order", sys.argv[2], k, "launches", len(v), "mean per launch %.6g" % (sum(v) / len(v)))
PY
	done
done
