#!/bin/bash
# L1 (TCP) and address-unit (TA) counters of the trace kernel for one workload and library variant.
# usage: profiles/tcp_counters.sh <workload> <variant name | "">
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
W=$1; n=$2
[ -n "$n" ] && export PT_CORE_LIB=$ROOT/pathtracer_amd/variant_$n.so || unset PT_CORE_LIB
OUT=$ROOT/gpurun_out/tcp_${n:-product}_$W
rm -rf "$OUT"
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"; do
	i=$((i+1))
	timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$i" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --no-prewarm --workload $W > "$OUT.$i.log" 2>&1 || { echo "run failed"; tail -5 "$OUT.$i.log"; exit 1; }
done
python3 - "$OUT" "$W ${n:-product}" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pt_k_trace" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("[%s]" % sys.argv[2], " ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(acc.items())))
PY
