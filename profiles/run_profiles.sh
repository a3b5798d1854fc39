#!/bin/bash
# Collects the rocprofv3 evidence for one benchmark configuration on the GPU box.
# usage: profiles/run_profiles.sh <tag> [bench args...]      (run from the repository root)
# Kernel trace and each counter group are separate runs, as the counter hardware requires.
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-secondary --no-prewarm $*"
# the kernel trace runs as long as the default bench (clocks settle over the first tens of launches)
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --steps ${TRACE_STEPS:-100} --warmup ${TRACE_WARMUP:-30} --no-cpu --no-secondary $* > "$OUT/trace.log" 2>&1 || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
	name=$(echo $grp | cut -d' ' -f1)
	timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- python3 $ARGS > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc run $name failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
echo "profiles written to $OUT"
find "$OUT" -name "*.csv" | head -40
