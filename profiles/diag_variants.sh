#!/bin/bash
# whole-frame comparison of library variants (pathtracer_amd/variant_<name>.so) against the per-lane grid walk of the
# default library, on one benchmark workload.  usage: profiles/diag_variants.sh <workload> name ...
w=$1; shift
PT_GRID_PER_LANE=1 timeout -k 10 300 python -u tests/diag_frames.py render $w /tmp/lane.npy || exit 1
for n in "$@"; do
	echo "== $n"
	PT_CORE_LIB=$PWD/pathtracer_amd/variant_$n.so timeout -k 10 200 python -u tests/diag_frames.py render $w /tmp/v.npy || exit 1
	timeout -k 10 300 python -u tests/diag_frames.py compare $w /tmp/v.npy /tmp/lane.npy | cut -c1-60 | head -5
done
