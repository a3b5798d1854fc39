#!/bin/bash
# A/B timing of library variants on the benchmark workload.
# usage: profiles/ab.sh [name ...]   (names of pathtracer_amd/variant_<name>.so, default: all, in order)
names="$@"
[ -z "$names" ] && names=$(ls pathtracer_amd/variant_*.so | sed 's/.*variant_\(.*\)\.so/\1/')
for n in $names; do
	f=pathtracer_amd/variant_$n.so
	PT_CORE_LIB=$PWD/$f timeout -k 10 200 python -u bench.py --steps ${STEPS:-200} --warmup ${WARMUP:-60} --no-cpu ${BENCH_ARGS:-} 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-14s %.0f Mpaths/s  ms_per_step %.3f  kernel_ms %.3f [%.3f..%.3f]' % ('$n', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min'], d['roofline']['kernel_ms_max']))
" || exit 1
done
