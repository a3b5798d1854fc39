#!/bin/bash
# quick GPU check used while tuning: parity tests, then the benchmark line without the CPU leg
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -4 && \
timeout -k 10 300 python -u bench.py --steps 200 --warmup 60 --no-cpu "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.0f Mpaths/s  ms_per_step %.3f  kernel_ms %.3f [%.3f..%.3f]' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min'], d['roofline']['kernel_ms_max']))
"
