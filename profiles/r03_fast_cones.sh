#!/bin/bash
# The two grid frames at full size with the cooperative cast's plain variants (PT_FAST_CONES=1: without the direction test that
# reproduces the reference cone intersector's phantom hits) beside the default, one bench line each
mkdir -p gpurun_out
for w in glass-1080p-128spp-d32 dense100k-4k-256spp-d8; do
	timeout -k 10 300 python3 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu --no-secondary --workload $w > gpurun_out/r03_default_$w.json 2>gpurun_out/r03_fast_cones.err || { tail gpurun_out/r03_fast_cones.err; exit 1; }
	PT_FAST_CONES=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu --no-secondary --workload $w > gpurun_out/r03_fast_cones_$w.json 2>gpurun_out/r03_fast_cones.err || { tail gpurun_out/r03_fast_cones.err; exit 1; }
done
python3 - <<'PY'
import json
for w in ("glass-1080p-128spp-d32", "dense100k-4k-256spp-d8"):
    for tag in ("default", "fast_cones"):
        d = json.loads([l for l in open(f"gpurun_out/r03_{tag}_{w}.json") if l.startswith("{")][-1])
        print(f"{w:28s} {tag:11s} {d['value']:.0f} Mpaths/s  ms_per_step {d['ms_per_step']:.2f}  kernel_ms {d['roofline']['kernel_ms']:.2f}  frac {d['roofline']['frac']:.3f}")
PY
