#!/usr/bin/env python3
"""Kernel time of ONE partition of the benchmark frame for part_count = 1, 2, 4, 8 on one GPU: what every
rank of an N-GPU run executes.  Ideal is t(1) / part_count; the gap is the frame's tail (too few work items
per resident wavefront).  usage: python profiles/partition_time.py [workload] [ranges ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pathtracer_amd import api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "demo-1080p-64spp-d8"
ranges_list = [int(a) for a in sys.argv[2:]] or [0]
world, objs, kw = bench.workload(name)
out = {"workload": name, "rows": []}
for ranges in ranges_list:
    base = None
    for parts in (1, 2, 4, 8):
        skw = dict(kw)
        st = api.default_settings(skw.pop("width"), skw.pop("height"), skw.pop("spp"), skw.pop("bounces"),
                                  part_index=0, part_count=parts, sample_ranges=ranges, **skw)
        r = api.Renderer(0)
        r.prepare(world, objs, st)
        for _ in range(10):
            r.trace(sync=False)
        r.sync()
        for _ in range(40):
            r.trace(sync=False)
        r.sync()
        s = r.stats()
        r.close()
        base = base or s.dominant_kernel_ms
        row = {"ranges": ranges, "part_count": parts, "kernel_ms": round(s.dominant_kernel_ms, 4), "frame_ms": round(s.trace_ms, 4),
               "speedup": round(base / s.dominant_kernel_ms, 2)}
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"partition_time_{name}.json"), "w"), indent=1)
