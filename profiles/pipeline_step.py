#!/usr/bin/env python3
"""Host-side cost of one pipelined multi-GPU step, rehearsed on one GPU: this process owns partition 0 of
`parts` and runs FramePipeline.step (trace, events, placement copy instead of the gather, assemble).
Prints wall time per step beside the trace kernel's duration: where the first exceeds the second, the step is
bound by the host's launch work, not by the GPU.  usage: python profiles/pipeline_step.py [parts] [ranges]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import api  # noqa: E402
from pathtracer_amd import dist as ptdist  # noqa: E402

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranges = int(sys.argv[2]) if len(sys.argv) > 2 else 0
world, objs, kw = bench.workload("demo-1080p-64spp-d8")
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), part_index=0, part_count=parts,
                          sample_ranges=ranges, **kw)
torch.cuda.set_device(0)
r = api.Renderer(0)
r.prepare(world, objs, st)
pipe = ptdist.FramePipeline([r], [0], parts, st.width, st.height, 0, 1)
for _ in range(50):
    pipe.step()
torch.cuda.synchronize()
r.sync()
steps = 400
t0 = time.perf_counter()
for _ in range(steps):
    pipe.step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
r.sync()
s = r.stats()
print(json.dumps({"parts": parts, "ranges": ranges, "wall_us_per_step": dt / steps * 1e6, "host_issue_us_per_step": t_issue / steps * 1e6,
                  "trace_kernel_us": s.dominant_kernel_ms * 1e3, "trace_and_resolve_us": s.trace_ms * 1e3}))
r.close()
