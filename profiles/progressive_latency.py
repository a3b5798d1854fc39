"""Progressive mode on the demo scene at 1080p with the 64-spp sample table: frame f traces sample
(f mod count) of every pixel and adds it into the accumulation buffer.  Prints the per-frame latency
(host clock around pt_set_frame + pt_trace with synchronisation) over 256 frames as one JSON line.
usage: python profiles/progressive_latency.py [frames]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, scenes  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
world, objs = scenes.default_scene()
st = api.default_settings(1920, 1080, 64, 8, seed=12345, frame_index=0)
r = api.Renderer(0)
r.prepare(world, objs, st)
for f in range(32):                      # warm-up, not timed
    r.set_frame(f)
    r.trace()
lat = []
for f in range(frames):
    t0 = time.perf_counter()
    r.set_frame(f)
    r.trace()                            # synchronises
    lat.append((time.perf_counter() - t0) * 1e3)
img = r.download()
lat = np.array(lat)
print(json.dumps({"workload": "progressive demo 1920x1080, one sample per pixel per frame", "frames": frames,
                  "median_ms": float(np.median(lat)), "p99_ms": float(np.percentile(lat, 99)), "max_ms": float(lat.max()),
                  "kernel_ms": float(r.stats().dominant_kernel_ms), "image_mean": float(img.mean())}))
r.close()
