"""One frame through pt_render_start -> done, the way the application would call it (host pointers in,
host image out), with the set-up share.  usage: python profiles/single_shot.py [workload]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pathtracer_amd import api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "demo-1080p-64spp-d8"
world, objs, kw = bench.workload(name)
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
import numpy as np  # noqa: E402
r = api.Renderer(0)
rows = []
# the application keeps one buffer from render to render; "pinned" as a second argument takes it from pt_image_alloc
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"
out = r.image_alloc(st.width, st.height) if pinned else np.zeros((st.height, st.width, 3), np.float32)
for shot in range(8):
    t0 = time.perf_counter()
    r.start(world, objs, st, out=out)
    r.wait()
    dt = (time.perf_counter() - t0) * 1e3
    s = r.stats()
    rows.append({"total_ms": round(dt, 2), "prepare_ms": round(s.prepare_ms, 2), "trace_ms": round(s.trace_ms, 2)})
print(json.dumps({"workload": name, "image": "pt_image_alloc" if pinned else "pageable", "paths": r.stats().paths, "shots": rows,
                  "mpaths_per_s_last": r.stats().paths / rows[-1]["total_ms"] / 1e3}))
r.close()
