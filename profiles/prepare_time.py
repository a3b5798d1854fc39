"""Set-up time (pt_prepare) of the 100 000-entity scene with the grid lists built on the host and on the
device.  usage: python profiles/prepare_time.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pathtracer_amd import api  # noqa: E402

world, objs, kw = bench.workload("dense100k-1080p-16spp-d8")
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
r = api.Renderer(0)
out = {}
for where in ("host", "device", "host", "device"):
    os.environ["PT_GRID_BUILD"] = where
    r.prepare(world, objs, st)
    out.setdefault(where, []).append(round(r.stats().prepare_ms, 2))
print(json.dumps({"workload": "dense100k (100 000 entities, 64^3-hint grid)", "prepare_ms": out, "cell_items": r.stats().cell_items}))
r.close()
