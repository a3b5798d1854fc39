"""Kernel time of a one-sample progressive frame of the demo scene at a given size, three runs.  usage: [PT_GRID_BLOCKS=n] python profiles/prog_4k.py [w h]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, scenes
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
world, objs = scenes.default_scene()
st = api.default_settings(w, h, 64, 8, seed=12345, frame_index=0)
r = api.Renderer(0); r.prepare(world, objs, st)
out = []
for rep in range(4):
    for f in range(60):
        r.set_frame(f); r.trace(sync=False)
    r.sync(); r.stats()
    for f in range(200):
        r.set_frame(60 + f); r.trace(sync=False)
    r.sync()
    s = r.stats()
    out.append("%.1f" % (s.dominant_kernel_ms * 1e3))
print(w, h, "blocks", s.grid_blocks, "kernel_us", " ".join(out))
