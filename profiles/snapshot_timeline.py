"""How the pixel count of pt_render_snapshot grows while a frame is traced (asynchronous render, glass scene)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import torch  # noqa: F401
except ImportError:
    pass
import numpy as np
from pathtracer_amd import api, scenes

world, objs = scenes.glass_scene(seed=3, per_type=16)
W, H = 1920, 1080
st = api.default_settings(W, H, 256, 32, seed=4242)
r = api.Renderer(0)
out = np.zeros((H, W, 3), np.float32)
snap = np.zeros_like(out)
for rep in range(2):
    r.start(world, objs, st, out=out)
    t0 = time.time()
    rows = []
    while True:
        p = r.progress()
        if p.done:
            break
        t1 = time.time()
        n = r.snapshot(snap)
        rows.append((round((t1 - t0) * 1e3, 2), round((time.time() - t1) * 1e3, 2), round(p.fraction, 3), n))
    r.wait()
    p = r.progress()
    print("render", rep, "seconds", round(p.seconds, 4), "polls", len(rows))
    for row in [q for q in rows if q[1] > 0.05][:40]:
        print("  t_ms %8.2f  snapshot_ms %6.2f  fraction %.3f  pixels %d" % row)
r.close()
