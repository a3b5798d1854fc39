"""Diagnosis aid (not collected by pytest): render one benchmark workload several times on the GPU, report the pixels
that differ between runs and what the oracle says for them.
usage: python tests/diag_repeat.py <workload> <runs>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
from pathtracer_amd import api
from oracle.oraclebind import FP, OracleScene

name = sys.argv[1] if len(sys.argv) > 1 else "dense100k-4k-256spp-d8"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
world, objs, kw = bench.workload(name)
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
gpu = api.Renderer(0)
gpu.prepare(world, objs, st)
gpu.tile_cost()
imgs = []
for r in range(runs):
    gpu.trace()
    imgs.append(gpu.download().copy())
    print("run", r, "paths", gpu.stats().paths, flush=True)
ref = imgs[0].view(np.uint32)
odd = np.zeros(ref.shape[:2], bool)
for r in range(1, runs):
    d = (imgs[r].view(np.uint32) != ref).any(axis=2)
    print("run", r, "differs from run 0 in", int(d.sum()), "pixels", flush=True)
    odd |= d
ys, xs = np.nonzero(odd)
print("pixels that differ in some run:", len(ys))
if len(ys):
    sc = OracleScene(world, objs, st)
    one = np.zeros((st.height, st.width, 3), np.float32)
    for y, x in list(zip(ys, xs))[:12]:
        sc.lib.pto_render_rect(sc.h, 0, 3, int(x), int(y), 1, 1, one.ctypes.data_as(FP))
        vals = [tuple(int(v) for v in imgs[r].view(np.uint32)[y, x]) for r in range(runs)]
        want = tuple(int(v) for v in one.view(np.uint32)[y, x])
        print((int(x), int(y)), "oracle", want, "runs equal to it:", [int(v == want) for v in vals], [imgs[r][y, x].tolist() for r in range(runs)][:2], one[y, x].tolist(), flush=True)
    sc.close()
