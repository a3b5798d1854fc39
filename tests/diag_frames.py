"""Whole frames of one benchmark workload rendered under different settings of the library's environment switches (they are
read once per process: PT_GRID_PER_LANE, PT_CELL_CULL, PT_CORE_LIB ...), compared bit for bit; what differs is put to the oracle.
Not collected by pytest; the GPU parity suite runs the render step as a child process.
usage: python tests/diag_frames.py render <workload> <file.npy>
       python tests/diag_frames.py compare <workload> <a.npy> <b.npy>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import api
from oracle.oraclebind import FP, OracleScene

mode, name = sys.argv[1], sys.argv[2]
world, objs, kw = bench.workload(name)
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
if mode == "render":
    gpu = api.Renderer(0)
    gpu.prepare(world, objs, st)
    gpu.trace()
    np.save(sys.argv[3], gpu.download())
    print("rendered", name, {k: v for k, v in os.environ.items() if k.startswith("PT_")}, flush=True)
else:
    a, b = np.load(sys.argv[3]), np.load(sys.argv[4])
    d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
    ys, xs = np.nonzero(d)
    print(name, "pixels that differ:", len(ys), flush=True)
    if len(ys):
        sc = OracleScene(world, objs, st)
        one = np.zeros((st.height, st.width, 3), np.float32)
        for y, x in list(zip(ys, xs))[:10]:
            sc.lib.pto_render_rect(sc.h, 0, 3, int(x), int(y), 1, 1, one.ctypes.data_as(FP))
            want = one.view(np.uint32)[y, x]
            print((int(x), int(y)), "first equals oracle:", bool((a.view(np.uint32)[y, x] == want).all()), "second:", bool((b.view(np.uint32)[y, x] == want).all()), a[y, x].tolist(), b[y, x].tolist(), one[y, x].tolist(), flush=True)
        sc.close()
