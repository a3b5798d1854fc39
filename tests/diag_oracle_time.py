"""Diagnosis aid (not collected by pytest): how long the oracle takes for a whole frame of a workload, and how the GPU's frame compares.
usage: python tests/diag_oracle_time.py <workload> [threads]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import api
from oracle.oraclebind import OracleScene
name = sys.argv[1]; threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
world, objs, kw = bench.workload(name)
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
gpu = api.Renderer(0)
img = gpu.render(world, objs, st)
sc = OracleScene(world, objs, st)
t0 = time.time()
ref = sc.render(policy=3, threads=threads)
print(name, "oracle whole frame on", threads, "threads:", round(time.time() - t0, 1), "s; paths", sc.path_count(), flush=True)
sc.close()
d = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
fin = np.isfinite(ref).all(axis=2)
print("pixels that differ:", int(d.sum()), "of", d.size, "; not finite in the oracle:", int((~fin).sum()), "; largest relative error", float((np.abs(img[fin].astype(np.float64) - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-3)).max()), flush=True)
