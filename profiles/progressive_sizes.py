"""Kernel time of a progressive frame (one sample per pixel) against the image size: the fixed part of a launch (start-up, group dealing, tail) beside the part that grows with the work.  usage: python profiles/progressive_sizes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, scenes
world, objs = scenes.default_scene()
for (w, h) in ((64, 64), (480, 270), (960, 540), (1920, 1080), (3840, 2160)):
    st = api.default_settings(w, h, 64, 8, seed=12345, frame_index=0)
    r = api.Renderer(0); r.prepare(world, objs, st)
    for f in range(30):
        r.set_frame(f); r.trace(sync=False)
    r.sync()
    for f in range(100):
        r.set_frame(30 + f); r.trace(sync=False)
    r.sync()
    s = r.stats()
    print(w, h, "groups", ((w + 7) // 8) * ((h + 7) // 8), "kernel_us %.1f min %.1f" % (s.dominant_kernel_ms * 1e3, s.trace_ms_min * 1e3))
    r.close()
