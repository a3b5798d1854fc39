"""Kernel time of a one-sample progressive frame of the demo scene at 1080p, whole and as one partition of 2 / 4 / 8
(what a rank of an N-GPU progressive run executes).  usage: [PT_MAX_BLOCKS_PER_CU=n] python profiles/prog_parts.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, scenes  # noqa: E402

world, objs = scenes.default_scene()
row = []
for parts in (1, 2, 4, 8):
    st = api.default_settings(1920, 1080, 64, 8, seed=12345, frame_index=0, part_index=0, part_count=parts)
    r = api.Renderer(0)
    r.prepare(world, objs, st)
    for f in range(40):
        r.set_frame(f)
        r.trace(sync=False)
    r.sync()
    r.stats()
    for f in range(40, 200):
        r.set_frame(f)
        r.trace(sync=False)
    r.sync()
    s = r.stats()
    row.append("%d: %.4f ms (%d blocks)" % (parts, s.dominant_kernel_ms, s.grid_blocks))
    r.close()
print("progressive frame, partitions " + "   ".join(row))
