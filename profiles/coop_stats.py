"""Diagnostic: event counters of the wave-cooperative cast (needs a library built with -DPT_COOP_STATS)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pathtracer_amd import api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "glass-1080p-16spp-d32"
world, objs, kw = bench.workload(name)
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
r = api.Renderer(0)
r.prepare(world, objs, st)
buf = (C.c_ulonglong * 16)()
r.lib.pt_debug_coop_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
r.lib.pt_debug_coop_stats(r.ctx, buf)
r.trace()
r.lib.pt_debug_coop_stats(r.ctx, buf)
s = r.stats()
names = ["", "segments", "chunks", "tests", "cells", "items", "deduped", "sphere_rejected", "accepted", "wave_casts"]
casts = s.rays
print(name, "casts", casts, "kernel_ms %.2f" % s.trace_ms)
names += ["cyc_walk|handout", "cyc_expand|camera", "cyc_drain|shade", "cyc_verdict", "cyc_cast", "cyc_generate_shade"]
print("  wave cycles total %d" % buf[0])
if buf[5]:
    span = (buf[4] - buf[3]) / 100.0
    print("  waves %d  kernel span %.1f us  mean wave busy %.1f us  = %.1f %% of the span" % (buf[5], span, buf[2] / buf[5] / 100.0, buf[2] / buf[5] / (buf[4] - buf[3]) * 100))
tn = {2: "chunk_fetch", 3: "chunk_intersect", 4: "chunk_cellsearch", 5: "chunk_writeback"}
for i in range(2, 6):
    print(f"  {tn[i]:20s} {buf[i]:14d}  {100.0 * buf[i] / max(1, buf[0]):5.1f} % of wave time   per segment {buf[i] / max(1, buf[1]):8.0f}")
print("  chunks direct/ellipsoid/cyl+cone/mixed", buf[6], buf[7], buf[8], buf[9])
print("  (a -DPT_COOP_TIMING_WAITS build keeps cycles in those four instead: walk until its loads are issued %.1f %%, wait for them %.1f %%,"
      " wait for list entries %.1f %%, wait for a chunk's first records %.1f %% of wave time)" % tuple(100.0 * buf[i] / max(1, buf[0]) for i in (6, 7, 8, 9)))
for i in range(10, 16):
    print(f"  {names[i]:20s} {buf[i]:14d}  {100.0 * buf[i] / max(1, buf[0]):5.1f} % of wave time   per segment {buf[i] / max(1, buf[1]):8.0f}")
for i in range(1, 10):
    print(f"  {names[i]:16s} {buf[i]:12d}   per wave-cast {buf[i] / max(1, buf[9]):8.2f}   per lane-cast {buf[i] / max(1, casts):7.3f}")
