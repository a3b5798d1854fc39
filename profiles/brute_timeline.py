"""Diagnostic (library built with -DPT_COOP_TIMING): how evenly the wavefronts of a brute-force frame end.  For partition 0 of
N of the demo frame: the kernel's span on the chip-wide clock (first wavefront's start to last wavefront's end), the mean
time a wavefront is busy, and the phase split of the busy time.  usage: PT_CORE_LIB=... python profiles/brute_timeline.py [parts ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pathtracer_amd import api  # noqa: E402

world, objs, kw = bench.workload("demo-1080p-64spp-d8")
for parts in [int(a) for a in sys.argv[1:]] or [1, 8]:
    skw = dict(kw)
    st = api.default_settings(skw.pop("width"), skw.pop("height"), skw.pop("spp"), skw.pop("bounces"), part_index=0, part_count=parts, **skw)
    r = api.Renderer(0)
    r.prepare(world, objs, st)
    buf = (C.c_ulonglong * 16)()
    r.lib.pt_debug_coop_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    for _ in range(5):
        r.trace()
    if r.lib.pt_debug_coop_stats(r.ctx, buf) != 16 or buf[0] == 0:
        raise SystemExit("needs a -DPT_COOP_TIMING build")
    r.trace()
    r.lib.pt_debug_coop_stats(r.ctx, buf)
    s = r.stats()
    waves = max(1, buf[5])
    span_us = (buf[4] - buf[3]) / 100.0
    busy_us = buf[2] / waves / 100.0
    print(f"partition 0 of {parts}: kernel {s.dominant_kernel_ms * 1e3:.0f} us, {waves} wavefronts, span first start -> last end {span_us:.1f} us, "
          f"mean wavefront busy {busy_us:.1f} us = {100 * busy_us / span_us:.0f} % of the span, groups {r.stats().grid_blocks} blocks")
    total = max(1, buf[0])
    for i, label in ((10, "pixel hand-out"), (11, "camera ray"), (14, "cast"), (12, "shade, range sums")):
        print(f"    {label:20s} {100.0 * buf[i] / total:5.1f} % of wave cycles")
    r.close()
