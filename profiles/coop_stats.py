"""Diagnostic: counters of the wave-cooperative cast.  Needs a library built with -DPT_COOP_STATS (event counts) or
-DPT_COOP_TIMING (cycles per phase, summed over wavefronts; add -DPT_COOP_TIMING_WAITS for the exposed memory waits);
the two builds use the same sixteen slots for different things, and this script tells them apart by slot 0 (the
kernel's wave cycles, which only a timing build fills).
usage: PT_CORE_LIB=pathtracer_amd/variant_<name>.so python profiles/coop_stats.py [workload] [waits]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pathtracer_amd import api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "glass-1080p-16spp-d32"
waits_build = len(sys.argv) > 2 and sys.argv[2] == "waits"
world, objs, kw = bench.workload(name)
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
r = api.Renderer(0)
r.prepare(world, objs, st)
buf = (C.c_ulonglong * 16)()
r.lib.pt_debug_coop_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
if r.lib.pt_debug_coop_stats(r.ctx, buf) != 16:
    raise SystemExit("this library is a normal build: no counters (build with -DPT_COOP_STATS or -DPT_COOP_TIMING)")
r.trace()
r.lib.pt_debug_coop_stats(r.ctx, buf)
s = r.stats()
casts = s.rays
print(f"{name}: {s.paths} paths, {casts} ray casts, kernel {s.dominant_kernel_ms:.2f} ms")
if buf[0] == 0:
    # event counts
    labels = {1: "segments", 2: "chunks", 3: "exact tests", 4: "cells recorded", 5: "list entries read", 6: "  of them repeats (listed in the cell before)",
              7: "  of them rejected by the sphere filter", 8: "hits inside a recorded cell", 9: "calls of the cast with walking lanes"}
    calls = max(1, buf[9])
    print(f"{'event':46s} {'count':>14s} {'per call':>10s} {'per ray':>9s}")
    for i in range(1, 10):
        print(f"{labels[i]:46s} {buf[i]:14d} {buf[i] / calls:10.2f} {buf[i] / max(1, casts):9.3f}")
    print(f"lanes with a new ray per call: {casts / calls:.1f} of 64; candidates per chunk: {buf[3] / max(1, buf[2]):.1f} of 64")
    ls = max(1, buf[10])
    print(f"walking lanes per segment: {buf[10] / max(1, buf[1]):.1f} of 64; of these lane-segments: {100 * buf[11] / ls:.1f} % find no list and go on, "
          f"{100 * buf[12] / ls:.1f} % find none and leave the grid, {100 * buf[13] / ls:.1f} % find one list")
    print(f"drains whose last chunk holds at most 16 candidates: {100 * buf[14] / max(1, buf[1]):.1f} % of segments; drains of at most 32 candidates: {100 * buf[15] / max(1, buf[1]):.1f} %")
else:
    total = buf[0]
    print(f"wave cycles, all wavefronts: {total}; segments {buf[1]}")
    rows = [(10, "walk (probe steps, list bounds, descriptors)"), (11, "expand (list entries -> candidates)"), (12, "drain (chunks of exact tests)"),
            (13, "verdict"), (14, "cast, all of the above"), (15, "hand-out, camera ray, shade, range sums")]
    for i, label in rows:
        print(f"  {label:50s} {100.0 * buf[i] / total:5.1f} % of wave time   {buf[i] / max(1, buf[1]):9.0f} cycles per segment")
    print("  inside the chunks:")
    for i, label in ((2, "fetch candidate and ray"), (3, "exact intersection"), (4, "search for the hit's cell, key"), (5, "rendezvous and write-back")):
        print(f"    {label:48s} {100.0 * buf[i] / total:5.1f} % of wave time")
    if waits_build:
        for i, label in ((6, "walk until its loads are issued"), (7, "wait for the list bounds"), (8, "wait for the list entries"), (9, "wait for a chunk's first records")):
            print(f"  {label:50s} {100.0 * buf[i] / total:5.1f} % of wave time")
    else:
        print(f"  chunks run: direct kinds {buf[6]}, ellipsoid {buf[7]}, cylinder / cone {buf[8]}, mixed {buf[9]}")
