"""Where a wavefront spends a progressive frame (needs a library built with -DPT_COOP_TIMING): hand-out, camera ray, cast, shade.  usage: PT_CORE_LIB=.../variant_timing.so python profiles/progressive_phases.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, scenes
world, objs = scenes.default_scene()
st = api.default_settings(1920, 1080, 64, 8, seed=12345, frame_index=0)
r = api.Renderer(0); r.prepare(world, objs, st)
buf = (C.c_ulonglong * 16)()
r.lib.pt_debug_coop_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for f in range(20):
    r.set_frame(f); r.trace()
r.lib.pt_debug_coop_stats(r.ctx, buf)
r.set_frame(20); r.trace()
r.lib.pt_debug_coop_stats(r.ctx, buf)
s = r.stats()
print("kernel_ms", s.dominant_kernel_ms, "waves", buf[5], "span us", (buf[4]-buf[3])/100.0, "mean wave busy us", buf[2]/max(1,buf[5])/100.0)
tot = buf[0]
for i, n in ((10, "handout"), (11, "camera"), (14, "cast"), (12, "shade")):
    print(f"  {n:8s} {100.0*buf[i]/max(1,tot):5.1f} % of wave cycles")
