"""Diagnosis aid (not collected by pytest; CPU only): which hits that the oracle accepts would the cooperative cast's
bounding-sphere filter have turned away?  usage: python tests/diag_filter.py <workload> x,y [x,y ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import api
from oracle.oraclebind import FP, OracleScene

f32 = np.float32
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))

def sphere_of(o):
    half = np.array(o.dim[:], np.float64) / 2
    pos = np.array(o.pos[:], np.float64)
    if o.type == api.PT_GEOM_SPHERE: return pos, half.max()
    if o.type == api.PT_GEOM_BOX: return pos, np.sqrt((half ** 2).sum())
    if o.type == api.PT_GEOM_CYLINDER: return pos, np.sqrt(max(half[0], half[1]) ** 2 + half[2] ** 2)
    rim, height = max(half[0], half[1]), 2 * half[2]
    above = (height ** 2 - rim ** 2) / (2 * height) if height > rim else 0.0
    qw, qx, qy, qz = (float(v) for v in o.rot)
    axis = np.array([2 * (qx * qz + qw * qy), 2 * (qy * qz - qw * qx), 1 - 2 * (qx * qx + qy * qy)])
    return pos + axis * (above - height / 2), ((height ** 2 + rim ** 2) / (2 * height) if height > rim else rim)

def rejects(rp, rd, centre, radius):
    l = (rp - centre.astype(f32)).astype(f32)
    dd, dl, ll = f32(np.dot(rd, rd)), f32(np.dot(rd, l)), f32(np.dot(l, l))
    r = fma(f32(radius), f32(1.002), f32(1e-3)); r2 = f32(r * r); dl2 = f32(dl * dl)
    slack = f32(f32(1e-5) * fma(dd, ll, dl2))
    disc = fma(-dd, f32(ll - r2), dl2)
    may = (disc + slack >= 0) and (dl <= 0 or ll <= fma(r2, f32(1.0001), slack))
    return (not may), dict(dl=float(dl), ll=float(ll), r2=float(r2), slack=float(slack), disc=float(disc))

name = sys.argv[1]
world, objs, kw = bench.workload(name)
st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
sc = OracleScene(world, objs, st)
HOOK = C.CFUNCTYPE(None, C.c_int, FP, FP, FP, C.c_float)
seen = [0]
def hook(oi, rp, rd, pos, dist):
    seen[0] += 1
    rp = np.array([rp[0], rp[1], rp[2]], f32); rd = np.array([rd[0], rd[1], rd[2]], f32); hp = np.array([pos[0], pos[1], pos[2]], f32)
    centre, radius = sphere_of(objs[oi])
    bad, info = rejects(rp, rd, centre, radius)
    if bad:
        o = objs[oi]
        if o.type == api.PT_GEOM_CONE:
            qw, qx, qy, qz = (np.float64(v) for v in o.rot)
            def rot_inv(v):
                u = np.array([-qx, -qy, -qz]); t = 2 * np.cross(u, v); return v + qw * t + np.cross(u, t)
            sinv = np.array([2 / o.dim[0], 2 / o.dim[1], 1 / o.dim[2]])
            a = rot_inv(rp.astype(np.float64) - np.array(o.pos[:], np.float64)) * sinv
            d = rot_inv(rd.astype(np.float64)) * sinv
            d = d / np.linalg.norm(d)
            pz = a[2] - 0.5
            qa = d[0] ** 2 + d[1] ** 2 - d[2] ** 2; qb = 2 * (a[0] * d[0] + a[1] * d[1] - pz * d[2]); qc = a[0] ** 2 + a[1] ** 2 - pz ** 2
            disc = qb * qb - 4 * qa * qc
            root = np.sqrt(max(disc, 0))
            t1, t2 = (-qb + root) / (2 * qa), (-qb - root) / (2 * qa)
            print("  unit space: p", a.round(4).tolist(), "d", d.round(5).tolist(), "a b c", round(qa, 6), round(qb, 4), round(qc, 4), "disc", round(disc, 4),
                  "t1 t2", round(t1, 4), round(t2, 4), "z1 z2", round(pz + d[2] * t1, 4), round(pz + d[2] * t2, 4), "t_base", round((-1 - pz) / d[2], 4))
        off = np.linalg.norm(hp.astype(np.float64) - centre)
        print("  turned away: entity", oi, "type", o.type, "dim", [round(v, 4) for v in o.dim], "rot", [round(v, 3) for v in o.rot], "radius", round(radius, 4),
              "distance", dist, "hit point is", round(off, 4), "from the centre;", {k: round(v, 6) for k, v in info.items()}, flush=True)
cb = HOOK(hook)
sc.lib.pto_set_hit_hook.argtypes = [HOOK]
sc.lib.pto_set_hit_hook.restype = None
sc.lib.pto_set_hit_hook(cb)
one = np.zeros((st.height, st.width, 3), f32)
for xy in sys.argv[2:]:
    x, y = (int(v) for v in xy.split(","))
    seen[0] = 0
    print("pixel", (x, y), flush=True)
    sc.lib.pto_render_rect(sc.h, 0, 3, x, y, 1, 1, one.ctypes.data_as(FP))
    print("  accepted hits looked at:", seen[0], flush=True)
sc.lib.pto_set_hit_hook(C.cast(None, HOOK))
sc.close()
