"""Diagnosis aid (not collected by pytest; CPU only): which hits that the oracle accepts would the cooperative cast's
bounding-sphere filter have turned away?  The oracle calls a hook for every hit its grid cast accepts; the hook repeats the
filter's arithmetic (pt_device.h, sphere_rejects) on the entity's filter sphere (pt_host.cpp, filter_sphere).
usage: python tests/diag_filter.py <workload> x,y [x,y ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api
from oracle.oraclebind import FP, OracleScene

f32 = np.float32
HOOK = C.CFUNCTYPE(None, C.c_int, FP, FP, FP, C.c_float)


def fma(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def sphere_of(o):
    half = np.array(o.dim[:], np.float64) / 2
    pos = np.array(o.pos[:], np.float64)
    if o.type == api.PT_GEOM_PLANE:
        return pos, np.inf
    if o.type == api.PT_GEOM_SPHERE:
        return pos, half.max()
    if o.type == api.PT_GEOM_BOX:
        return pos, np.sqrt((half ** 2).sum())
    if o.type == api.PT_GEOM_CYLINDER:
        return pos, np.sqrt(max(half[0], half[1]) ** 2 + half[2] ** 2)
    rim, height = max(half[0], half[1]), 2 * half[2]
    above = (height ** 2 - rim ** 2) / (2 * height) if height > rim else 0.0
    qw, qx, qy, qz = (float(v) for v in o.rot)
    axis = np.array([2 * (qx * qz + qw * qy), 2 * (qy * qz - qw * qx), 1 - 2 * (qx * qx + qy * qy)])
    return pos + axis * (above - height / 2), ((height ** 2 + rim ** 2) / (2 * height) if height > rim else rim)


def rejects(rp, rd, centre, radius):
    if not np.isfinite(radius):
        return False, {}
    l = (rp - centre.astype(f32)).astype(f32)
    dd, dl, ll = f32(np.dot(rd, rd)), f32(np.dot(rd, l)), f32(np.dot(l, l))
    r = fma(f32(radius), f32(1.002), f32(1e-3))
    r2 = f32(r * r)
    dl2 = f32(dl * dl)
    slack = f32(f32(1e-5) * fma(dd, ll, dl2))
    disc = fma(-dd, f32(ll - r2), dl2)
    may = (disc + slack >= 0) and (dl <= 0 or ll <= fma(r2, f32(1.0001), slack))
    return (not may), dict(dl=float(dl), ll=float(ll), r2=float(r2), slack=float(slack), disc=float(disc))


def cone_slope(o, rd):
    """a = dx^2 + dy^2 - dz^2 of the ray's direction in the cone's unit space (double precision)."""
    qw, qx, qy, qz = (np.float64(v) for v in o.rot)
    u = np.array([-qx, -qy, -qz])
    v = rd.astype(np.float64)
    t = 2 * np.cross(u, v)
    d = (v + qw * t + np.cross(u, t)) * np.array([2 / o.dim[0], 2 / o.dim[1], 1 / o.dim[2]])
    d = d / np.linalg.norm(d)
    return d[0] ** 2 + d[1] ** 2 - d[2] ** 2


def turned_away(world, objs, st, pixels, say=None):
    """Per pixel: the accepted hits of its paths that the filter would have turned away, as (entity, type, slope) tuples."""
    sc = OracleScene(world, objs, st)
    found = []

    def hook(oi, rp, rd, pos, dist):
        rp = np.array([rp[0], rp[1], rp[2]], f32)
        rd = np.array([rd[0], rd[1], rd[2]], f32)
        centre, radius = sphere_of(objs[oi])
        bad, info = rejects(rp, rd, centre, radius)
        if bad:
            o = objs[oi]
            slope = cone_slope(o, rd) if o.type == api.PT_GEOM_CONE else None
            found.append((oi, int(o.type), slope))
            if say:
                hp = np.array([pos[0], pos[1], pos[2]], np.float64)
                say(f"  turned away: entity {oi} type {o.type} dim {[round(v, 4) for v in o.dim]} radius {radius:.4f} distance {dist:.4f} "
                    f"hit point {np.linalg.norm(hp - centre):.4f} from the sphere's centre; unit-space slope term {slope}; {info}")
    cb = HOOK(hook)
    sc.lib.pto_set_hit_hook.argtypes = [HOOK]
    sc.lib.pto_set_hit_hook.restype = None
    sc.lib.pto_set_hit_hook(cb)
    one = np.zeros((st.height, st.width, 3), f32)
    out = []
    for x, y in pixels:
        del found[:]
        sc.lib.pto_render_rect(sc.h, 0, 3, int(x), int(y), 1, 1, one.ctypes.data_as(FP))
        out.append(list(found))
    sc.lib.pto_set_hit_hook(C.cast(None, HOOK))
    sc.close()
    return out


if __name__ == "__main__":
    import bench
    world, objs, kw = bench.workload(sys.argv[1])
    st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
    pixels = [tuple(int(v) for v in xy.split(",")) for xy in sys.argv[2:]]
    for p, hits in zip(pixels, turned_away(world, objs, st, pixels, say=lambda m: print(m, flush=True))):
        print("pixel", p, ":", len(hits), "accepted hits the filter would have turned away", flush=True)
