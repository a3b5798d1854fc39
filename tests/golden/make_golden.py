"""Generates tests/golden/ref_vectors.npz from the reference's own code (oracle/_ref/libptref.so).

Run in the build container, where /root/reference exists and `make -C oracle ref` has produced the
library:   python tests/golden/make_golden.py
The file holds inputs and the reference's outputs only (data, no code); tests/test_oracle_golden.py
replays the inputs through the CPU oracle on machines that have neither the reference nor the
library.
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oraclebind  # noqa: E402
from oracle.oraclebind import FP, IP, UP, U8P  # noqa: E402
from pathtracer_amd import scenes  # noqa: E402
from pathtracer_amd.api import default_settings, object_array, pt_object  # noqa: E402

P = lambda a: a.ctypes.data_as(FP)  # noqa: E731
CASES = {
    # name: (scene factory, width, height, spp, bounces, seed)
    "default": (lambda: scenes.default_scene(), 96, 54, 16, 8, 12345),
    "mixed40": (lambda: scenes.mixed_scene(seed=7, n=40), 80, 45, 16, 8, 777),
    "glass": (lambda: scenes.glass_scene(seed=3, per_type=16), 80, 45, 16, 32, 4242),
    # GEOM_TYPE_PLANE (code/raycast.cpp:885-889): under the brute-force threshold, and in a grid scene
    "planes_small": (lambda: scenes.planes_small_scene(), 80, 45, 16, 8, 31),
    "planes_grid": (lambda: scenes.planes_grid_scene(), 80, 45, 16, 8, 32),
}


from rayutil import rays, scene_rays  # noqa: E402


def objects_blob(objs):
    arr = object_array(objs)
    return np.frombuffer(bytes(arr), np.uint8)[: C.sizeof(pt_object) * len(objs)].copy()


def main():
    ref = oraclebind.load_ref()
    out = {}
    # PCG
    for k, (state, seq) in enumerate([(0, 0), (0, 12345), (42, 54), (2**40 + 7, 2**33 + 1)]):
        ref.ref_pcg_seed(state, seq)
        u = np.zeros(16, np.uint32)
        ref.ref_pcg_u32(16, u.ctypes.data_as(UP))
        i = np.zeros(64, np.int32)
        ref.ref_pcg_int(0, 19999, 64, i.ctypes.data_as(IP))
        f = np.zeros(64, np.float32)
        ref.ref_pcg_float(0, 1, 0.001, 64, P(f))
        out[f"pcg{k}_seed"] = np.array([state, seq], np.uint64)
        out[f"pcg{k}_u32"], out[f"pcg{k}_int"], out[f"pcg{k}_float"] = u, i, f

    # raw intersectors
    n = 1024
    rng = np.random.default_rng(5)
    sp = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    sr = rng.uniform(0.2, 4, n).astype(np.float32)
    bd = rng.uniform(0.1, 8, (n, 3)).astype(np.float32)
    o, d = rays(n, 11, 6.0, aim=sp, aim_jitter=1.5)
    uo, ud = rays(n, 12, 2.5, aim=np.zeros((1, 3)), aim_jitter=1.0)
    out["isect_o"], out["isect_d"], out["isect_sp"], out["isect_sr"], out["isect_bd"] = o, d, sp, sr, bd
    out["isect_uo"], out["isect_ud"] = uo, ud
    for fn, args in [("line_sphere", (o, d, sp, sr)), ("line_box", (o, d, sp, bd)), ("line_cylinder", (uo, ud)),
                     ("line_cone", (uo, ud))]:
        dist, pos, nrm = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        getattr(ref, "ref_" + fn)(n, *[P(a) for a in args], P(dist), P(pos), P(nrm))
        hit = dist != -1
        pos[~hit] = 0
        nrm[~hit] = 0
        out[fn + "_dist"], out[fn + "_pos"], out[fn + "_nrm"] = dist, pos, nrm
    coll = np.zeros(n, np.int32)
    ref.ref_sphere_collision(n, P(o), P(d), P(sp), P(sr), coll.ctypes.data_as(IP))
    out["sphere_collision"] = coll
    _, nn = rays(n, 22)
    ior = rng.uniform(1.0, 3.0, n).astype(np.float32)
    ior[:32] = 1.0
    kr, rf, rl = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    ref.ref_optics(n, P(d), P(nn), P(ior), P(kr), P(rf), P(rl))
    out["optics_n"], out["optics_ior"], out["optics_kr"], out["optics_refract"], out["optics_reflect"] = nn, ior, kr, rf, rl

    # scenes
    for name, (factory, w, h, spp, bounces, seed) in CASES.items():
        world, objs = factory()
        st = default_settings(w, h, spp, bounces, seed=seed)
        ref.ref_pcg_seed(0, st.seed)
        dirs, disc = np.zeros((20000, 3), np.float32), np.zeros((20000, 2), np.float32)
        ref.ref_tables(P(dirs), P(disc))
        oa = object_array(objs)
        ns = ref.ref_scene_setup(C.byref(world), oa, len(objs), C.byref(st), 1)
        pre = name + "_"
        out[pre + "objects"] = objects_blob(objs)
        out[pre + "world"] = np.frombuffer(bytes(world), np.uint8).copy()
        out[pre + "settings"] = np.frombuffer(bytes(st), np.uint8).copy()
        out[pre + "dirs_head"], out[pre + "disc_head"] = dirs[:64].copy(), disc[:64].copy()
        out[pre + "tables_sha"] = np.frombuffer(hashlib.sha256(dirs.tobytes() + disc.tobytes()).digest(), np.uint8).copy()
        smp, offs = np.zeros((ns, 2), np.float32), np.zeros(10, np.int32)
        ref.ref_get_samples(P(smp), offs.ctypes.data_as(IP))
        out[pre + "samples"], out[pre + "offsets"] = smp, offs
        rec = np.zeros((len(objs), 36), np.float32)
        ref.ref_get_records(P(rec))
        for i, o in enumerate(objs):
            if o.type == 4:      # a plane's bounding radius and unit-space scales: never set, never read by the reference - kept as zeros
                rec[i, 29] = 0
                rec[i, 14:20] = 0
        out[pre + "records"] = rec
        wc = np.zeros(27, np.float32)
        ref.ref_get_world(P(wc))
        out[pre + "worldconsts"] = wc
        g = np.zeros(9, np.int32)
        ref.ref_get_grid(g.ctypes.data_as(IP))
        out[pre + "grid"] = g
        cells = int(g[3] * g[4] * g[5])
        start, items = [0], []
        buf = np.zeros(4096, np.int32)
        for c in range(cells):
            k = ref.ref_get_cell(c, buf.ctypes.data_as(IP), 4096)
            items.extend(buf[:k].tolist())
            start.append(len(items))
        out[pre + "cell_start"], out[pre + "cell_items"] = np.array(start, np.uint32), np.array(items, np.uint32)

        nr = 1500
        ro, rd = scene_rays(nr, 31, objs)
        out[pre + "ray_o"], out[pre + "ray_d"] = ro, rd
        oi = np.random.default_rng(3).integers(0, len(objs), nr).astype(np.int32)
        dist, pos, nrm = np.zeros(nr, np.float32), np.zeros((nr, 3), np.float32), np.zeros((nr, 3), np.float32)
        ref.ref_get_intersection(nr, oi.ctypes.data_as(IP), P(ro), P(rd), P(dist), P(pos), P(nrm))
        pos[dist == -1] = 0
        nrm[dist == -1] = 0
        out[pre + "gi_obj"], out[pre + "gi_dist"], out[pre + "gi_pos"], out[pre + "gi_nrm"] = oi, dist.copy(), pos.copy(), nrm.copy()
        cnt, seqhash, outside = np.zeros(nr, np.int32), np.zeros(nr, np.uint64), np.zeros(nr, np.int32)
        cells_buf = np.zeros((512, 3), np.int32)
        ov = C.c_int32(0)
        for i in range(nr):
            k = ref.ref_traverse(P(ro[i]), P(rd[i]), 512, cells_buf.ctypes.data_as(IP), C.byref(ov))
            cnt[i], outside[i] = k, ov.value
            hsh = 1469598103934665603
            for c in cells_buf[:k]:
                idx = int(c[2]) * int(g[4]) * int(g[3]) + int(c[1]) * int(g[3]) + int(c[0])
                hsh = ((hsh ^ ((idx + 1) & 0xFFFFFFFF)) * 1099511628211) & ((1 << 64) - 1)
            seqhash[i] = hsh
        out[pre + "dda_count"], out[pre + "dda_hash"], out[pre + "dda_outside"] = cnt, seqhash, outside
        for use_grid in (0, 1):
            ids = np.zeros(nr, np.int32)
            ref.ref_cast(nr, P(ro), P(rd), use_grid, ids.ctypes.data_as(IP), P(pos), P(nrm))
            pos[ids == 0] = 0
            nrm[ids == 0] = 0
            out[pre + f"cast{use_grid}_id"], out[pre + f"cast{use_grid}_pos"], out[pre + f"cast{use_grid}_nrm"] = ids, pos.copy(), nrm.copy()
        img = np.zeros((h, w, 3), np.float32)
        ref.ref_render(st.seed, P(img))
        out[pre + "image"] = img
        if world.aperture == 0:
            hits = np.zeros((h, w), np.int32)
            ref.ref_primary_hits(hits.ctypes.data_as(IP))
            out[pre + "primary"] = hits.astype(np.int16)

    v = np.linspace(0, 1, 1025).astype(np.float32)
    s8 = np.zeros(v.size, np.uint8)
    ref.ref_srgb8(v.size, P(v), s8.ctypes.data_as(U8P))
    out["srgb_in"], out["srgb_out"] = v, s8
    # a scene file written by the reference's own saveScene (code/raycast.cpp:1578-1588)
    world, objs = scenes.mixed_scene(seed=21, n=11)
    st = default_settings(64, 36, 4, 2)
    ref.ref_pcg_seed(0, 1)
    ref.ref_scene_setup(C.byref(world), object_array(objs), len(objs), C.byref(st), 1)
    ref.ref_save_scene(os.path.join(ROOT, "tests", "golden", "mixed12.scene").encode())

    path = os.path.join(ROOT, "tests", "golden", "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
