"""A longer run of the ray-cast comparison than the test suite holds: many random scenes, brute-force sized (3 to 9 entities,
every ray tests every entity) and grid sized (20 to 120 entities), ids, hit positions and normals of the device against the
oracle bit for bit.  From seed 1000 on every fourth scene also holds one to three planes (code/raycast.cpp:885-889, 937-955),
and one scene in twenty-five has some thousand entities, enough for the grid to be built on the device.  Not collected by pytest; run on the GPU box:  python tests/soak_cast_parity.py [first_seed] [scenes]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oraclebind import OracleScene  # noqa: E402
from pathtracer_amd import api, scenes  # noqa: E402
from rayutil import scene_rays  # noqa: E402

P = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
IP = C.POINTER(C.c_int32)


def with_planes(objs, seed):
    """Seeds from 1000 on, every fourth: one to three planes at random places of the entity list, tilted at random."""
    if seed < 1000 or seed % 4 != 3:
        return objs
    rng = np.random.default_rng(seed)
    objs = list(objs)
    for _ in range(1 + seed % 3):
        axis = rng.normal(size=3)
        rot = scenes._axis_quat(tuple(axis / np.linalg.norm(axis)), float(rng.uniform(0, 180)))
        pos = tuple(rng.uniform(-12, 12, size=3))
        plane = api.make_object(pos, (1, 1, 1), api.PT_GEOM_PLANE, rot=rot, color=tuple(rng.uniform(0.2, 0.9, size=3)), refl=float(rng.choice([0.0, 0.5, 1.0])))
        objs.insert(int(rng.integers(0, len(objs) + 1)), plane)
    return objs


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
gpu = api.Renderer(0)
rays = bad_scenes = 0
for seed in range(first, first + count):
    small = seed % 2 == 0
    n_obj = 3 + seed % 7 if small else 20 + (seed * 7) % 101
    world, objs = scenes.mixed_scene(seed=seed, n=n_obj, spread=6.0 + seed % 9)
    objs = with_planes(objs, seed)
    st = api.default_settings(64, 36, 4, 4, seed=4242, grid_hint=6 + seed % 7, brute_force_below=10)
    if seed >= 1000 and seed % 25 == 7:
        # now and then a scene large enough for the grid to be built on the device (4096 entities and more)
        world, objs = scenes.dense_scene(n=4096 + (seed * 37) % 3000, seed=seed, half_extent=50.0 + seed % 30, height=20.0 + seed % 20)
        st = api.default_settings(64, 36, 4, 4, seed=4242, grid_hint=16 + seed % 17, brute_force_below=10)
    gpu.prepare(world, objs, st)
    sc = OracleScene(world, objs, st)
    n = 20000
    o, d = scene_rays(n, seed, objs, spread=20.0)
    ids, pos, nrm = gpu.cast_rays(o, d)
    rid, rpos, rnrm = np.zeros(n, np.int32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    sc.lib.pto_cast(sc.h, n, P(o), P(d), 0 if len(objs) < st.brute_force_below else 1, rid.ctypes.data_as(IP), P(rpos), P(rnrm))
    sc.close()
    hit = rid != 0
    same = (ids == rid).all() and (pos[hit].view(np.uint32) == rpos[hit].view(np.uint32)).all() and (nrm[hit].view(np.uint32) == rnrm[hit].view(np.uint32)).all()
    rays += n
    bad_scenes += 0 if same else 1
    print(f"seed {seed}: {len(objs)} entities, {int(hit.sum())} of {n} rays hit, {'identical' if same else 'DIFFERENT: %d ids' % int((ids != rid).sum())}", flush=True)
print(f"{count} scenes, {rays} rays, {bad_scenes} scenes with a difference")
# whole images of the same kind of scenes without rough surfaces (no libm call involved): identical bits expected
bad_images = 0
for seed in range(first, first + count // 2):
    n_obj = 3 + seed % 7 if seed % 2 == 0 else 20 + (seed * 5) % 60
    world, objs = scenes.mixed_scene(seed=seed, n=n_obj, rough_share=0.0, spread=6.0 + seed % 9)
    objs = with_planes(objs, seed)
    st = api.default_settings(96, 54, 8, 4 + 4 * (seed % 3), seed=seed, grid_hint=6 + seed % 7, brute_force_below=10)
    img = gpu.render(world, objs, st)
    sc = OracleScene(world, objs, st)
    ref = sc.render(policy=api.PT_SEED_PER_SAMPLE, threads=8)
    sc.close()
    same = (img.view(np.uint32) == ref.view(np.uint32)).all()
    bad_images += 0 if same else 1
    print(f"image seed {seed}: {len(objs)} entities, depth {st.max_bounces}, {'identical' if same else 'DIFFERENT: %d values' % int((img.view(np.uint32) != ref.view(np.uint32)).sum())}", flush=True)
gpu.close()
print(f"{count // 2} images, {bad_images} with a difference")
sys.exit(1 if bad_scenes or bad_images else 0)
