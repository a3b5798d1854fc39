"""Longer run than the test suite wants (not collected by pytest): whole frames of random scenes (four in five above the
brute-force threshold), every pixel against the oracle's frame (sixteen host threads).  Scenes without rough entities must be the
oracle's bits, scenes with rough entities within the radiance tolerance, every pixel.
With PT_FAST_CONES=1 (the cooperative cast's plain variants also for scenes with cones) pixels may be off where the
oracle accepted a hit that the bounding-sphere filter turns away - and each of those must be a cone hit by a ray parallel
to one of its generators (the reference intersector's phantom hit; how many there are depends on the scene's cones and is
reported, not judged).
usage: python tests/soak_whole_frames.py [first_seed [count]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pathtracer_amd import api, scenes
from oracle.oraclebind import OracleScene
from diag_filter import turned_away

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gpu = api.Renderer(0)
bad = 0
total_pixels = total_differ = 0
for seed in range(first, first + count):
    kind = seed % 6
    if kind == 0:
        world, objs = scenes.mixed_scene(seed=seed, n=20 + (seed * 7) % 101, rough_share=0.0, spread=8.0 + seed % 13)
        st = api.default_settings(960, 540, 32, 8 + seed % 9, seed=1000 + seed, grid_hint=6 + seed % 7, brute_force_below=10)
    elif kind == 1:
        world, objs = scenes.dense_scene(n=3000 + (seed * 977) % 20000, seed=seed, half_extent=60.0 + seed % 50, height=30.0)
        st = api.default_settings(960, 540, 32, 8, seed=1000 + seed, grid_hint=16 + seed % 24)
    elif kind == 2:
        world, objs = scenes.mixed_scene(seed=seed, n=20 + (seed * 7) % 101, spread=8.0 + seed % 13)
        st = api.default_settings(960, 540, 16, 8 + seed % 9, seed=1000 + seed, grid_hint=6 + seed % 7, brute_force_below=10)
    elif kind == 5:
        # crowded: a hundred entities in a few cells, so that lists are long, descriptors and pool run out and segments end early
        world, objs = scenes.mixed_scene(seed=seed, n=80 + seed % 60, rough_share=0.0, spread=2.5 + seed % 3)
        st = api.default_settings(960, 540, 32, 8 + seed % 9, seed=1000 + seed, grid_hint=2 + seed % 3, brute_force_below=10)
    elif kind == 4:
        world, objs = scenes.mixed_scene(seed=seed, n=2 + seed % 7, rough_share=0.0, spread=6.0 + seed % 5)     # below the threshold: the brute-force cast
        st = api.default_settings(960, 540, 32, 8 + seed % 9, seed=1000 + seed, grid_hint=6 + seed % 7, brute_force_below=10)
    else:
        world, objs = scenes.planes_grid_scene(seed=seed, n=20 + (seed * 5) % 90)        # three planes among every kind and material
        st = api.default_settings(960, 540, 16, 8 + seed % 5, seed=1000 + seed, grid_hint=6 + seed % 7, brute_force_below=10)
    if os.environ.get("PT_SOAK_VARY"):
        # the settings varied as well: ragged image sizes, sample counts that are no power of two, explicit sample ranges,
        # a wider blue-noise tile, depth of field
        st.width, st.height = 960 - (seed * 13) % 40, 540 - (seed * 7) % 30
        st.spp_wanted = (8, 12, 16, 27, 32, 48)[seed % 6]
        st.sample_ranges = (0, 1, 2, 4, 8)[(seed // 6) % 5]
        st.sample_grid_width = (3, 3, 4, 2)[(seed // 5) % 4]
        if seed % 3 == 1:
            world.aperture = 0.3 + (seed % 7) * 0.1
            world.focal_distance = 20.0 + seed % 15
    t0 = time.time()
    img = gpu.render(world, objs, st)
    sc = OracleScene(world, objs, st)
    ref = sc.render(policy=3, threads=16)
    paths = sc.path_count()
    sc.close()
    ok = gpu.stats().paths == paths
    fin = np.isfinite(ref).all(axis=2)
    ok = ok and bool((np.isfinite(img).all(axis=2) == fin).all())
    rough = any(o.roughness != 0.0 for o in objs)
    differ = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2) & fin
    note = ""
    if rough:
        e = np.abs(img[fin].astype(np.float64) - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-3)
        beyond = int((e > 1e-4).sum())
        # a phantom hit changes a whole path: such pixels are beyond any radiance tolerance, and are looked at like the exact scenes'
        far = (np.abs(img.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-3) > 1e-4).any(axis=2) & fin
        pixels = [(int(x), int(y)) for y, x in zip(*np.nonzero(far))]
        note = f"rough: {beyond} values beyond 1e-4, largest {e.max():.2e}"
    else:
        pixels = [(int(x), int(y)) for y, x in zip(*np.nonzero(differ))]
    explained = 0
    if pixels:
        say = (lambda m: print(m[:420], flush=True)) if os.environ.get("PT_SOAK_SAY") else None
        for p, hits in zip(pixels[:40], turned_away(world, objs, st, pixels[:40], say=say)):
            cone_only = len(hits) > 0 and all(t == api.PT_GEOM_CONE and abs(s) < 1e-6 for _, t, s in hits)
            explained += 1 if cone_only else 0
            if say or not cone_only:
                print("   pixel", p, "explained" if cone_only else "UNEXPLAINED", hits, flush=True)
        ok = ok and explained == min(len(pixels), 40) and bool(os.environ.get("PT_FAST_CONES"))
    total_pixels += fin.size
    total_differ += len(pixels)
    bad += 0 if ok else 1
    print(f"seed {seed} kind {kind} entities {len(objs)} paths {paths} : {len(pixels)} pixels off ({explained} phantom cone hits) {note} "
          f"{'ok' if ok else 'FAILED'} [{time.time() - t0:.1f} s]", flush=True)
print(f"{count} scenes, {total_pixels} pixels, {total_differ} off (all phantom cone hits: {bad == 0}); failed scenes: {bad}", flush=True)
sys.exit(1 if bad else 0)
