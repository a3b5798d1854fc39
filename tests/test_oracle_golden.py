"""Replays the committed reference vectors (tests/golden/ref_vectors.npz, generated from the
reference's own code by tests/golden/make_golden.py) through the CPU oracle. Bit-exact throughout:
integer, index and fp32 words alike.  Needs neither the reference tree nor a GPU."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from oracle.oraclebind import FP, IP, UP, U8P, OracleScene
from pathtracer_amd.api import pt_object, pt_settings, pt_world

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz")
P = lambda a: a.ctypes.data_as(FP)  # noqa: E731


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def same_bits(a, b, what):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    bad = np.flatnonzero(a.ravel() != b.ravel())
    assert bad.size == 0, f"{what}: {bad.size}/{a.size} words differ, first {bad[:5]}"


def load_case(gold, name):
    blob = gold[name + "_objects"].tobytes()
    n = len(blob) // C.sizeof(pt_object)
    objs = [pt_object.from_buffer_copy(blob, i * C.sizeof(pt_object)) for i in range(n)]
    world = pt_world.from_buffer_copy(gold[name + "_world"].tobytes())
    st = pt_settings.from_buffer_copy(gold[name + "_settings"].tobytes())
    return world, objs, st


def test_pcg(gold, oracle_lib):
    for k in range(4):
        state, seq = (int(v) for v in gold[f"pcg{k}_seed"])
        oracle_lib.pto_pcg_seed(state, seq)
        u = np.zeros(16, np.uint32)
        oracle_lib.pto_pcg_u32(16, u.ctypes.data_as(UP))
        i = np.zeros(64, np.int32)
        oracle_lib.pto_pcg_int(0, 19999, 64, i.ctypes.data_as(IP))
        f = np.zeros(64, np.float32)
        oracle_lib.pto_pcg_float(0, 1, 0.001, 64, P(f))
        assert (u == gold[f"pcg{k}_u32"]).all() and (i == gold[f"pcg{k}_int"]).all()
        same_bits(f, gold[f"pcg{k}_float"], "randomFloatPCG")


def test_raw_intersectors(gold, oracle_lib):
    n = 1024
    o, d, sp, sr, bd = (np.ascontiguousarray(gold[k]) for k in ("isect_o", "isect_d", "isect_sp", "isect_sr", "isect_bd"))
    uo, ud = np.ascontiguousarray(gold["isect_uo"]), np.ascontiguousarray(gold["isect_ud"])
    for fn, args in [("line_sphere", (o, d, sp, sr)), ("line_box", (o, d, sp, bd)), ("line_cylinder", (uo, ud)),
                     ("line_cone", (uo, ud))]:
        dist, pos, nrm = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        getattr(oracle_lib, "pto_" + fn)(n, *[P(a) for a in args], P(dist), P(pos), P(nrm))
        same_bits(dist, gold[fn + "_dist"], fn)
        hit = dist != -1
        assert hit.sum() > n // 4
        same_bits(pos[hit], gold[fn + "_pos"][hit], fn + " pos")
        same_bits(nrm[hit], gold[fn + "_nrm"][hit], fn + " normal")
    coll = np.zeros(n, np.int32)
    oracle_lib.pto_sphere_collision(n, P(o), P(d), P(sp), P(sr), coll.ctypes.data_as(IP))
    assert (coll == gold["sphere_collision"]).all()
    nn, ior = np.ascontiguousarray(gold["optics_n"]), np.ascontiguousarray(gold["optics_ior"])
    kr, rf, rl = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    oracle_lib.pto_optics(n, P(d), P(nn), P(ior), P(kr), P(rf), P(rl))
    same_bits(kr, gold["optics_kr"], "fresnel")
    same_bits(rf, gold["optics_refract"], "refract")
    same_bits(rl, gold["optics_reflect"], "reflect")


@pytest.mark.parametrize("name", ["default", "mixed40", "glass", "planes_small", "planes_grid"])
def test_scene_stages(gold, oracle_lib, name):
    world, objs, st = load_case(gold, name)
    sc = OracleScene(world, objs, st)
    dirs, disc = sc.tables()
    same_bits(dirs[:64], gold[name + "_dirs_head"], "direction table head")
    same_bits(disc[:64], gold[name + "_disc_head"], "disc table head")
    sha = hashlib.sha256(dirs.tobytes() + disc.tobytes()).digest()
    assert sha == gold[name + "_tables_sha"].tobytes()
    smp, offs = sc.samples()
    assert (offs == gold[name + "_offsets"]).all()
    same_bits(smp, gold[name + "_samples"], "blue-noise samples")
    same_bits(sc.records(), gold[name + "_records"], "precalc records")
    same_bits(sc.world_consts(), gold[name + "_worldconsts"], "camera/world constants")
    g, start, items = sc.grid()
    assert (g == gold[name + "_grid"]).all()
    assert (start == gold[name + "_cell_start"]).all() and (items == gold[name + "_cell_items"]).all()

    ro, rd = np.ascontiguousarray(gold[name + "_ray_o"]), np.ascontiguousarray(gold[name + "_ray_d"])
    nr = ro.shape[0]
    oi = np.ascontiguousarray(gold[name + "_gi_obj"])
    dist, pos, nrm = np.zeros(nr, np.float32), np.zeros((nr, 3), np.float32), np.zeros((nr, 3), np.float32)
    oracle_lib.pto_get_intersection(sc.h, nr, oi.ctypes.data_as(IP), P(ro), P(rd), P(dist), P(pos), P(nrm))
    same_bits(dist, gold[name + "_gi_dist"], "getIntersection")
    hit = dist != -1
    same_bits(pos[hit], gold[name + "_gi_pos"][hit], "getIntersection pos")
    same_bits(nrm[hit], gold[name + "_gi_nrm"][hit], "getIntersection normal")

    cells = np.zeros((512, 3), np.int32)
    ov = C.c_int32(0)
    for i in range(nr):
        k = oracle_lib.pto_traverse(sc.h, P(ro[i]), P(rd[i]), 512, cells.ctypes.data_as(IP), C.byref(ov))
        assert k == gold[name + "_dda_count"][i] and ov.value == gold[name + "_dda_outside"][i]
        h = 1469598103934665603
        for c in cells[:k]:
            idx = int(c[2]) * int(g[4]) * int(g[3]) + int(c[1]) * int(g[3]) + int(c[0])
            h = ((h ^ ((idx + 1) & 0xFFFFFFFF)) * 1099511628211) & ((1 << 64) - 1)
        assert h == int(gold[name + "_dda_hash"][i]), i
    for use_grid in (0, 1):
        ids = np.zeros(nr, np.int32)
        oracle_lib.pto_cast(sc.h, nr, P(ro), P(rd), use_grid, ids.ctypes.data_as(IP), P(pos), P(nrm))
        assert (ids == gold[name + f"_cast{use_grid}_id"]).all()
        h = ids != 0
        same_bits(pos[h], gold[name + f"_cast{use_grid}_pos"][h], "cast pos")
        same_bits(nrm[h], gold[name + f"_cast{use_grid}_nrm"][h], "cast normal")
    sc.close()


@pytest.mark.parametrize("name", ["default", "mixed40", "glass", "planes_small", "planes_grid"])
def test_full_image(gold, oracle_lib, name):
    """Whole pipeline under the per-job seed policy equals the reference image word for word."""
    world, objs, st = load_case(gold, name)
    sc = OracleScene(world, objs, st)
    img = sc.render(policy=1, seed_base=st.seed)
    same_bits(img, gold[name + "_image"], name + " image")
    same_bits(sc.render(policy=1, seed_base=st.seed, threads=3), img, "pthread scheduler")
    if world.aperture == 0:
        ids, _ = sc.primary()
        assert (ids == gold[name + "_primary"].astype(np.int32)).all()
    sc.close()


def test_srgb8(gold, oracle_lib):
    v = np.ascontiguousarray(gold["srgb_in"])
    o = np.zeros(v.size, np.uint8)
    oracle_lib.pto_srgb8(v.size, P(v), o.ctypes.data_as(U8P))
    assert (o == gold["srgb_out"]).all()


def test_seed_policies_agree_statistically(oracle_lib):
    """The per-sample policy (the one the device path uses) draws the same distribution as the
    per-job policy pinned above: high-spp images agree in the mean."""
    from pathtracer_amd import scenes
    from pathtracer_amd.api import default_settings
    world, objs = scenes.default_scene()
    st = default_settings(96, 54, 256, 8, seed=5)
    sc = OracleScene(world, objs, st)
    a = sc.render(policy=1, seed_base=5, threads=4)
    b = sc.render(policy=3, threads=4)
    mae = np.abs(a - b).mean()
    assert mae < 4e-3, mae
    mse = ((a - b) ** 2).mean()
    assert 10 * np.log10(1.0 / mse) > 38
    # pinhole primary rays do not depend on the policy: rows that see only sky are identical
    assert (a[:8] == b[:8]).all() and (a[:8] == 1.0).all()
    # the per-sample image does not depend on the thread count
    assert (sc.render(policy=3, threads=1) == b).all()
    sc.close()


def test_zero_sample_tile_is_black_not_nan(oracle_lib):
    """spp 1 can leave a blue-noise tile empty; the reference divides 0/0 there (code/raycast.cpp:1270)."""
    from pathtracer_amd import scenes
    from pathtracer_amd.api import default_settings
    world, objs = scenes.default_scene()
    found = False
    for seed in range(40):
        st = default_settings(30, 30, 1, 4, seed=seed)
        sc = OracleScene(world, objs, st)
        _, offs = sc.samples()
        if (np.diff(offs) == 0).any():
            img = sc.render(policy=3)
            assert np.isfinite(img).all()
            tile = int(np.flatnonzero(np.diff(offs) == 0)[0])
            assert (img[tile // 3::3, tile % 3::3] == 0).all()
            found = True
            sc.close()
            break
        sc.close()
    assert found


def test_range_sums_only_move_rounding(oracle_lib):
    """Per-sample policy: 1 range is the reference's single sequence of additions; more ranges change
    the image by rounding only, and progressive accumulation (one sample per frame) is unaffected."""
    from oracle.oraclebind import OracleScene
    from pathtracer_amd import scenes
    from pathtracer_amd.api import default_settings
    world, objs = scenes.default_scene()
    imgs = {}
    for ranges in (1, 2, 8):
        sc = OracleScene(world, objs, default_settings(48, 27, 64, 8, seed=99, sample_ranges=ranges))
        imgs[ranges] = sc.render(policy=3, threads=4)
        sc.close()
    assert np.abs(imgs[2].astype(np.float64) - imgs[1]).max() <= 4e-6
    assert np.abs(imgs[8].astype(np.float64) - imgs[1]).max() <= 4e-6
    assert (imgs[2] != imgs[1]).any() or (imgs[8] != imgs[1]).any()


def test_oracle_under_sanitizers():
    """The CPU restatement, driven through its whole surface by oracle/san_driver.c (two scenes, both seed policies, the
    threaded renderer, casts, progressive frames), built with -fsanitize=address,undefined and with -fsanitize=thread:
    no report from either, and the same checksum from both."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "san"])
    sums = []
    for name in ("san_asan", "san_tsan"):
        r = subprocess.run([os.path.join(root, "oracle", "_san", name)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert not re.search(r"Sanitizer|runtime error", r.stderr), r.stderr[-2000:]
        sums.append(r.stdout.strip())
    assert sums[0] == sums[1] and sums[0].startswith("checksum")
