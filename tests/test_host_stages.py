"""CPU checks of the product's host side (no GPU): the C-ABI library loads and exports every symbol
include/pt_api.h declares, argument validation works, and the host stages (flatten, camera/world
constants, tables + blue noise, grid CSR) equal the CPU oracle bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle.oraclebind import OracleScene
from pathtracer_amd import api, scenes
from pathtracer_amd.api import default_settings, fptr, iptr, object_array, uptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_exports_every_declared_symbol(core_lib):
    header = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", header)
    assert len(set(names)) >= 20
    for n in sorted(set(names)):
        assert hasattr(core_lib, n), f"libptcore.so does not export {n}"


def test_fails_loudly_without_gpu(core_lib):
    """No CPU fallback: without a HIP device a context cannot even be created."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    ctx = core_lib.pt_create(0)
    assert not ctx
    assert b"HIP device" in core_lib.pt_last_error(None) or b"device" in core_lib.pt_last_error(None)


def test_default_scene_matches_python_generator(core_lib):
    w = api.pt_world()
    o = (api.pt_object * 2)()
    n = C.c_int(0)
    assert core_lib.pt_default_scene(C.byref(w), o, 2, C.byref(n)) == 0 and n.value == 2
    pw, po = scenes.default_scene()
    assert bytes(w) == bytes(pw)
    assert bytes(o) == bytes(object_array(po))
    s = api.pt_settings()
    core_lib.pt_default_settings(C.byref(s))
    assert (s.width, s.height, s.spp_wanted, s.sample_grid_width, s.max_bounces, s.grid_hint, s.brute_force_below) == (768, 432, 100, 3, 10, 10, 10)


CASES = {
    "default": lambda: scenes.default_scene(),
    "mixed40": lambda: scenes.mixed_scene(seed=7, n=40),
    "glass": lambda: scenes.glass_scene(seed=3, per_type=16),
    "crowded": lambda: scenes.mixed_scene(seed=9, n=300, spread=12.0),
    "dense2k": lambda: scenes.dense_scene(n=2000, seed=4, half_extent=40.0, height=20.0),
    "planes_small": lambda: scenes.planes_small_scene(),
    "planes_grid": lambda: scenes.planes_grid_scene(),
}


@pytest.mark.parametrize("name", list(CASES))
def test_flatten_world_grid_equal_oracle(core_lib, oracle_lib, name):
    world, objs = CASES[name]()
    hint = 24 if name == "dense2k" else 10
    st = default_settings(160, 90, 16, 8, seed=99, grid_hint=hint)
    sc = OracleScene(world, objs, st)
    n = len(objs)
    rec = np.zeros((n, 36), np.float32)
    assert core_lib.pt_host_flatten(object_array(objs), n, fptr(rec)) == 0
    assert (bits(rec) == bits(sc.records())).all()
    wc = np.zeros(27, np.float32)
    assert core_lib.pt_host_world(C.byref(world), st.width, st.height, fptr(wc)) == 0
    assert (bits(wc) == bits(sc.world_consts())).all()
    g, start, items = sc.grid()
    gi = np.zeros(9, np.int32)
    ct, it = C.c_int(0), C.c_int(0)
    assert core_lib.pt_host_grid(fptr(rec), n, hint, iptr(gi), None, 0, None, 0, C.byref(ct), C.byref(it)) == 0
    assert (gi == g).all() and ct.value == len(start) - 1 and it.value == len(items)
    cs = np.zeros(ct.value + 1, np.uint32)
    ci = np.zeros(max(1, it.value), np.uint32)
    assert core_lib.pt_host_grid(fptr(rec), n, hint, iptr(gi), uptr(cs), cs.size, uptr(ci), ci.size, None, None) == 0
    assert (cs == start).all() and (ci[:it.value] == items).all()
    sc.close()


@pytest.mark.parametrize("spp,gw,seed", [(1, 3, 1), (4, 3, 2), (16, 3, 12345), (64, 3, 12345), (100, 3, 7), (256, 3, 3), (16, 2, 5), (16, 4, 6)])
def test_tables_and_samples_equal_oracle(core_lib, oracle_lib, spp, gw, seed):
    world, objs = scenes.default_scene()
    st = default_settings(64, 36, spp, 4, seed=seed, sample_grid_width=gw)
    sc = OracleScene(world, objs, st)
    od, oc = sc.tables()
    osmp, ooff = sc.samples()
    d = np.zeros((20000, 3), np.float32)
    c = np.zeros((20000, 2), np.float32)
    smp = np.zeros((max(1, osmp.shape[0]) + 8, 2), np.float32)
    off = np.zeros(gw * gw + 1, np.int32)
    cnt = C.c_int(0)
    assert core_lib.pt_host_tables(seed, spp, gw, fptr(d), fptr(c), fptr(smp), smp.shape[0], iptr(off), C.byref(cnt)) == 0
    assert cnt.value == osmp.shape[0] and (off == ooff).all()
    assert (bits(d) == bits(od)).all() and (bits(c) == bits(oc)).all()
    assert (bits(smp[:cnt.value]) == bits(osmp)).all()
    # table invariants the tracer relies on
    ln = np.sqrt((d.astype(np.float64) ** 2).sum(1))
    assert np.abs(ln - 1).max() < 1e-6
    assert (np.sqrt((c.astype(np.float64) ** 2).sum(1)) <= 0.5 + 1e-7).all()
    assert (smp[:cnt.value] >= 0).all() and (smp[:cnt.value] < 1).all()
    sc.close()


def test_host_stage_argument_errors(core_lib):
    assert core_lib.pt_host_flatten(None, 3, None) != 0
    assert core_lib.pt_host_tables(1, 0, 3, None, None, None, 0, None, None) != 0
    assert core_lib.pt_host_tables(1, 16, 9, None, None, None, 0, None, None) != 0
    d = np.zeros(9, np.int32)
    assert core_lib.pt_host_grid(None, 0, 10, iptr(d), None, 0, None, 0, None, None) != 0


def test_sample_ranges_rule(core_lib):
    """pt_sample_ranges: the asked count, or from spp_wanted; always 1 for progressive frames."""
    import ctypes as C
    from pathtracer_amd.api import default_settings
    got = {}
    for spp in (1, 16, 31, 32, 64, 127, 128, 256, 4096):
        st = default_settings(64, 36, spp, 8)
        got[spp] = core_lib.pt_sample_ranges(C.byref(st))
    assert got == {1: 1, 16: 1, 31: 1, 32: 2, 64: 4, 127: 4, 128: 8, 256: 16, 4096: 16}
    st = default_settings(64, 36, 64, 8, sample_ranges=16)
    assert core_lib.pt_sample_ranges(C.byref(st)) == 16
    st.frame_index = 3
    assert core_lib.pt_sample_ranges(C.byref(st)) == 1
    for bad in (3, 32, -1):
        st = default_settings(64, 36, 64, 8, sample_ranges=bad)
        assert core_lib.pt_sample_ranges(C.byref(st)) < 0


def test_bench_workloads_and_cpu_leg():
    """Every named benchmark workload builds its scene and settings, and the CPU leg's small case runs."""
    import bench
    from pathtracer_amd.api import default_settings
    for name in ("demo-1080p-64spp-d8", "demo-640x360-16spp-d8", "glass-1080p-128spp-d32", "glass-1080p-16spp-d32",
                 "dense10k-1080p-16spp-d8", "mixed9-1080p-64spp-d16", "demo-1080p-progressive", "demo-320x180-4spp-d2"):
        world, objs, kw = bench.workload(name)
        kw = dict(kw)
        st = default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
        assert st.width > 0 and len(objs) >= 2
    world, objs, _ = bench.workload("demo-640x360-16spp-d8")
    row = bench.cpu_c1(world, objs, 2)
    assert row["paths"] > 0 and row["seconds_1_threads"] > 0 and row["mpaths_per_s_2_threads"] > 0
    # the roofline block of the driver's line: achieved rate against the vector peak, and the measured bytes against the
    # twelve bytes a pixel the path has to write
    from pathtracer_amd.api import pt_stats
    k = pt_stats()
    k.pixels, k.trace_ms_min, k.trace_ms_max = 1920 * 1080, 1.6, 1.7
    roof = bench.roofline_block("demo-1080p-64spp-d8", {"kstats": k, "kernel_ms": 1.65, "paths": 139622400}, 1)
    assert roof["bound"] == "valu" and roof["peak"] == bench.PEAK_FP32_VALU_TFLOPS and 0 < roof["frac"] < 1
    assert roof["algorithmic_bytes"] == 12 * 1920 * 1080
    if roof["traffic"] is not None:
        assert roof["traffic_ratio"] == pytest.approx(roof["traffic"] / roof["algorithmic_bytes"])
    assert callable(bench.progressive_block)
    counters = {k: 1 for k in list(bench.FLOP_WEIGHTS) + ["uneven_sphere"]}
    assert bench.flops_from_counters(counters) == sum(v for k, v in bench.FLOP_WEIGHTS.items() if k != "unit_space") + 3 * bench.FLOP_WEIGHTS["unit_space"]


@pytest.mark.parametrize("name,tolerance", [("demo-1080p-64spp-d8", 0.01), ("glass-1080p-128spp-d32", 0.01), ("mixed9-1080p-64spp-d16", 0.02)])
def test_known_flops_per_path(name, tolerance):
    """The flop-per-path figures bench.py quotes when its CPU leg is skipped: event counters of the oracle times the
    per-event weights, here on the same scene and samples per pixel at a small image size."""
    import bench
    world, objs, kw = bench.workload(name)
    got, counters = bench.oracle_flops_per_path(world, objs, kw, 4e5)
    assert counters["paths"] > 1e5
    assert got == pytest.approx(bench.KNOWN_FLOPS_PER_PATH[name], rel=tolerance)


def test_host_stages_under_sanitizers(tmp_path):
    """The library's host stages and scene-file reader compiled for the host alone with -fsanitize=address,undefined and
    driven by tests/native/host_san_driver.cpp (300 entities, every sample count and tile width, four grid hints)."""
    import subprocess
    exe = str(tmp_path / "host_san")
    src = [os.path.join(ROOT, "tests", "native", "host_san_driver.cpp"), os.path.join(ROOT, "pathtracer_amd", "csrc", "pt_host.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pathtracer_amd", "csrc")] + src + ["-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    assert r.stdout.startswith("checksum")


def test_write_through_store_keeps_its_wait_states():
    """The one inline-assembly statement of the library is a 16-byte store the compiler cannot see into: the two wait
    states that must separate it from an instruction overwriting its data registers have to be part of the statement."""
    import re
    src = open(os.path.join(ROOT, "pathtracer_amd", "csrc", "pt_kernels.hip")).read()
    stores = re.findall(r'asm volatile\("(global_store[^"]*)"', src)
    assert stores, "the write-through store is gone: drop this test with it"
    for s in stores:
        assert re.search(r"global_store_dwordx4 [^\\]*\\n\\ts_nop 1$", s), s
    # nothing else in the device sources issues memory instructions behind the compiler's back
    for name in os.listdir(os.path.join(ROOT, "pathtracer_amd", "csrc")):
        text = open(os.path.join(ROOT, "pathtracer_amd", "csrc", name)).read()
        for body in re.findall(r'asm volatile\("([^"]*)"', text):
            assert not re.search(r"\b(global|flat|buffer|scratch|ds)_", body) or body in stores, (name, body)


def test_list_entry_keeps_the_id_and_still_encloses_the_entity(core_lib):
    """The cooperative cast's 16-byte list entry carries the entity's id in bits of the bounding sphere's centre.
    Whatever the id, the sphere read back must enclose the entity's own (the filter built on it may pass too much,
    never too little), and kind, neighbour bits and id must come back unchanged."""
    rng = np.random.default_rng(11)
    for _ in range(4000):
        scale = 10.0 ** rng.uniform(-3, 5)
        centre = (rng.uniform(-1, 1, 3) * scale).astype(np.float32)
        if rng.random() < 0.1:
            centre[rng.integers(0, 3)] = 0.0
        radius = np.float32(10.0 ** rng.uniform(-3, 3))
        ident = int(rng.integers(0, 1 << 23)) if rng.random() < 0.8 else (1 << 23) - 1
        tag = int(rng.integers(0, 512))
        entry = np.zeros(4, np.uint32)
        back = C.c_uint32(0)
        assert core_lib.pt_host_list_entry(fptr(centre), float(radius), tag, ident, uptr(entry), C.byref(back)) == 0
        assert back.value == ident
        assert int(entry[3]) & 511 == tag
        got = entry[:3].view(np.float32).astype(np.float64)
        got_radius = float(np.array([int(entry[3]) & 0xffff0000], np.uint32).view(np.float32)[0])
        moved = float(np.sqrt(((got - centre.astype(np.float64)) ** 2).sum()))
        assert got_radius >= float(radius) + moved, (centre, radius, ident, got, got_radius)
        assert got_radius <= float(radius) * 1.01 + 130 * float(np.abs(centre).max()) * 2.0 ** -23 + 1e-36
    bad = np.zeros(4, np.uint32)
    assert core_lib.pt_host_list_entry(fptr(np.zeros(3, np.float32)), 1.0, 0, 1 << 23, uptr(bad), None) != 0


def test_cell_order_puts_the_thinnest_cells_first(core_lib):
    def order(count, size):
        out = np.zeros(3, np.int32)
        assert core_lib.pt_host_cell_order(iptr(np.array(count, np.int32)), iptr(np.array(size, np.int32)), iptr(out)) == 0
        return list(out)
    assert order((5, 7, 9), (3, 3, 3)) == [1, 5, 35]            # cubes: the reference's order, x fastest
    assert order((68, 68, 104), (6, 6, 1)) == [104, 104 * 68, 1]
    assert order((4, 5, 6), (2, 1, 3)) == [5, 1, 20]
    assert order((4, 5, 6), (2, 2, 1)) == [6, 24, 1]


def _quat_rotate(q, v):
    """rot * v (Quat * Vec3, code/rt_math.cpp:2902-2907), in double precision."""
    w, u = float(q[0]), np.array([q[1], q[2], q[3]], np.float64)
    t = 2 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def _surface_points(o, rng, count):
    """Points of an entity's surface in the world (double precision): its extreme points and random ones."""
    half = np.array(o.dim[:], np.float64) / 2
    pos = np.array(o.pos[:], np.float64)
    if o.type == api.PT_GEOM_BOX:
        local = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64) * half
        local = np.vstack([local, rng.uniform(-1, 1, (count, 3)) * half])
    elif o.type == api.PT_GEOM_SPHERE:
        d = rng.normal(size=(count, 3))
        local = d / np.linalg.norm(d, axis=1, keepdims=True) * half
        local = np.vstack([local, np.diag(half), -np.diag(half)])
    else:
        ang = rng.uniform(0, 2 * np.pi, count)
        ring = np.stack([np.cos(ang) * half[0], np.sin(ang) * half[1]], axis=1)
        if o.type == api.PT_GEOM_CYLINDER:
            z = np.where(rng.integers(0, 2, count) == 0, -half[2], half[2])
            local = np.column_stack([ring, z])
        else:       # cone: apex at +dz/2, base of semi-axes dx/2, dy/2 at -dz/2 (code/raycast.cpp:352-388 in unit space)
            t = rng.uniform(0, 1, count)[:, None]
            local = np.vstack([np.column_stack([ring * t, half[2] - 2 * half[2] * t[:, 0]]),
                               np.column_stack([ring, np.full(count, -half[2])]), [[0, 0, half[2]]]])
    return pos + np.array([_quat_rotate(o.rot, p) for p in local])


def test_filter_sphere_encloses_the_entity(core_lib):
    """The sphere the cooperative cast's filter knows of an entity holds every point of the entity's surface - for a cone it
    is the smallest such sphere, through the apex and the base's rim, about a point below the entity's position."""
    rng = np.random.default_rng(17)
    world, objs = scenes.mixed_scene(seed=31, n=60)
    _, dense = scenes.dense_scene(n=400, seed=5, half_extent=40.0, height=20.0)
    objs = objs + dense
    rec = np.zeros((len(objs), 36), np.float32)
    assert core_lib.pt_host_flatten(object_array(objs), len(objs), fptr(rec)) == 0
    tighter = 0
    for i, o in enumerate(objs):
        out = np.zeros(4, np.float32)
        assert core_lib.pt_host_filter_sphere(fptr(rec[i]), fptr(out)) == 0
        centre, radius = out[:3].astype(np.float64), float(out[3])
        pts = _surface_points(o, rng, 64)
        far = np.sqrt(((pts - centre) ** 2).sum(axis=1)).max()
        assert far <= radius * (1 + 1e-6) + 1e-5, (i, o.type, far, radius)
        assert radius <= far * 1.05 + 1e-3, (i, o.type, far, radius)          # and not much more (the sampled rim points miss the far end by a little)
        if o.type == api.PT_GEOM_CONE:
            half = np.array(o.dim[:], np.float64) / 2
            tighter += radius < np.sqrt(max(half[0], half[1]) ** 2 + half[2] ** 2) * 0.999
    assert tighter > 20
    plane = api.make_object((0, 0, 0), (1, 1, 1), api.PT_GEOM_PLANE)
    prec = np.zeros((1, 36), np.float32)
    assert core_lib.pt_host_flatten(object_array([plane]), 1, fptr(prec)) == 0
    out = np.zeros(4, np.float32)
    assert core_lib.pt_host_filter_sphere(fptr(prec[0]), fptr(out)) == 0 and np.isinf(out[3])
    assert core_lib.pt_host_filter_sphere(None, fptr(out)) != 0


def test_cone_slopes_vanish_on_the_generators(core_lib):
    """The direction test of the cone-exact trace variants: the form built from pt_host_cone_slopes is the reference's
    a = dx^2 + dy^2 - dz^2 of the normalised unit-space direction (code/raycast.cpp:352-388) for any direction, zero for the
    directions of the cone's generators; cones of extreme proportions get no data."""
    rng = np.random.default_rng(23)
    for k in range(200):
        q = rng.normal(size=4)
        q = q / np.linalg.norm(q) if k % 4 else np.array([1.0, 0, 0, 0])
        dim = rng.uniform(0.4, 5.0, 3)
        if k % 5 == 0:
            dim[1] = dim[0]
        o = api.make_object(tuple(rng.uniform(-20, 20, 3)), tuple(dim), api.PT_GEOM_CONE, rot=tuple(float(v) for v in q))
        rec = np.zeros((1, 36), np.float32)
        assert core_lib.pt_host_flatten(object_array([o]), 1, fptr(rec)) == 0
        s = np.zeros(8, np.float32)
        assert core_lib.pt_host_cone_slopes(fptr(rec[0]), fptr(s)) == 0
        s = s.astype(np.float64)
        qf = np.array(o.rot[:], np.float64)
        conj = np.array([qf[0], -qf[1], -qf[2], -qf[3]])
        df = np.array(o.dim[:], np.float64)

        def form(d):
            px, pz = np.dot(d, s[0:3]) ** 2, np.dot(d, s[4:7]) ** 2
            a = s[3] * px + s[7] * (np.dot(d, d) - pz) - pz
            return a / (a + 2 * pz)

        def reference(d):
            u = _quat_rotate(conj, d) * np.array([2 / df[0], 2 / df[1], 1 / df[2]])
            u = u / np.linalg.norm(u)
            return u[0] ** 2 + u[1] ** 2 - u[2] ** 2

        for _ in range(8):
            d = rng.normal(size=3)
            d = d / np.linalg.norm(d)
            assert abs(form(d) - reference(d)) < 2e-5, (k, form(d), reference(d))
        ang = rng.uniform(0, 2 * np.pi)
        gen = _quat_rotate(qf, np.array([np.cos(ang) * df[0] / 2, np.sin(ang) * df[1] / 2, -df[2]]))      # apex to a point of the base's rim
        gen = gen / np.linalg.norm(gen)
        assert abs(form(gen)) < 2e-5 and abs(form(-gen)) < 2e-5
    needle = api.make_object((0, 0, 0), (0.05, 0.05, 6.0), api.PT_GEOM_CONE)
    rec = np.zeros((1, 36), np.float32)
    assert core_lib.pt_host_flatten(object_array([needle]), 1, fptr(rec)) == 0
    s = np.zeros(8, np.float32)
    assert core_lib.pt_host_cone_slopes(fptr(rec[0]), fptr(s)) == 0 and np.isnan(s).all()
    ball = api.make_object((0, 0, 0), (1, 1, 1), api.PT_GEOM_SPHERE)
    assert core_lib.pt_host_flatten(object_array([ball]), 1, fptr(rec)) == 0
    assert core_lib.pt_host_cone_slopes(fptr(rec[0]), fptr(s)) != 0
