"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs and against the committed reference vectors.

Bars: entity ids, hit positions / normals and DDA cell-sequence hashes bit-exact (a mismatch
fraction above 1e-5 fails; none is expected); radiance within 1e-4 relative per pixel and per
sample (Oren-Nayar uses device acos/sin/tan, everything else is IEEE-exact and compares bitwise).
"""
import ctypes as C
import os
import time

import numpy as np
import pytest

from oracle.oraclebind import FP, IP, OracleScene
from pathtracer_amd import api, scenes
from pathtracer_amd.api import default_settings, make_object
from rayutil import scene_rays

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz")
P = lambda a: a.ctypes.data_as(FP)  # noqa: E731
REL_TOL = 1e-4          # north-star tolerance on radiance
EDGE_FRACTION = 1e-5    # allowed share of epsilon-edge mismatches on ids / cell hashes


@pytest.fixture(scope="module")
def gpu():
    r = api.Renderer(0)
    yield r
    r.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rel_err(a, b):
    return np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1e-3)


CASES = {
    # name: (factory, grid_hint, brute_force_below)
    "default": (lambda: scenes.default_scene(), 10, 10),
    "default_grid": (lambda: scenes.default_scene(), 10, 0),       # demo scene forced through the grid walk
    "mixed40": (lambda: scenes.mixed_scene(seed=7, n=40), 10, 10),
    "glass": (lambda: scenes.glass_scene(seed=3, per_type=16), 10, 10),
    "crowded": (lambda: scenes.mixed_scene(seed=9, n=300, spread=12.0), 10, 10),
    "dense2k": (lambda: scenes.dense_scene(n=2000, seed=4, half_extent=40.0, height=20.0), 24, 10),
    # brute-force scenes (fewer than 10 entities): every kind and material, and spheres / upright boxes only
    # (the lean kernel variant, like "default")
    "small_mixed": (lambda: scenes.mixed_scene(seed=11, n=8), 10, 10),
    "plain9": (lambda: scenes.plain_scene(seed=5, n=8), 10, 10),
    # upright boxes whose faces all lie on cell bounds (cells of 6 x 6 x 1): hit points on a cell's bound or an ulp beside it
    "aligned_boxes": (lambda: scenes.aligned_boxes_scene(), 14, 10),
    # the reference's fifth primitive, the infinite plane: tested by every ray that misses the grid, never by a bounding sphere,
    # and listed in whole layers of grid cells - in a brute-force scene, in a grid scene, and the small scene forced through the grid
    "planes_small": (lambda: scenes.planes_small_scene(), 10, 10),
    "planes_grid": (lambda: scenes.planes_grid_scene(), 10, 10),
    "planes_small_grid": (lambda: scenes.planes_small_scene(), 10, 0),
}
ROUGH_FREE = {"default", "default_grid", "plain9", "aligned_boxes", "planes_small", "planes_small_grid"}


def settings_for(name, w, h, spp, bounces, seed=4242, **kw):
    _, hint, brute = CASES[name]
    return default_settings(w, h, spp, bounces, seed=seed, grid_hint=hint, brute_force_below=brute, **kw)


@pytest.mark.parametrize("name", list(CASES))
def test_cast_rays_bit_exact(gpu, name):
    world, objs = CASES[name][0]()
    st = settings_for(name, 64, 36, 4, 4)
    gpu.prepare(world, objs, st)
    sc = OracleScene(world, objs, st)
    n = 40000
    o, d = scene_rays(n, 77, objs, spread={"dense2k": 60.0, "aligned_boxes": 60.0}.get(name, 30.0))
    if name == "aligned_boxes":
        # a quarter of the rays run along an axis onto the faces: their hit points lie exactly on cell bounds
        rng = np.random.default_rng(5)
        k = n // 4
        axis = rng.integers(0, 3, k)
        sign = rng.choice(np.array([-1.0, 1.0], np.float32), k)
        oo = rng.uniform(-45, 45, (k, 3)).astype(np.float32)
        oo[:, 2] = rng.uniform(0.25, 11.75, k).astype(np.float32)
        dd = np.zeros((k, 3), np.float32)
        dd[np.arange(k), axis] = sign
        oo[np.arange(k), axis] = -sign * np.float32(70.0)
        o[:k], d[:k] = oo, dd
    ids, pos, nrm = gpu.cast_rays(o, d)
    rid, rpos, rnrm = np.zeros(n, np.int32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    use_grid = 0 if len(objs) < st.brute_force_below else 1
    sc.lib.pto_cast(sc.h, n, P(o), P(d), use_grid, rid.ctypes.data_as(IP), P(rpos), P(rnrm))
    sc.close()
    assert (rid != 0).mean() > 0.25, "ray set must hit often"
    bad = ids != rid
    assert bad.mean() <= EDGE_FRACTION, f"{bad.sum()} of {n} entity ids differ"
    hit = (rid != 0) & ~bad
    assert (bits(pos[hit]) == bits(rpos[hit])).all(), "hit positions differ"
    assert (bits(nrm[hit]) == bits(rnrm[hit])).all(), "hit normals differ"
    if name == "aligned_boxes":
        h = rpos[hit].astype(np.float64)
        on_bound = (np.mod(h[:, 0], 6) == 0) | (np.mod(h[:, 1], 6) == 0) | (np.mod(h[:, 2], 1) == 0)
        assert on_bound.sum() > 5000, "the ray set must put many hit points exactly on cell bounds"


@pytest.mark.parametrize("seed", [21, 22, 23, 24, 25, 26])
def test_cast_rays_bit_exact_on_more_scenes(gpu, seed):
    """The same comparison on six more scenes of every kind (rotated and upright, crowded enough that a wavefront's chunk
    of candidates holds all kinds at once): ids, hit positions and normals of 30 000 rays each, bit for bit."""
    world, objs = scenes.mixed_scene(seed=seed, n=40 + 20 * (seed % 3), spread=10.0 + 2 * (seed % 4))
    st = default_settings(64, 36, 4, 4, seed=4242, grid_hint=8 + seed % 5, brute_force_below=10)
    gpu.prepare(world, objs, st)
    sc = OracleScene(world, objs, st)
    n = 30000
    o, d = scene_rays(n, seed, objs, spread=25.0)
    ids, pos, nrm = gpu.cast_rays(o, d)
    rid, rpos, rnrm = np.zeros(n, np.int32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    sc.lib.pto_cast(sc.h, n, P(o), P(d), 1, rid.ctypes.data_as(IP), P(rpos), P(rnrm))
    sc.close()
    assert (rid != 0).mean() > 0.25
    assert (ids == rid).all(), f"{(ids != rid).sum()} of {n} entity ids differ"
    hit = rid != 0
    assert (bits(pos[hit]) == bits(rpos[hit])).all() and (bits(nrm[hit]) == bits(rnrm[hit])).all()


@pytest.mark.parametrize("name", list(CASES))
def test_primary_ids_and_cell_hashes(gpu, name):
    world, objs = CASES[name][0]()
    st = settings_for(name, 320, 180, 16, 4)
    gpu.prepare(world, objs, st)
    ids, hs = gpu.debug_primary()
    sc = OracleScene(world, objs, st)
    rid, rhs = sc.primary()
    sc.close()
    assert (ids != rid).mean() <= EDGE_FRACTION
    assert (hs != rhs).mean() <= EDGE_FRACTION
    if len(objs) >= st.brute_force_below:
        assert (rhs != 0).any(), "grid path must produce cell hashes"


@pytest.mark.parametrize("name,w,h,spp,bounces", [("default", 320, 180, 16, 8), ("default_grid", 160, 90, 16, 8),
                                                   ("mixed40", 160, 90, 16, 8), ("glass", 160, 90, 16, 32),
                                                   ("crowded", 128, 72, 4, 8), ("dense2k", 128, 72, 4, 8),
                                                   ("small_mixed", 160, 90, 16, 16), ("plain9", 160, 90, 16, 8),
                                                   ("aligned_boxes", 160, 90, 16, 8), ("planes_small", 160, 90, 16, 8),
                                                   ("planes_grid", 160, 90, 16, 8), ("planes_small_grid", 96, 54, 16, 8)])
def test_image_parity(gpu, name, w, h, spp, bounces):
    world, objs = CASES[name][0]()
    st = settings_for(name, w, h, spp, bounces)
    img = gpu.render(world, objs, st)
    stats = gpu.stats()
    sc = OracleScene(world, objs, st)
    ref = sc.render(policy=3, threads=8)
    cnt = sc.counters()
    paths = sc.path_count()
    sc.close()
    assert np.isfinite(img).all() and img.min() >= 0 and img.max() <= 1
    assert stats.paths == paths
    assert stats.rays == cnt["casts"], "device ray-cast count differs from the oracle's"
    if name in ROUGH_FREE:
        assert (bits(img) == bits(ref)).all(), "image must be bit-identical where no libm call is involved"
    else:
        e = rel_err(img, ref)
        assert (e > REL_TOL).mean() <= EDGE_FRACTION, f"max rel err {e.max():.3e}, {(e > REL_TOL).sum()} values beyond tolerance"
    mse = ((img.astype(np.float64) - ref) ** 2).mean()
    assert mse < 1e-9


@pytest.mark.parametrize("name,w,h,spp,ranges", [("default", 160, 90, 64, 0), ("default", 160, 90, 64, 1), ("default", 160, 90, 64, 8),
                                                  ("default_grid", 96, 54, 40, 16), ("glass", 96, 54, 64, 4), ("default", 64, 36, 4, 16),
                                                  ("small_mixed", 160, 90, 32, 2)])
def test_sample_range_sums(gpu, name, w, h, spp, ranges):
    """A pixel's samples are added range by range and the range sums in range order: the device's work
    items are those ranges (different lanes, any order in time), the oracle adds in the same order, so
    the images agree as before; against one range (the reference's single sequence) only the rounding
    of the sum moves.  16 ranges over buckets of a few samples leaves some ranges empty."""
    world, objs = CASES[name][0]()
    st = settings_for(name, w, h, spp, 8, sample_ranges=ranges)
    img = gpu.render(world, objs, st)
    sc = OracleScene(world, objs, st)
    ref = sc.render(policy=3, threads=8)
    sc.close()
    if name in ROUGH_FREE:
        assert (bits(img) == bits(ref)).all()
    else:
        e = rel_err(img, ref)
        assert (e > REL_TOL).mean() <= EDGE_FRACTION, e.max()
    st1 = settings_for(name, w, h, spp, 8, sample_ranges=1)
    one = gpu.render(world, objs, st1)
    assert np.abs(img.astype(np.float64) - one).max() <= 4e-6, "range sums may differ from one sequence by rounding only"
    expect = api.load_core().pt_sample_ranges(C.byref(st))
    assert expect == (ranges if ranges else 4)


@pytest.mark.parametrize("name,w,h,spp,bounces", [("default", 160, 90, 16, 8), ("glass", 96, 54, 16, 32), ("mixed40", 96, 54, 16, 8)])
def test_cast_histogram(gpu, name, w, h, spp, bounces):
    """The harness's bounce histogram (pt_cast_histogram): how many rays each camera sample's path casts, against the
    oracle's count of the same paths; the bins add up to the frame's paths and, weighted, to its rays."""
    world, objs = CASES[name][0]()
    st = settings_for(name, w, h, spp, bounces)
    gpu.render(world, objs, st)
    stats = gpu.stats()
    hist = gpu.cast_histogram()
    sc = OracleScene(world, objs, st)
    sc.render(policy=3, threads=1)
    want = sc.cast_histogram()
    sc.close()
    assert hist[0] == 0 and int(hist.sum()) == stats.paths
    assert int((hist * np.arange(hist.size, dtype=np.uint64)).sum()) == stats.rays
    assert (hist[bounces + 1:] == 0).all()
    if name in ROUGH_FREE:
        assert (hist == want).all()
    else:       # Oren-Nayar's libm calls can move an attenuation to exactly zero or not: path lengths may differ for a few paths
        assert np.abs(hist.astype(np.int64) - want.astype(np.int64)).sum() <= max(2, int(EDGE_FRACTION * stats.paths))
    few = gpu.cast_histogram(bins=4)
    assert (few[:3] == hist[:3]).all() and int(few[3]) == int(hist[3:].sum())


def test_per_sample_radiance_via_progressive_frames(gpu):
    """Frame f of the progressive mode traces sample (f mod count) of every pixel, so the first frames
    expose single-sample radiance: each must match the oracle within 1e-4 relative."""
    world, objs = scenes.mixed_scene(seed=7, n=40)
    st = settings_for("mixed40", 96, 54, 16, 8, frame_index=0)
    gpu.prepare(world, objs, st)
    sc = OracleScene(world, objs, st)
    acc_ref = np.zeros((54, 96, 3), np.float32)
    prev = np.zeros_like(acc_ref)
    for f in range(20):
        gpu.set_frame(f)
        gpu.trace()
        shown = gpu.download()
        sc.lib.pto_progressive_frame(sc.h, f, P(acc_ref))
        want = np.minimum(acc_ref / np.float32(f + 1), np.float32(1.0))
        e = rel_err(shown, want)
        assert (e > REL_TOL).mean() <= EDGE_FRACTION, (f, e.max())
        single = acc_ref - prev          # clamped radiance of this frame's sample
        prev = acc_ref.copy()
        assert np.isfinite(single).all()
    sc.close()


def test_progressive_frames_at_full_size(gpu):
    """The interactive path at the headline size: six one-sample frames of the demo scene at 1920x1080, written straight to the
    image by the trace kernel; every pixel of every frame is the oracle's accumulated frame, bit for bit."""
    world, objs = scenes.default_scene()
    st = default_settings(1920, 1080, 64, 8, seed=12345, frame_index=0)
    gpu.prepare(world, objs, st)
    sc = OracleScene(world, objs, st)
    acc_ref = np.zeros((1080, 1920, 3), np.float32)
    for f in range(6):
        gpu.set_frame(f)
        gpu.trace()
        shown = gpu.download()
        sc.lib.pto_progressive_frame(sc.h, f, P(acc_ref))
        want = np.minimum(acc_ref / np.float32(f + 1), np.float32(1.0))
        assert (bits(shown) == bits(want)).all(), (f, int((bits(shown) != bits(want)).any(axis=2).sum()))
    sc.close()


def test_progressive_converges_to_full_render(gpu):
    world, objs = scenes.default_scene()
    st_full = settings_for("default", 96, 54, 16, 8)
    full = gpu.render(world, objs, st_full)
    st = settings_for("default", 96, 54, 16, 8, frame_index=0)
    gpu.prepare(world, objs, st)
    sc = OracleScene(world, objs, st_full)
    _, offs = sc.samples()
    sc.close()
    counts = np.diff(offs)
    frames = {}
    for f in range(int(counts.max())):
        gpu.set_frame(f)
        gpu.trace()
        if (f + 1) in counts:
            frames[f + 1] = gpu.download()
    # after count(tile) frames the pixels of that tile class equal the full render bit for bit
    for t, c in enumerate(counts):
        ty, tx = divmod(t, 3)
        a = frames[int(c)][ty::3, tx::3]
        b = full[ty::3, tx::3]
        assert (bits(a) == bits(b)).all(), t


def test_reference_golden_vectors_on_gpu(gpu):
    """Straight against the reference's own outputs (tests/golden): casts through the prepared scene
    and pinhole primary hits."""
    gold = np.load(GOLD)
    for name in ("default", "mixed40", "glass", "planes_small", "planes_grid"):
        blob = gold[name + "_objects"].tobytes()
        n = len(blob) // C.sizeof(api.pt_object)
        objs = [api.pt_object.from_buffer_copy(blob, i * C.sizeof(api.pt_object)) for i in range(n)]
        world = api.pt_world.from_buffer_copy(gold[name + "_world"].tobytes())
        st = api.pt_settings.from_buffer_copy(gold[name + "_settings"].tobytes())
        gpu.prepare(world, objs, st)
        use_grid = 0 if n < st.brute_force_below else 1
        ids, pos, nrm = gpu.cast_rays(gold[name + "_ray_o"], gold[name + "_ray_d"])
        assert (ids == gold[name + f"_cast{use_grid}_id"]).all()
        hit = ids != 0
        assert (bits(pos[hit]) == bits(gold[name + f"_cast{use_grid}_pos"][hit])).all()
        assert (bits(nrm[hit]) == bits(gold[name + f"_cast{use_grid}_nrm"][hit])).all()
        if world.aperture == 0:
            pid, _ = gpu.debug_primary()
            assert (pid == gold[name + "_primary"].astype(np.int32)).all()


def test_partition_invariance(gpu):
    """Tile partitions (the multi-GPU split) give the single-partition image bit for bit."""
    world, objs = scenes.mixed_scene(seed=7, n=40)
    st = settings_for("mixed40", 200, 117, 16, 8)       # not a multiple of the 8x8 tile
    whole = gpu.render(world, objs, st)
    for parts in (2, 3, 8):
        img = np.full_like(whole, -1)
        for p in range(parts):
            stp = settings_for("mixed40", 200, 117, 16, 8, part_index=p, part_count=parts)
            gpu.prepare(world, objs, stp)
            gpu.trace()
            gpu._check(gpu.lib.pt_download(gpu.ctx, api.fptr(img)), "pt_download")
        assert (bits(img) == bits(whole)).all(), parts


@pytest.mark.parametrize("name", ["default", "default_grid", "glass", "dense2k", "small_mixed"])
def test_both_store_variants_of_the_trace_kernel_agree(gpu, name):
    """An asynchronous render (pt_render_start) runs the kernel twin whose range sums go through the caches to memory, a
    resident-scene trace (pt_trace) the plain one: same image, bit for bit, for every cast variant."""
    world, objs = CASES[name][0]()
    st = settings_for(name, 168, 93, 16, 8)
    through = gpu.render(world, objs, st).copy()
    gpu.prepare(world, objs, st)
    gpu.trace()
    plain = gpu.download()
    assert (bits(through) == bits(plain)).all()


def test_assemble_on_device(gpu):
    """The multi-GPU data flow on one GPU: every partition traced into a caller-owned device buffer
    (as bench.py does with torch tensors) on a caller-owned stream, buffers side by side as after the
    gather, pt_assemble on the same stream."""
    import torch
    from pathtracer_amd import dist as ptdist
    world, objs = scenes.mixed_scene(seed=7, n=40)
    w, h = 200, 117
    whole = gpu.render(world, objs, settings_for("mixed40", w, h, 16, 8))
    for parts in (2, 4, 8):
        stride = ptdist.partition_stride(w, h, parts)
        gathered = torch.full((parts * stride,), -1.0, dtype=torch.float32, device="cuda")
        image = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        stream = torch.cuda.Stream()
        assert stream.cuda_stream != 0
        for p in range(parts):
            gpu.prepare(world, objs, settings_for("mixed40", w, h, 16, 8, part_index=p, part_count=parts))
            assert gpu.partition_floats() <= stride
            gpu.trace(d_rgb=gathered[p * stride:].data_ptr(), stream=stream.cuda_stream, sync=False)
            gpu.sync()                   # the next prepare re-uses the context's buffers
        gpu.assemble(gathered.data_ptr(), stride, parts, w, h, image.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        got = image.cpu().numpy().reshape(h, w, 3)
        assert (bits(got) == bits(whole)).all(), parts


def test_pipelined_partition_frames(gpu):
    """The double-buffered frame pipeline of the multi-GPU bench (trace on one stream, gather and assemble of
    the previous frame on another), run here with three partitions owned by one process.  The frames are
    progressive ones, so every frame's image differs from the one before and a gather that read a tile buffer
    too early, or a trace that overwrote it too soon, would show: each assembled frame must equal the same
    frame of an unpartitioned context bit for bit.  No host synchronisation between the frames."""
    torch = pytest.importorskip("torch")
    from pathtracer_amd import dist as ptdist
    world, objs = scenes.mixed_scene(seed=7, n=40)
    w, h, frames = 200, 120, 12
    gpu.prepare(world, objs, settings_for("mixed40", w, h, 8, 8, frame_index=0))
    want = []
    for f in range(frames):
        gpu.set_frame(f)
        gpu.trace()
        want.append(gpu.download())
    assert not (bits(want[3]) == bits(want[4])).all()
    parts = []
    for p in range(3):
        r = api.Renderer(0)
        r.prepare(world, objs, settings_for("mixed40", w, h, 8, 8, part_index=p, part_count=3, frame_index=0))
        parts.append(r)
    pipe = ptdist.FramePipeline(parts, [0, 1, 2], 3, w, h, keep_frames=frames)
    for f in range(frames):
        for r in parts:
            r.set_frame(f)
        pipe.step()
    last = pipe.finish().cpu().numpy().reshape(h, w, 3)
    kept = [k.cpu().numpy().reshape(h, w, 3) for k in pipe.kept]
    for r in parts:
        r.sync()
        r.close()
    for f in range(frames):
        assert (bits(kept[f]) == bits(want[f])).all(), f
    assert (bits(last) == bits(want[-1])).all()


@pytest.mark.parametrize("parts", [1, 3])
def test_group_resident_progressive_frames(gpu, parts):
    """pt_group_prepare / _set_frame / _trace / _download: the application's frame loop on several partitions, driven from C.
    Twelve progressive frames queued without a host synchronisation in between (ring of three tile and gathered buffers,
    copies on the partitions' streams, assemble on the root's side stream); every frame's image differs from the one
    before, so a copy that read a tile buffer too early or a trace that overwrote it too soon would show.  Frames
    downloaded on the way (after 1, 2, 5 and 12 frames) and the last one must equal the same frames of an unpartitioned context."""
    world, objs = scenes.mixed_scene(seed=7, n=40)
    w, h, frames = 200, 120, 12
    gpu.prepare(world, objs, settings_for("mixed40", w, h, 8, 8, frame_index=0))
    want = []
    for f in range(frames):
        gpu.set_frame(f)
        gpu.trace()
        want.append(gpu.download())
    assert not (bits(want[3]) == bits(want[4])).all()
    g = api.Group([0] * parts)
    with pytest.raises(api.PtError, match="before pt_group_prepare"):
        g.trace()
    g.prepare(world, objs, settings_for("mixed40", w, h, 8, 8, frame_index=0))
    for f in range(frames):
        g.set_frame(f)
        g.trace(sync=False)
        if f in (0, 1, 4):
            assert (bits(g.download()) == bits(want[f])).all(), f
    assert (bits(g.download()) == bits(want[frames - 1])).all()
    # a full render through the same group afterwards, then resident frames again: the two interfaces do not disturb each other
    st = settings_for("mixed40", w, h, 16, 8)
    whole = gpu.render(world, objs, st)
    assert (bits(g.render(world, objs, st)) == bits(whole)).all()
    g.prepare(world, objs, st)
    g.trace()
    assert (bits(g.download()) == bits(whole)).all()
    stats = [g.part_stats(p) for p in range(parts)]
    assert sum(s.paths for s in stats) == gpu.stats().paths
    g.close()


def test_group_on_two_devices_when_there_are_two(gpu):
    """The group path across different GPUs (peer copies over xGMI, peer access, per-device static data): runs only where the
    box has two devices, so that the first multi-GPU machine this suite meets pins it."""
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU here")
    world, objs = scenes.mixed_scene(seed=7, n=40)
    st = settings_for("mixed40", 200, 117, 16, 8)
    whole = gpu.render(world, objs, st)
    g = api.Group([0, 1])
    assert (bits(g.render(world, objs, st)) == bits(whole)).all()
    g.prepare(world, objs, st)
    for _ in range(4):
        g.trace(sync=False)
    assert (bits(g.download()) == bits(whole)).all()
    g.close()


@pytest.mark.parametrize("parts", [1, 2, 5])
def test_group_render_gathers_on_the_device(gpu, parts):
    """pt_group_*: one frame split over several contexts of one process (here all on the one GPU), partitions copied
    device to device into the root's gathered buffer, assembled there, downloaded once.  Bit-identical to the
    unpartitioned image, also for a ragged image size and on repeated use of the group; after an abort only finished
    pixels are written and the progress record counts them."""
    world, objs = scenes.mixed_scene(seed=7, n=40)
    group = api.Group([0] * parts)
    for w, h, spp in ((200, 117, 16), (333, 77, 8)):
        st = settings_for("mixed40", w, h, spp, 8)
        whole = gpu.render(world, objs, st)
        img = group.render(world, objs, st, out=np.full((h, w, 3), -3.0, np.float32))
        assert (bits(img) == bits(whole)).all()
        p = group.progress()
        assert p.done == 1 and p.aborted == 0 and p.fraction == 1.0 and p.paths_done == gpu.stats().paths
        assert sum(group.part_stats(k).paths for k in range(parts)) == gpu.stats().paths
    # abort: a long frame, stopped early
    world, objs = scenes.glass_scene(seed=3, per_type=16)
    st = settings_for("glass", 1280, 720, 1024, 32)         # a quarter of a second of tracing
    whole = gpu.render(world, objs, st)
    out = np.full((720, 1280, 3), -3.0, np.float32)
    group.start(world, objs, st, out=out)
    snap = np.full_like(out, -3.0)
    seen = 0
    t0 = time.time()
    while time.time() - t0 < 60:
        p = group.progress()
        if p.done or p.fraction > 0.2:
            break
        seen = group.snapshot(snap)              # every partition's finished pixels, scattered into the one image
    shown = (snap != -3.0).all(axis=2)
    assert int(shown.sum()) == seen and ((snap == -3.0).all(axis=2) | shown).all()
    assert (bits(snap[shown]) == bits(whole[shown])).all()
    group.abort()
    group.wait()
    p = group.progress()
    assert p.done == 1 and p.aborted == 1
    written = (out != -3.0).all(axis=2)
    assert ((out == -3.0).all(axis=2) | written).all() and 0 < written.sum() < 1280 * 720
    assert (bits(out[written]) == bits(whole[written])).all()
    assert p.fraction == pytest.approx(written.sum() / (1280 * 720), abs=1e-6)
    group.close()


def test_headless_harness(gpu, tmp_path):
    """The C++ harness drives start / progress / wait through the C ABI and writes the image."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "harness", "pt_headless")
    if not os.path.exists(exe):
        pytest.fail("harness/pt_headless is not built (python -c 'import __graft_entry__ as g; g.build()')")
    world, objs = scenes.mixed_scene(seed=7, n=40)
    scene_file = str(tmp_path / "mixed.scene")
    api.save_scene(scene_file, world, objs)
    pfm, ppm = str(tmp_path / "o.pfm"), str(tmp_path / "o.ppm")
    r = subprocess.run([exe, "--scene", scene_file, "--size", "160x90", "--spp", "16", "--bounces", "8", "--seed", "4242",
                        "--pfm", pfm, "--out", ppm, "--quiet"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    st = settings_for("mixed40", 160, 90, 16, 8)
    want = gpu.render(world, objs, st)
    assert info["paths"] == gpu.stats().paths and info["aborted"] == 0 and info["objects"] == len(objs)
    raw = open(pfm, "rb").read()
    got = np.frombuffer(raw[-160 * 90 * 12:], np.float32).reshape(90, 160, 3)[::-1]
    assert (bits(got) == bits(want)).all()
    body = open(ppm, "rb").read()
    assert body.startswith(b"P6") and len(body) >= 160 * 90 * 3
    # several partitions through the group calls (three contexts on the one GPU): the gathered and assembled image is the same
    r = subprocess.run([exe, "--scene", scene_file, "--size", "160x90", "--spp", "16", "--bounces", "8", "--seed", "4242",
                        "--gpus", "0,0,0", "--pfm", pfm, "--quiet"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["paths"] == gpu.stats().paths and info["gpus"] == 3 and info["aborted"] == 0
    raw = open(pfm, "rb").read()
    got = np.frombuffer(raw[-160 * 90 * 12:], np.float32).reshape(90, 160, 3)[::-1]
    assert (bits(got) == bits(want)).all()
    # the bounce histogram of the same frame: its bins add up to the paths and, weighted, to the rays
    r = subprocess.run([exe, "--scene", scene_file, "--size", "160x90", "--spp", "16", "--bounces", "8", "--seed", "4242",
                        "--gpus", "0,0", "--bounce-histogram", "--quiet"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    hist = info["casts_per_path"]
    assert sum(hist) == info["paths"] and sum((k + 1) * v for k, v in enumerate(hist)) == info["rays"] and len(hist) <= 8
    # the interactive mode on three partitions: twenty progressive frames from C, latency figures, and the accumulated image
    r = subprocess.run([exe, "--scene", scene_file, "--size", "160x90", "--spp", "16", "--bounces", "8", "--seed", "4242",
                        "--gpus", "0,0,0", "--progressive", "20", "--pfm", pfm, "--quiet"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["mode"] == "progressive" and info["frames"] == 20 and 0 < info["frame_ms_median"] <= info["frame_ms_p99"]
    gpu.prepare(world, objs, settings_for("mixed40", 160, 90, 16, 8, frame_index=0))
    for f in range(20):
        gpu.set_frame(f)
        gpu.trace()
    raw = open(pfm, "rb").read()
    got = np.frombuffer(raw[-160 * 90 * 12:], np.float32).reshape(90, 160, 3)[::-1]
    assert (bits(got) == bits(gpu.download())).all()
    # abort path: a long render stopped from the frame loop still ends cleanly
    r = subprocess.run([exe, "--scene", scene_file, "--size", "1920x1080", "--spp", "256", "--bounces", "32", "--abort-after", "0.02",
                        "--poll-ms", "2", "--quiet"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "aborted" in r.stdout


def test_determinism_and_range_at_full_size(gpu):
    """The headline size (1080p, 64 spp, depth 8): two runs are bitwise equal, the image is finite and in
    [0,1], primary ids equal the oracle's, sky rows are exactly the background colour."""
    world, objs = scenes.default_scene()
    st = default_settings(1920, 1080, 64, 8, seed=12345)
    gpu.prepare(world, objs, st)
    gpu.trace()
    a = gpu.download()
    gpu.trace()
    b = gpu.download()
    assert (bits(a) == bits(b)).all()
    assert np.isfinite(a).all() and a.min() >= 0 and a.max() <= 1
    assert (a[:100] == 1.0).all()
    ids, _ = gpu.debug_primary()
    sc = OracleScene(world, objs, st)
    rid, _ = sc.primary()
    paths = sc.path_count()
    # a 64-row band of the full-size image against the oracle, bit for bit
    band = np.zeros((1080, 1920, 3), np.float32)
    sc.lib.pto_render_rect(sc.h, 0, 3, 0, 520, 1920, 64, P(band))
    sc.close()
    assert (ids != rid).mean() <= EDGE_FRACTION
    assert gpu.stats().paths == paths
    assert (bits(a[520:584]) == bits(band[520:584])).all()


@pytest.mark.parametrize("name", ["demo-1080p-64spp-d8", "mixed9-1080p-64spp-d16"])
def test_whole_brute_force_frames_against_the_oracle(gpu, name):
    """Every pixel of the headline frame (and of the nine-entity frame of every kind and material at the same size)
    against the oracle's whole frame, rendered on sixteen host threads: bit for bit where no entity is rough, within
    the radiance tolerance where Oren-Nayar's libm calls are involved."""
    import bench
    world, objs, kw = bench.workload(name)
    st = default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
    img = gpu.render(world, objs, st)
    sc = OracleScene(world, objs, st)
    ref = sc.render(policy=3, threads=16)
    paths = sc.path_count()
    sc.close()
    assert gpu.stats().paths == paths
    fin = np.isfinite(ref).all(axis=2)
    assert (np.isfinite(img).all(axis=2) == fin).all() and (~fin).mean() <= EDGE_FRACTION
    if all(o.roughness == 0.0 for o in objs):
        assert (bits(img[fin]) == bits(ref[fin])).all(), f"{int((bits(img) != bits(ref)).any(axis=2).sum())} pixels differ"
    else:
        e = rel_err(img[fin], ref[fin])
        assert (e > REL_TOL).sum() <= max(1, int(EDGE_FRACTION * e.size) + 1)
        assert np.abs(img[fin].astype(np.float64) - ref[fin]).mean() < 1e-6


@pytest.mark.parametrize("name", ["glass-1080p-128spp-d32", "dense100k-1920x1080-128spp-d8"])
def test_whole_grid_frames_against_the_oracle(gpu, name):
    """Every pixel of the glass frame at full size and of the 100 000-entity frame at 1080p / 128 spp against the oracle's
    whole frame (sixteen host threads, 12 and 19 seconds).  The glass frame, whose entities are rough, within the radiance
    tolerance everywhere (the largest relative error seen is 3e-7: libm against the device's own powf and acosf).  The
    100 000-entity frame bit for bit."""
    import bench
    world, objs, kw = bench.workload(name)
    st = default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
    img = gpu.render(world, objs, st)
    sc = OracleScene(world, objs, st)
    ref = sc.render(policy=3, threads=16)
    paths = sc.path_count()
    sc.close()
    assert gpu.stats().paths == paths
    fin = np.isfinite(ref).all(axis=2)
    assert (np.isfinite(img).all(axis=2) == fin).all() and (~fin).mean() <= EDGE_FRACTION
    if all(o.roughness == 0.0 for o in objs):
        differ = (bits(img) != bits(ref)).any(axis=2) & fin
        assert not differ.any(), f"{int(differ.sum())} pixels differ"
    else:
        e = rel_err(img[fin], ref[fin])
        assert (e > REL_TOL).sum() <= max(1, int(EDGE_FRACTION * e.size) + 1)
        assert np.abs(img[fin].astype(np.float64) - ref[fin]).mean() < 1e-6


def test_cones_of_extreme_proportions(gpu):
    """Needles, discs and lopsided cones among a grid scene's entities: the direction test that lets a cone past the bounding-sphere
    filter gives such a cone no data (every ray keeps it), and the frame is the oracle's, bit for bit."""
    world, objs = scenes.mixed_scene(seed=21, n=36, rough_share=0.0, spread=10.0)
    rng = np.random.default_rng(5)
    for k, dim in enumerate([(0.05, 0.05, 6.0), (0.3, 0.3, 9.0), (6.0, 6.0, 0.02), (5.0, 0.04, 2.0), (0.04, 5.0, 2.0), (1e-3, 1e-3, 3.0), (3.0, 3.0, 3.0)]):
        q = rng.normal(size=4)
        q = q / np.linalg.norm(q)
        rot = (1, 0, 0, 0) if k % 3 == 0 else tuple(float(v) for v in q)
        objs.append(make_object((float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8)), float(rng.uniform(2, 8))), dim, api.PT_GEOM_CONE, rot=rot,
                                color=(0.8, 0.6, 0.4), refl=1.0 if k % 2 else 0.0))
    st = default_settings(320, 180, 16, 8, seed=99, grid_hint=8, brute_force_below=10)
    img = gpu.render(world, objs, st)
    sc = OracleScene(world, objs, st)
    ref = sc.render(policy=3, threads=8)
    paths = sc.path_count()
    sc.close()
    assert gpu.stats().paths == paths
    fin = np.isfinite(ref).all(axis=2)
    assert (np.isfinite(img).all(axis=2) == fin).all()
    assert (bits(img[fin]) == bits(ref[fin])).all()


def _crops_by_cost(cost, width, height, cw=32, ch=8):
    """Where a full-size frame is compared with the oracle: 32x8 crops (four tiles side by side) at the dearest tile, the
    cheapest tile that still casts more than its sixteen primary rays, three more of the dearest tiles of distinct tile
    rows, the four image corners, and a whole image row through the dearest tile (as 32-pixel pieces, every fourth)."""
    tiles_x = (width + 7) // 8
    order = np.argsort(-cost.astype(np.int64), kind="stable")
    crops, rows = [], set()

    def add(x0, y0):
        x0, y0 = int(min(max(0, x0), width - cw)), int(min(max(0, y0), height - ch))
        if (x0, y0) not in crops:
            crops.append((x0, y0))

    for t in order[:64]:
        ty, tx = divmod(int(t), tiles_x)
        if ty not in rows and len(rows) < 4:
            rows.add(ty)
            add(tx * 8 - 8, ty * 8)
    busy = np.flatnonzero(cost > 16)
    if busy.size:
        t = int(busy[np.argmin(cost[busy])])
        add((t % tiles_x) * 8, (t // tiles_x) * 8)
    for x0, y0 in ((0, 0), (width - cw, 0), (0, height - ch), (width - cw, height - ch)):
        add(x0, y0)
    row = (int(order[0]) // tiles_x) * 8
    for x0 in range(0, width - cw + 1, 4 * cw):
        add(x0, row)
    return crops, cw, ch


@pytest.mark.parametrize("name", ["glass-1080p-128spp-d32", "dense100k-4k-256spp-d8", "mixed9-1080p-64spp-d16", "demo-1080p-64spp-d8"])
def test_grid_scenes_at_full_size(gpu, name):
    """The two grid-path configurations at their full sizes (49 refractive entities with depth of field at
    1080p / 128 spp / depth 32; 100 000 entities on a 64^3-hint grid at 3840x2160 / 256 spp), and the two brute-force
    frames long enough to run the trace kernel's six-wavefront variants (nine entities of every kind and material; the
    demo scene; a launch of an eighth of either, as the group at the end renders, runs the five-wavefront ones): two runs
    are bitwise equal, the image is finite and in [0,1], the path count is the oracle's, and crops of the
    full-size image placed by the device's own cost estimate - the dearest tiles, the cheapest busy one, the
    corners, a row through the dearest tile - match the oracle's pixels (same sample ranges, same streams):
    bit for bit where no entity is rough (the 100 000-entity scene), within the radiance tolerance where
    Oren-Nayar's libm calls are involved (the glass scene).  One crop is also rendered through a group of
    eight partitions."""
    import bench
    world, objs, kw = bench.workload(name)
    st = default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
    gpu.prepare(world, objs, st)
    cost = gpu.tile_cost()
    gpu.trace()
    a = gpu.download()
    gpu.trace()
    b = gpu.download()
    assert (bits(a) == bits(b)).all()
    if len(objs) < st.brute_force_below and "PT_BRUTE_WAVES" not in os.environ and "PT_GRID_BLOCKS" not in os.environ:
        # the long launch runs six workgroups per compute unit, an eighth of the same frame five
        blocks_long = gpu.stats().grid_blocks
        eighth = default_settings(st.width, st.height, st.spp_wanted, st.max_bounces, seed=st.seed, part_index=0, part_count=8)
        gpu.prepare(world, objs, eighth)
        gpu.trace()
        assert blocks_long * 5 == gpu.stats().grid_blocks * 6, (blocks_long, gpu.stats().grid_blocks)
        gpu.prepare(world, objs, st)
    sc = OracleScene(world, objs, st)
    # A degenerate path (the reference normalises a zero vector on some refraction chains) makes its pixel
    # NaN in the reference's arithmetic too: such pixels must be rare and NaN in the oracle as well.
    bad = ~np.isfinite(a).all(axis=2)
    assert bad.mean() <= EDGE_FRACTION
    for y, x in list(zip(*np.nonzero(bad)))[:8]:
        one = np.zeros((st.height, st.width, 3), np.float32)
        sc.lib.pto_render_rect(sc.h, 0, 3, int(x), int(y), 1, 1, P(one))
        assert not np.isfinite(one[y, x]).all(), (x, y)
    good = a[~bad]
    assert good.min() >= 0 and good.max() <= 1
    assert gpu.stats().paths == sc.path_count()
    assert cost.size == ((st.width + 7) // 8) * ((st.height + 7) // 8) and cost.max() > 16
    crops, cw, ch = _crops_by_cost(cost, st.width, st.height)
    assert len(crops) >= 8
    rough_free = all(o.roughness == 0.0 for o in objs)
    want = np.zeros((st.height, st.width, 3), np.float32)
    beyond = values = 0
    busy_crops = 0
    for x0, y0 in crops:
        sc.lib.pto_render_rect(sc.h, 0, 3, x0, y0, cw, ch, P(want))
        got, ref = a[y0:y0 + ch, x0:x0 + cw], want[y0:y0 + ch, x0:x0 + cw]
        busy_crops += 1 if (ref.max() > 0 and ref.min() < 1) else 0
        fin = np.isfinite(ref).all(axis=2)
        assert (np.isfinite(got).all(axis=2) == fin).all(), (x0, y0)
        if rough_free:
            assert (bits(got[fin]) == bits(ref[fin])).all(), f"crop at {x0},{y0} differs from the oracle"
        else:
            e = rel_err(got[fin], ref[fin])
            beyond += int((e > REL_TOL).sum())
            values += e.size
            assert np.abs(got[fin].astype(np.float64) - ref[fin]).mean() < 1e-6, (x0, y0)
    sc.close()
    assert busy_crops >= 4, "the crops must sit on the scene's content"
    if not rough_free:
        assert beyond <= max(1, int(EDGE_FRACTION * values) + 1), f"{beyond} of {values} values beyond the radiance tolerance"
    # the dearest crop once more, rendered by a group of eight partitions (the multi-GPU split) on this one GPU
    x0, y0 = crops[0]
    g = api.Group([0] * 8)
    whole = g.render(world, objs, st)
    g.close()
    assert (bits(whole[y0:y0 + ch, x0:x0 + cw]) == bits(a[y0:y0 + ch, x0:x0 + cw])).all()
    same = bits(whole) == bits(a)
    assert (same | ~np.isfinite(a)).all(), "the eight-partition frame differs from the one-partition frame"


def _frame_in_a_child_process(name, path, **env):
    """One whole frame of a benchmark workload rendered by a child process under environment switches of the library (they
    are read once per process)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.run([sys.executable, "-u", os.path.join(here, "diag_frames.py"), "render", name, str(path)], env=dict(os.environ, **env), check=True, timeout=300)
    return np.load(str(path))


@pytest.mark.parametrize("name", ["dense100k-4k-256spp-d8", "glass-1080p-128spp-d32"])
def test_whole_frames_cooperative_cast_against_per_lane_walk(gpu, name, tmp_path):
    """Every pixel of the two grid frames at full size, rendered by the cooperative cast and by the per-lane grid walk
    (PT_GRID_PER_LANE=1, read once per process: a child process renders it) - two implementations of the reference's
    traversal that share the intersectors and nothing else: the same bits, every pixel.
    The cooperative cast's plain variants (a scene without cones runs them; PT_FAST_CONES=1 forces them) may differ in an
    epsilon-edge share of the pixels, and where they do the per-lane walk must be the oracle's pixel: the reference's cone
    intersector divides by dx^2 + dy^2 - dz^2 of the ray's unit-space direction (code/raycast.cpp:352-388), which rounds to
    zero for a ray parallel to one of the cone's generators; one root is then NaN or garbage, the other is replaced by the
    crossing of the base plane, and the cast accepts that point - up to a cell away from the cone - when it lies in a cell
    that lists the cone.  The plain variants' bounding-sphere filter has turned such a ray away before the intersector
    sees it (12 of 8.3 million pixels of the 100 000-entity frame; none of the glass frame); the variants a scene with
    cones runs by default let it through by its direction."""
    import bench
    world, objs, kw = bench.workload(name)
    st = default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
    a = gpu.render(world, objs, st)
    b = _frame_in_a_child_process(name, tmp_path / "lane.npy", PT_GRID_PER_LANE="1")
    assert a.shape == b.shape
    assert (bits(a) == bits(b)).all(), f"{int((bits(a) != bits(b)).any(axis=2).sum())} pixels differ"
    c = _frame_in_a_child_process(name, tmp_path / "fast.npy", PT_FAST_CONES="1")
    differ = (bits(c) != bits(b)).any(axis=2)
    assert differ.mean() <= EDGE_FRACTION, f"{int(differ.sum())} pixels differ with PT_FAST_CONES=1"
    if differ.any():
        assert any(o.type == api.PT_GEOM_CONE for o in objs)
        sc = OracleScene(world, objs, st)
        one = np.zeros((st.height, st.width, 3), np.float32)
        for y, x in list(zip(*np.nonzero(differ)))[:16]:
            sc.lib.pto_render_rect(sc.h, 0, 3, int(x), int(y), 1, 1, P(one))
            assert (bits(one[y, x]) == bits(b[y, x])).all(), (int(x), int(y))
        sc.close()


@pytest.mark.parametrize("seed", list(range(100, 110)))
def test_cast_soak_first_scenes(gpu, seed):
    """The first ten scenes of the longer soak run (tests/soak_cast_parity.py): random scenes, brute-force sized on even
    seeds and grid sized on odd ones; ids, hit positions and normals of 20 000 rays against the oracle, bit for bit."""
    small = seed % 2 == 0
    n_obj = 3 + seed % 7 if small else 20 + (seed * 7) % 101
    world, objs = scenes.mixed_scene(seed=seed, n=n_obj, spread=6.0 + seed % 9)
    st = api.default_settings(64, 36, 4, 4, seed=4242, grid_hint=6 + seed % 7, brute_force_below=10)
    gpu.prepare(world, objs, st)
    sc = OracleScene(world, objs, st)
    n = 20000
    o, d = scene_rays(n, seed, objs, spread=20.0)
    ids, pos, nrm = gpu.cast_rays(o, d)
    rid, rpos, rnrm = np.zeros(n, np.int32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    sc.lib.pto_cast(sc.h, n, P(o), P(d), 0 if len(objs) < st.brute_force_below else 1, rid.ctypes.data_as(IP), P(rpos), P(rnrm))
    sc.close()
    hit = rid != 0
    assert hit.mean() > 0.1
    assert (ids == rid).all()
    assert (bits(pos[hit]) == bits(rpos[hit])).all() and (bits(nrm[hit]) == bits(rnrm[hit])).all()


@pytest.mark.parametrize("name", ["mixed40", "crowded", "dense2k", "dense30k", "aligned_boxes"])
def test_grid_built_on_device_equals_host_build(gpu, name, monkeypatch):
    """The device passes (count, scan, fill, order) must give the host build's CSR word for word (ascending
    ids per cell, the reference's append order), the same packed cell ranges, and therefore the same image;
    and both are the reference's lists (pinned in the CPU tests) less the entities that do not reach a cell."""
    if name == "dense30k":
        world, objs = scenes.dense_scene(n=30000, seed=11, half_extent=120.0, height=60.0)
        st = default_settings(96, 54, 4, 8, seed=77, grid_hint=40)
    else:
        world, objs = CASES[name][0]()
        st = settings_for(name, 96, 54, 4, 8)
    got = {}
    for where in ("host", "device"):
        monkeypatch.setenv("PT_GRID_BUILD", where)
        img = gpu.render(world, objs, st)
        start, lists, ranges, on_device = gpu.debug_grid()
        assert on_device == (where == "device")
        got[where] = (start, lists, ranges, img)
    monkeypatch.delenv("PT_GRID_BUILD")
    for a, b in zip(got["host"][:3], got["device"][:3]):
        assert a.shape == b.shape and (a == b).all()
    assert (bits(got["host"][3]) == bits(got["device"][3])).all()
    # Against the reference's lists (the oracle's, pinned to the reference on the CPU): a cell's list here is the reference's list
    # of that cell, in its order, less the entities that stay clear of the cell - every entity left out has its own bounding
    # sphere (worked out here, in double precision, from the scene's description) wholly outside the cell's box.
    sc = OracleScene(world, objs, st)
    g, cs, ci = sc.grid()
    sc.close()
    dev_start, dev_items = got["device"][0].astype(np.int64), got["device"][1]
    assert dev_start.shape == cs.shape
    cmin, cnt, size = np.array(g[0:3], np.float64), np.array(g[3:6], np.int64), np.array(g[6:9], np.float64)
    half = np.array([[o.dim[0], o.dim[1], o.dim[2]] for o in objs], np.float64) / 2
    pos = np.array([[o.pos[0], o.pos[1], o.pos[2]] for o in objs], np.float64)
    kind = np.array([o.type for o in objs])
    radius = np.where(kind == api.PT_GEOM_SPHERE, half.max(axis=1),
                      np.where(kind == api.PT_GEOM_BOX, np.sqrt((half ** 2).sum(axis=1)), np.sqrt(np.maximum(half[:, 0], half[:, 1]) ** 2 + half[:, 2] ** 2)))
    # a cone's sphere goes through its apex and the rim of its base (about the base's centre when the cone is flatter than wide)
    for e in np.flatnonzero(kind == api.PT_GEOM_CONE):
        rim, height = max(half[e, 0], half[e, 1]), 2 * half[e, 2]
        above_base = (height ** 2 - rim ** 2) / (2 * height) if height > rim else 0.0
        qw, qx, qy, qz = (float(v) for v in objs[e].rot)
        axis = np.array([2 * (qx * qz + qw * qy), 2 * (qy * qz - qw * qx), 1 - 2 * (qx * qx + qy * qy)])
        pos[e] += axis * (above_base - height / 2)
        radius[e] = (height ** 2 + rim ** 2) / (2 * height) if height > rim else rim
    dropped = kept = 0
    for cell in range(len(cs) - 1):
        ref = ci[cs[cell]:cs[cell + 1]]
        dev = dev_items[dev_start[cell]:dev_start[cell + 1]]
        if len(ref) == len(dev):
            assert (ref == dev).all()
            kept += len(dev)
            continue
        assert np.isin(dev, ref).all() and (np.diff(dev.astype(np.int64)) > 0).all()
        kept += len(dev)
        lo = (np.array([cell % cnt[0], (cell // cnt[0]) % cnt[1], cell // (cnt[0] * cnt[1])], np.float64) + cmin) * size
        for e in ref[~np.isin(ref, dev)]:
            gap = np.maximum(np.maximum(lo - pos[e], 0.0), pos[e] - (lo + size))
            assert np.sqrt((gap ** 2).sum()) > radius[e], (cell, int(e))
            dropped += 1
    assert kept == len(dev_items) and kept + dropped == len(ci)


@pytest.mark.parametrize("w,h,spp", [(1, 1, 16), (8, 8, 4), (9, 17, 64), (24, 8, 4), (300, 7, 128)])
def test_small_and_ragged_images(gpu, w, h, spp):
    """Images smaller than a work group, with ragged edge tiles, or smaller than the reference's own tile
    jobs can describe (its job size is (int)sqrt(W*H/204), zero below 204 pixels): the oracle renders them
    as one rectangle."""
    world, objs = scenes.default_scene()
    st = default_settings(w, h, spp, 8, seed=5)
    img = gpu.render(world, objs, st)
    sc = OracleScene(world, objs, st)
    ref = np.zeros((h, w, 3), np.float32)
    sc.lib.pto_render_rect(sc.h, 0, 3, 0, 0, w, h, P(ref))
    paths = sc.path_count()
    sc.close()
    assert gpu.stats().paths == paths
    assert (bits(img) == bits(ref)).all()


def test_device_grid_build_hands_long_lists_to_the_host(gpu, monkeypatch):
    """The device build sorts a cell's list with one thread, so a cell that lists more than 4096 entities
    sends the build back to the host; the render is the oracle's either way."""
    rng = np.random.default_rng(3)
    world, base = scenes.default_scene()
    objs = [base[0]]
    for _ in range(5000):                                   # 5000 small spheres inside one grid cell
        p = rng.uniform(-0.3, 0.3, 3) + np.array([3.0, 3.0, 6.5])
        objs.append(make_object(tuple(float(v) for v in p), (0.05, 0.05, 0.05), api.PT_GEOM_SPHERE, color=(0.8, 0.3, 0.3)))
    st = default_settings(48, 27, 2, 3, seed=9, grid_hint=4)
    monkeypatch.setenv("PT_GRID_BUILD", "device")
    img = gpu.render(world, objs, st)
    start, lists, _, on_device = gpu.debug_grid()
    monkeypatch.delenv("PT_GRID_BUILD")
    assert not on_device and int(np.diff(start).max()) > 4096
    sc = OracleScene(world, objs, st)
    ref = np.zeros((27, 48, 3), np.float32)
    sc.lib.pto_render_rect(sc.h, 0, 3, 0, 0, 48, 27, P(ref))
    _, cs, ci = sc.grid()
    sc.close()
    assert (cs == start).all() and (ci == lists).all()
    assert (bits(img) == bits(ref)).all()


def test_zero_sample_tile_is_black(gpu):
    world, objs = scenes.default_scene()
    for seed in range(40):
        st = default_settings(30, 30, 1, 4, seed=seed)
        sc = OracleScene(world, objs, st)
        _, offs = sc.samples()
        if (np.diff(offs) == 0).any():
            ref = sc.render(policy=3)
            sc.close()
            img = gpu.render(world, objs, st)
            assert np.isfinite(img).all() and (bits(img) == bits(ref)).all()
            assert gpu.stats().zero_sample_pixels > 0
            return
        sc.close()
    pytest.fail("no seed produced an empty tile")


def test_start_progress_abort(gpu):
    world, objs = scenes.glass_scene(seed=3, per_type=16)
    st = settings_for("glass", 1920, 1080, 64, 32)
    out = gpu.start(world, objs, st)
    with pytest.raises(api.PtError):
        gpu.start(world, objs, st)            # busy
    seen = []
    t0 = time.time()
    while time.time() - t0 < 60:
        p = gpu.progress()
        seen.append(p.fraction)
        if p.done:
            break
        if p.fraction > 0.05:
            gpu.abort()
        time.sleep(0.002)
    gpu.wait()
    p = gpu.progress()
    assert p.done == 1 and p.seconds > 0
    assert all(0 <= f <= 1 for f in seen)
    assert out.shape == (1080, 1920, 3)
    # a render left alone: the fraction polled while it runs never goes down and ends at 1
    st3 = settings_for("glass", 1280, 720, 64, 32)
    gpu.start(world, objs, st3)
    polled = []
    t0 = time.time()
    while time.time() - t0 < 60:
        p = gpu.progress()
        polled.append(p.fraction)
        if p.done:
            break
        time.sleep(0.001)
    gpu.wait()
    assert polled == sorted(polled) and polled[-1] == 1.0 and gpu.progress().aborted == 0
    assert len(polled) < 3 or max(polled[:-1]) > 0.1, "the fraction must move while the render runs"
    # an un-aborted render of the same context afterwards is complete and correct
    st2 = settings_for("default", 160, 90, 16, 8)
    w2, o2 = scenes.default_scene()
    img = gpu.render(w2, o2, st2)
    p = gpu.progress()
    assert p.done == 1 and p.aborted == 0 and p.fraction == 1.0 and p.paths_done == gpu.stats().paths
    sc = OracleScene(w2, o2, st2)
    assert (bits(img) == bits(sc.render(policy=3, threads=4))).all()
    sc.close()


def test_page_locked_image(gpu):
    """pt_image_alloc: the finished image is copied straight into a page-locked buffer; same floats as into a pageable
    one (which goes through the staging image and the mover threads from 4 MB up: 1280x720 is 11 MB)."""
    world, objs = scenes.mixed_scene(seed=7, n=40)
    st = settings_for("mixed40", 1280, 720, 4, 4)
    pageable = gpu.render(world, objs, st).copy()
    image = gpu.image_alloc(1280, 720)
    image[:] = -1
    gpu.start(world, objs, st, out=image)
    gpu.wait()
    assert (bits(image) == bits(pageable)).all()
    gpu.image_free(image)
    with pytest.raises(api.PtError):
        gpu.image_alloc(0, 10)


def test_srgb8_present(gpu):
    """Screenshot bytes are the CPU path's, bit for bit: over a rendered image, over a sample of every
    magnitude including the neighbourhood of each step of the encoder, and - through the histogram of a
    device sweep - over every float from 0.0 to 1.0 (the thresholds themselves are pinned to the reference's
    expression in the CPU tests)."""
    from oracle.oraclebind import U8P
    world, objs = scenes.default_scene()
    st = settings_for("default", 160, 90, 16, 8)
    gpu.prepare(world, objs, st)
    gpu.trace()
    lin = gpu.download()
    got = gpu.present_srgb8()
    sc = OracleScene(world, objs, st)
    want = np.zeros(lin.size, np.uint8)
    sc.lib.pto_srgb8(lin.size, P(lin), want.ctypes.data_as(U8P))
    assert (got.ravel() == want).all()
    thr = api.srgb8_thresholds()
    tb = thr.view(np.uint32).astype(np.int64)
    near = (tb[:, None] + np.arange(-2048, 2048)[None, :]).clip(0, 0x3f800000).astype(np.uint32).view(np.float32).ravel()
    rng = np.random.default_rng(2)
    v = np.concatenate([near, rng.random(1 << 20, dtype=np.float32), np.linspace(0, 1, 100003, dtype=np.float32),
                        np.array([0.0, 1.0, 0.0031308, np.nan], np.float32)])[:-1]      # 4n + 3 values: the encoder's tail loop runs
    v = np.ascontiguousarray(v, np.float32)
    want = np.zeros(v.size, np.uint8)
    sc.lib.pto_srgb8(v.size, P(v), want.ctypes.data_as(U8P))
    sc.close()
    assert (gpu.srgb8(v) == want).all()
    bad, hist = gpu.srgb8_sweep(0, 0x3f800000 + 1)
    assert bad == 0
    edges = np.concatenate([[0], tb, [0x3f800000 + 1]])
    assert (hist.astype(np.int64) == np.diff(edges)).all()


def test_pick_rays_test_every_entity(gpu):
    """The editor's picking ray (code/app.cpp:1637) is castRayAll with `outside` false: every entity, grid or no
    grid - also on scenes the tracer walks the grid for, and for rays that never meet the grid."""
    for name in ("default", "mixed40", "crowded"):
        world, objs = CASES[name][0]()
        st = settings_for(name, 64, 36, 4, 4)
        gpu.prepare(world, objs, st)
        sc = OracleScene(world, objs, st)
        n = 20000
        o, d = scene_rays(n, 78, objs, spread=60.0)
        ids, pos, nrm = gpu.cast_rays(o, d, pick=True)
        rid, rpos, rnrm = np.zeros(n, np.int32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        sc.lib.pto_cast(sc.h, n, P(o), P(d), 2, rid.ctypes.data_as(IP), P(rpos), P(rnrm))
        sc.close()
        assert (ids == rid).all() and (rid != 0).mean() > 0.25
        hit = rid != 0
        assert (bits(pos[hit]) == bits(rpos[hit])).all() and (bits(nrm[hit]) == bits(rnrm[hit])).all()


def test_bounding_sphere_prefilter_on_adversarial_rays(gpu):
    """The brute-force cast replaces the two square roots of the reference's bounding-sphere test
    (code/rt_math.cpp:2361-2373) by margin tests wherever the answer is clear.  Rays built to sit on the margins:
    tangent to a bounding sphere within a few ulps, origins at |lp|^2 = (0.999 +- 1e-6) r^2 and (1 +- 1e-6) r^2,
    directions of squared length around 4, origins on the sphere looking along the tangent plane."""
    world, objs = scenes.plain_scene(seed=5, n=8)
    more_w, more = scenes.mixed_scene(seed=11, n=8)
    for w_, objs_ in ((world, objs), (more_w, more)):
        st = default_settings(64, 36, 4, 4, seed=4242)
        gpu.prepare(w_, objs_, st)
        sc = OracleScene(w_, objs_, st)
        rec = np.zeros(len(objs_) * api.PT_F_STRIDE, np.float32)
        api.load_core().pt_host_flatten(api.object_array(objs_), len(objs_), api.fptr(rec))
        rec = rec.reshape(len(objs_), api.PT_F_STRIDE)
        rng = np.random.default_rng(17)
        os_, ds_ = [], []
        for k in range(len(objs_)):
            c = rec[k, api.F["POS"]:api.F["POS"] + 3].astype(np.float64)
            r = float(rec[k, api.F["BRADIUS"]])
            m = 6000
            u = rng.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
            t = np.cross(u, rng.normal(size=(m, 3))); t /= np.linalg.norm(t, axis=1)[:, None]
            q = m // 6
            # tangent lines: closest approach at distance r (1 + e), e within +-4e-6, seen from 2r .. 6r away
            e = rng.uniform(-4e-6, 4e-6, m)
            touch = c + u * (r * (1 + e))[:, None]
            back = rng.uniform(2, 6, m)[:, None] * r
            o = touch - t * back
            d = t.copy()
            # origins at |lp|^2 = f r^2 with f around the two switch points, any direction (aimed and random)
            f = np.where(rng.random(q) < 0.5, 0.999, 1.0) + rng.uniform(-2e-6, 2e-6, q)
            o[q:2 * q] = c + u[q:2 * q] * (r * np.sqrt(f))[:, None]
            d[q:2 * q] = np.where(rng.random((q, 1)) < 0.5, t[q:2 * q], -u[q:2 * q] + 0.3 * t[q:2 * q])
            d[q:2 * q] /= np.linalg.norm(d[q:2 * q], axis=1)[:, None]
            # the same origins with directions of squared length around 4 (the shortcut's other condition)
            o[2 * q:3 * q] = o[q:2 * q]
            d[2 * q:3 * q] = d[q:2 * q] * np.sqrt(4 + rng.uniform(-1e-5, 1e-5, q))[:, None]
            # origins on the sphere, direction in the tangent plane tilted by a few ulps in or out
            o[3 * q:4 * q] = c + u[3 * q:4 * q] * r
            d[3 * q:4 * q] = t[3 * q:4 * q] + u[3 * q:4 * q] * rng.uniform(-3e-7, 3e-7, q)[:, None]
            os_.append(o); ds_.append(d)
        o = np.ascontiguousarray(np.concatenate(os_), np.float32)
        d = np.ascontiguousarray(np.concatenate(ds_), np.float32)
        n = o.shape[0]
        for pick in (False, True):
            ids, pos, nrm = gpu.cast_rays(o, d, pick=pick)
            rid, rpos, rnrm = np.zeros(n, np.int32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
            sc.lib.pto_cast(sc.h, n, P(o), P(d), 2 if pick else 0, rid.ctypes.data_as(IP), P(rpos), P(rnrm))
            assert (ids == rid).all(), f"{(ids != rid).sum()} of {n} ids differ (pick={pick})"
            hit = rid != 0
            assert (bits(pos[hit]) == bits(rpos[hit])).all() and (bits(nrm[hit]) == bits(rnrm[hit])).all()
        sc.close()


def test_abort_leaves_unfinished_pixels_alone_and_snapshots_grow(gpu):
    """Escape in the reference (code/app.cpp:1425-1428) leaves a partly written buffer; the frame loop shows the
    half-written buffer every frame (code/app.cpp:1456-1461).  Here: snapshots taken while a render runs hold a
    growing number of final pixels and nothing else; after an abort exactly the finished pixels were written,
    every other pixel keeps the caller's sentinel, and the progress record counts them."""
    world, objs = scenes.glass_scene(seed=3, per_type=16)
    W, H = 1920, 1080
    st = settings_for("glass", W, H, 512, 32)             # a frame of a few hundred milliseconds
    full = gpu.render(world, objs, st)
    sentinel = np.float32(-7.0)
    # a fresh context: its buffers have never held an image
    r = api.Renderer(0)
    out = np.full((H, W, 3), sentinel, np.float32)
    snap = np.full_like(out, sentinel)
    r.start(world, objs, st, out=out)
    counts, early = [], None
    t0 = time.time()
    while time.time() - t0 < 60:
        p = r.progress()
        if p.done:
            break
        n = r.snapshot(snap)
        counts.append(n)
        if early is None and n > 0:
            early = (n, snap.copy())
        if n > 0.15 * W * H:
            r.abort()
    r.wait()
    p = r.progress()
    assert p.done == 1 and p.aborted == 1
    assert counts == sorted(counts) and counts[-1] > 0.15 * W * H, "snapshots must grow while the render runs"
    for n, image in (early, (counts[-1], snap)):
        shown = (image != sentinel).all(axis=2)
        assert ((image == sentinel).all(axis=2) | shown).all()
        assert int(shown.sum()) == n                        # snapshots only ever add pixels
        assert (bits(image[shown]) == bits(full[shown])).all(), "a snapshot pixel is not the final value"
    assert early[0] < counts[-1]
    written = (out != sentinel).all(axis=2)
    untouched = (out == sentinel).all(axis=2)
    assert (written | untouched).all()
    assert counts[-1] <= written.sum() < W * H
    assert (bits(out[written]) == bits(full[written])).all()
    assert p.fraction == pytest.approx(written.sum() / (W * H), abs=1e-6)
    sc = OracleScene(world, objs, st)
    _, offs = sc.samples()
    sc.close()
    per_tile = np.diff(offs).reshape(3, 3)
    yy, xx = np.nonzero(written)
    assert p.paths_done == int(per_tile[yy % 3, xx % 3].sum())
    # the resident-scene interface after an abort: pt_trace clears the request and renders the whole frame
    r.trace()
    assert (bits(r.download()) == bits(full)).all()
    r.close()


def test_unread_launches_do_not_pile_up_events(gpu):
    # a caller that traces without ever reading the statistics out: the timed launches stop at 256 pending ones
    world, objs = scenes.default_scene()
    r = api.Renderer(0)
    r.prepare(world, objs, api.default_settings(64, 48, 16, 4, seed=1))
    for _ in range(4000):
        r.trace(sync=False)
    r.sync()
    st = r.stats()
    assert 4 <= st.trace_launches <= 256
    r.trace(sync=False)                     # and timing goes on after the read-out
    r.sync()
    assert r.stats().trace_launches == 1
    r.close()


def test_argument_errors(gpu):
    world, objs = scenes.default_scene()
    bad = make_object((0, 0, 0), (1, 1, 1), 5)            # one past GEOM_TYPE_PLANE
    with pytest.raises(api.PtError, match="unknown geometry type"):
        gpu.prepare(world, objs + [bad], default_settings())
    only_planes = [make_object((0, 0, 0), (1, 1, 1), api.PT_GEOM_PLANE)]
    with pytest.raises(api.PtError, match="unusable grid"):     # the grid's extent comes from the entities that are not planes
        gpu.prepare(world, only_planes, default_settings())
    with pytest.raises(api.PtError):
        gpu.prepare(world, objs, default_settings(width=0))
    with pytest.raises(api.PtError):
        gpu.prepare(world, objs, default_settings(seed_policy=1))
    with pytest.raises(api.PtError):
        gpu.prepare(world, objs, default_settings(part_index=2, part_count=2))
    r2 = api.Renderer(0)
    with pytest.raises(api.PtError, match="before pt_prepare"):
        r2.trace()
    r2.close()
    # an empty scene renders the background
    st = default_settings(32, 18, 4, 2)
    img = gpu.render(world, [], st)
    assert (img == 1.0).all()


# ---- the short division / square-root forms against IEEE arithmetic ------------------------------------
def _arith_operands(n, seed):
    """Triples and denominators of every kind: ordinary magnitudes, exact zeros of both signs, denormals,
    huge and tiny values, infinities and NaNs, so that both the short forms and their fall-backs run."""
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal((n, 3)) * np.exp(rng.uniform(-12, 12, (n, 1)))).astype(np.float32)
    raw = rng.integers(0, 2**32, (n, 3), dtype=np.uint64).astype(np.uint32).view(np.float32)      # any bit pattern
    a[n // 2:] = raw[n // 2:]
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, 3e38, -3e38, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e-30, 1e30, 2.0**-40, 2.0**40], np.float32)
    pick = rng.integers(0, special.size, (n, 3))
    mask = rng.random((n, 3)) < 0.08
    a[mask] = special[pick[mask]]
    b = (rng.standard_normal(n) * np.exp(rng.uniform(-12, 12, n))).astype(np.float32)
    b[::7] = raw[::7, 0]
    b[::11] = special[pick[::11, 0]]
    return a, b


def _same(x, y):
    return (bits(x) == bits(y)) | (np.isnan(x) & np.isnan(y))


def _assert_same(got, want, what, *operands):
    ok = _same(got, want)
    if ok.all():
        return
    rows = np.unique(np.argwhere(~ok)[:, 0])[:6]
    lines = [f"{what}: {(~ok).sum()} of {ok.size} results differ"]
    for r in rows:
        lines.append("  operands " + " | ".join(str(bits(o[r]).tolist() if np.ndim(o[r]) else hex(int(bits(o[r])))) + " " + str(o[r]) for o in operands)
                     + f" -> got {got[r]} {bits(got[r]).tolist() if np.ndim(got[r]) else hex(int(bits(got[r])))} want {want[r]} {bits(want[r]).tolist() if np.ndim(want[r]) else hex(int(bits(want[r])))}")
    raise AssertionError("\n".join(lines))


def test_short_division_and_root_are_ieee(gpu):
    n = 400000
    a, b = _arith_operands(n, 5)
    with np.errstate(all="ignore"):
        # numpy's float32 divide and sqrt are the correctly rounded IEEE operations, as the CPU path's are
        got = gpu.arith(1, a, b)
        _assert_same(got, a / b[:, None], "Vec3 / float against IEEE division", a, b)
        got = gpu.arith(3, a)
        _assert_same(got, a / np.float32(3.14159274101257324), "Vec3 / pi against IEEE division", a)
        x = np.abs(a[:, 0])
        x[::5] = a[::5, 1]                                   # negative arguments too
        got = gpu.arith(0, x)
        _assert_same(got, np.sqrt(x), "square root against IEEE", x)
        got = gpu.arith(4, a)
        _assert_same(got, np.float32(1) / a, "1 / Vec3 against IEEE division", a)
        ss = (a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2]
        length = np.sqrt(ss)
        got = gpu.arith(2, a)
        _assert_same(got, a / length[:, None], "normalise against a / sqrt(x*x + y*y + z*z)", a)


@pytest.mark.parametrize("op", [0, 1, 2, 3, 4])
def test_short_forms_equal_the_compilers_over_all_bit_patterns(gpu, op):
    # on the device: every one of the 2^32 float bit patterns as (first) operand, short form against the long one
    assert gpu.arith_sweep(op, 0, 1 << 32) == 0
