"""The allocation-free ray query against the ordinary one: pt_cast_rays_noalloc stages its rays in the
library's static device array instead of a buffer the context allocates, and must return what pt_cast_rays
returns, bit for bit - on two calls in a row, the second with fewer rays (the array keeps the first call's
contents), on the brute-force and on the grid cast, and it must refuse more rays than the array holds."""
import ctypes as C

import numpy as np
import pytest

from pathtracer_amd import api, scenes
from pathtracer_amd.api import default_settings, fptr, iptr
from rayutil import scene_rays

pytestmark = pytest.mark.gpu


def _query(r, fn, o, d):
    n = o.shape[0]
    ids = np.full(n, -5, np.int32)
    pos = np.full((n, 3), -5, np.float32)
    nrm = np.full((n, 3), -5, np.float32)
    rc = fn(r.ctx, fptr(o), fptr(d), n, iptr(ids), fptr(pos), fptr(nrm))
    return rc, ids, pos, nrm


@pytest.mark.parametrize("name", ["default", "mixed40"])
def test_noalloc_ray_query_equals_the_ordinary_one(name):
    world, objs = scenes.default_scene() if name == "default" else scenes.mixed_scene(seed=7, n=40)
    r = api.Renderer(0)
    r.prepare(world, objs, default_settings(64, 36, 4, 4, seed=4242))
    lib = r.lib
    limit = 4096
    for n, seed in ((limit, 5), (1000, 6), (1, 7)):          # the second and third calls use less of the array
        o, d = scene_rays(max(n, 64), seed, objs, spread=30.0)
        o, d = np.ascontiguousarray(o[:n]), np.ascontiguousarray(d[:n])
        rc, ids, pos, nrm = _query(r, lib.pt_cast_rays_noalloc, o, d)
        assert rc == 0, lib.pt_last_error(r.ctx)
        rc, wid, wpos, wnrm = _query(r, lib.pt_cast_rays, o, d)
        assert rc == 0
        assert (ids == wid).all()
        assert (pos.view(np.uint32) == wpos.view(np.uint32)).all()
        assert (nrm.view(np.uint32) == wnrm.view(np.uint32)).all()
        if n > 100:
            assert (wid != 0).mean() > 0.25
    o, d = scene_rays(limit + 1, 8, objs)
    rc, ids, _, _ = _query(r, lib.pt_cast_rays_noalloc, o, d)
    assert rc == -6 and b"at most 4096" in lib.pt_last_error(r.ctx)       # PT_ERR_LIMIT
    assert (ids == -5).all()                                             # nothing written
    r.close()


def _memory(r):
    free, total, cap = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    rc = r.lib.pt_debug_device_memory(r.ctx, C.byref(free), C.byref(total), C.byref(cap))
    assert rc == 0, r.lib.pt_last_error(r.ctx)
    return free.value, cap.value


def test_noalloc_ray_query_allocates_nothing():
    """A hundred calls move neither the device's free memory (hipMemGetInfo) nor the capacity of the context's own
    scratch buffer; the ordinary query, asked for more rays than it has staged before, does grow that buffer -
    which shows that the probe would see an allocation."""
    world, objs = scenes.mixed_scene(seed=7, n=40)
    r = api.Renderer(0)
    r.prepare(world, objs, default_settings(64, 36, 4, 4, seed=4242))
    lib = r.lib
    lib.pt_debug_device_memory.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    o, d = scene_rays(4096, 11, objs, spread=30.0)
    rc, first, _, _ = _query(r, lib.pt_cast_rays_noalloc, o, d)       # one call before the reading: the runtime's own lazy set-up is not the library's
    assert rc == 0
    free0, cap0 = _memory(r)
    for k in range(100):
        n = 4096 if k % 3 == 0 else 1 + (k * 37) % 4000
        rc, ids, _, _ = _query(r, lib.pt_cast_rays_noalloc, np.ascontiguousarray(o[:n]), np.ascontiguousarray(d[:n]))
        assert rc == 0
        assert (ids == first[:n]).all()
    free1, cap1 = _memory(r)
    assert free1 == free0, f"device free memory moved by {free0 - free1} bytes over 100 allocation-free calls"
    assert cap1 == cap0
    big_o, big_d = scene_rays(300000, 12, objs, spread=30.0)
    rc, _, _, _ = _query(r, lib.pt_cast_rays, big_o, big_d)
    assert rc == 0
    free2, cap2 = _memory(r)
    assert cap2 > cap1 and free2 < free1, "the probe does not see the ordinary query's allocation"
    r.close()


def test_noalloc_ray_query_is_refused_while_a_render_is_active():
    world, objs = scenes.dense_scene(n=20000, seed=4, half_extent=120.0, height=60.0)
    r = api.Renderer(0)
    st = default_settings(1920, 1080, 64, 8, seed=1, grid_hint=32)
    r.prepare(world, objs, st)
    o, d = scene_rays(64, 3, objs, spread=30.0)
    out = r.start(world, objs, st)
    rc, _, _, _ = _query(r, r.lib.pt_cast_rays_noalloc, o, d)
    busy = not r.progress().done
    r.wait()
    if busy:
        assert rc == -3 and b"render is active" in r.lib.pt_last_error(r.ctx)      # PT_ERR_BUSY
    rc, ids, _, _ = _query(r, r.lib.pt_cast_rays_noalloc, o, d)
    assert rc == 0
    assert np.isfinite(out).all()
    r.close()
