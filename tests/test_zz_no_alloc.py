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
