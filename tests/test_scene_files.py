"""Scene files in the reference application's format (code/raycast.cpp:1578-1604): the library's
loader reads what the reference's own saveScene wrote, the reference's own loadScene reads what the
library's saver wrote, and a committed file written by the reference loads to the expected scene."""
import ctypes as C
import os

import numpy as np
import pytest

from pathtracer_amd import api, scenes
from pathtracer_amd.api import default_settings, object_array

GOLD_SCENE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mixed12.scene")


def same_objects(a, b):
    return len(a) == len(b) and all(bytes(x) == bytes(y) for x, y in zip(a, b))


def gold_scene():
    return scenes.mixed_scene(seed=21, n=11)


def test_round_trip(core_lib, tmp_path):
    world, objs = scenes.glass_scene(seed=3, per_type=16)
    p = str(tmp_path / "glass.scene")
    api.save_scene(p, world, objs)
    assert os.path.getsize(p) == 224 + 172 * len(objs)
    w2, o2 = api.load_scene(p)
    assert bytes(w2) == bytes(world) and same_objects(o2, objs)


def test_loads_file_written_by_the_reference(core_lib):
    """tests/golden/mixed12.scene was written by the reference's saveScene (see make_golden.py)."""
    world, objs = gold_scene()
    assert os.path.getsize(GOLD_SCENE) == 224 + 172 * len(objs)
    w2, o2 = api.load_scene(GOLD_SCENE)
    assert bytes(w2) == bytes(world) and same_objects(o2, objs)


def test_errors(core_lib, tmp_path):
    with pytest.raises(api.PtError):
        api.load_scene(str(tmp_path / "missing.scene"))
    short = tmp_path / "short.scene"
    short.write_bytes(b"\0" * 100)
    with pytest.raises(api.PtError):
        api.load_scene(str(short))


@pytest.mark.ref
def test_against_reference_save_and_load(core_lib, ref_lib, tmp_path):
    world, objs = scenes.mixed_scene(seed=7, n=40)
    st = default_settings(64, 36, 4, 2)
    ref_lib.ref_pcg_seed(0, 1)
    ref_lib.ref_scene_setup(C.byref(world), object_array(objs), len(objs), C.byref(st), 1)
    p = str(tmp_path / "from_ref.scene").encode()
    ref_lib.ref_save_scene(p)
    w2, o2 = api.load_scene(p.decode())
    assert bytes(w2) == bytes(world) and same_objects(o2, objs)
    # and the other way round
    q = str(tmp_path / "from_lib.scene")
    api.save_scene(q, world, objs)
    w3 = api.pt_world()
    o3 = (api.pt_object * len(objs))()
    n = ref_lib.ref_load_scene(q.encode(), C.byref(w3), o3, len(objs))
    assert n == len(objs) and bytes(w3) == bytes(world) and same_objects(list(o3), objs)
