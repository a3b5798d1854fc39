"""Renders one parity-test case and saves the image (debugging aid: compare two library variants).
usage: PT_CORE_LIB=... python profiles/dbg/render_case.py <case> <w> <h> <spp> <bounces> <out.npy>"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch  # noqa: F401
from pathtracer_amd import api, scenes
from pathtracer_amd.api import default_settings
CASES = {
    "default_grid": (lambda: scenes.default_scene(), 10, 0),
    "glass": (lambda: scenes.glass_scene(seed=3, per_type=16), 10, 10),
    "mixed40": (lambda: scenes.mixed_scene(seed=7, n=40), 10, 10),
}
name, w, h, spp, b, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
world, objs = CASES[name][0]()
st = default_settings(w, h, spp, b, seed=4242, grid_hint=CASES[name][1], brute_force_below=CASES[name][2])
r = api.Renderer(0)
img = r.render(world, objs, st)
print(name, "paths", r.stats().paths, "rays", r.stats().rays, "mean", float(img.mean()))
np.save(out, img)
r.close()
