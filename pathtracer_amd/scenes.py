"""Scene generators for the benchmark / parity configurations.

The reference ships no scene files (its data/ holds fonts and images only), so every scene beyond
the built-in demo scene (code/raycast.cpp:1294-1356) is synthesised here from a PCG32 stream, the
generator family the reference itself uses (code/rt_misc.cpp:1152-1251).
"""
import math

import numpy as np

from .api import (PT_GEOM_BOX, PT_GEOM_CONE, PT_GEOM_CYLINDER, PT_GEOM_PLANE, PT_GEOM_SPHERE, make_object,
                  make_world)

_M = (1 << 64) - 1


class PCG32:
    """pcg32 (setseq 64/32 XSH-RR), as code/rt_misc.cpp:1172-1197."""

    def __init__(self, initstate, initseq):
        self.state = 0
        self.inc = ((initseq << 1) | 1) & _M
        self.next()
        self.state = (self.state + initstate) & _M
        self.next()

    def next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & _M
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, lo, hi):
        return lo + (hi - lo) * (self.next() / 4294967296.0)

    def below(self, n):
        return self.next() % n


def default_scene():
    """getDefaultScene: ground box + mirror sphere, white sky (code/raycast.cpp:1294-1356)."""
    world = make_world(cam_pos=(0, -30, 7), cam_rot=(0, 0, 0), fov=90.0, plane_width=10.0,
                       bg=(1, 1, 1), ambient=(1, 1, 1), light=(1, 1, 1), light_rot=(0, 90),
                       focal_distance=30.0, aperture=0.0)
    objs = [make_object((0, 0, 0.1), (50, 50, 0.01), PT_GEOM_BOX, color=(0.5, 0.5, 0.5)),
            make_object((0, 0, 7), (8, 8, 8), PT_GEOM_SPHERE, color=(0.7, 0.7, 0.7), refl=1.0)]
    return world, objs


def _rand_quat(rng):
    while True:
        ax = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)])
        n = np.linalg.norm(ax)
        if 1e-3 < n <= 1:
            break
    ax /= n
    a = rng.uniform(0, 2 * math.pi)
    return (math.cos(a / 2), ax[0] * math.sin(a / 2), ax[1] * math.sin(a / 2), ax[2] * math.sin(a / 2))


def _aabb_half(dim, q):
    w, x, y, z = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return np.abs(r) @ (np.asarray(dim) * 0.5)


def glass_scene(seed=3, per_type=16, max_per_cell=4):
    """Divergence stress: the demo ground box plus 3*per_type ellipsoids / cylinders / cones, three
    quarters of them refractive, thin-lens camera.  Placements that would put more than
    `max_per_cell` objects into one 5x5x1 grid cell are redrawn, so the reference's cell-list
    growth defect (code/app.cpp:1329-1336) is never triggered by this scene."""
    rng = PCG32(0, seed)
    world = make_world(aperture=0.8, focal_distance=30.0)
    objs = [make_object((0, 0, 0.1), (50, 50, 0.01), PT_GEOM_BOX, color=(0.5, 0.5, 0.5))]
    occupancy = {}
    total = 3 * per_type
    glass = set(range(total)) - set(range(3, total, 4))    # 3 of every 4 are refractive
    k = 0
    for gtype in (PT_GEOM_SPHERE, PT_GEOM_CYLINDER, PT_GEOM_CONE):
        for _ in range(per_type):
            for _attempt in range(200):
                pos = (round(rng.uniform(-18, 18), 2), round(rng.uniform(-18, 18), 2),
                       round(rng.uniform(1, 12), 2))
                dim = (rng.uniform(1.5, 5), rng.uniform(1.5, 5), rng.uniform(1.5, 5))
                q = _rand_quat(rng)
                half = _aabb_half(dim, q) + 0.05
                lo = np.floor((np.asarray(pos) - half) / (5, 5, 1)).astype(int)
                hi = np.floor((np.asarray(pos) + half) / (5, 5, 1)).astype(int)
                cells = [(x, y, z) for x in range(lo[0], hi[0] + 1) for y in range(lo[1], hi[1] + 1)
                         for z in range(lo[2], hi[2] + 1)]
                # the ground box occupies every (x, y, 0) cell
                if all(occupancy.get(c, 1 if c[2] == 0 else 0) < max_per_cell for c in cells):
                    for c in cells:
                        occupancy[c] = occupancy.get(c, 1 if c[2] == 0 else 0) + 1
                    break
            col = (rng.uniform(0.5, 0.95), rng.uniform(0.5, 0.95), rng.uniform(0.5, 0.95))
            if k in glass:
                o = make_object(pos, dim, gtype, rot=q, color=col, ior=rng.uniform(1.2, 2.4))
            else:
                o = make_object(pos, dim, gtype, rot=q, color=col, rough=0.5)
            objs.append(o)
            k += 1
    return world, objs


def dense_scene(n=100000, seed=4, half_extent=200.0, height=100.0):
    """n mixed primitives (a quarter of each type) scattered in a slab; 10 % mirrors, 10 % glass
    (ior 1.5), 80 % diffuse, 1 % emissive.  Meant for a fine grid (grid_hint 64)."""
    rng = PCG32(0, seed)
    world = make_world(cam_pos=(0, -1.6 * half_extent, 0.9 * height), cam_rot=(0, -0.25, 0), fov=70.0,
                       plane_width=10.0)
    objs = []
    for i in range(n):
        gtype = (PT_GEOM_SPHERE, PT_GEOM_BOX, PT_GEOM_CYLINDER, PT_GEOM_CONE)[i % 4]
        pos = (rng.uniform(-half_extent, half_extent), rng.uniform(-half_extent, half_extent),
               rng.uniform(0, height))
        dim = (rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5))
        q = _rand_quat(rng) if rng.below(2) else (1, 0, 0, 0)
        col = (rng.uniform(0.3, 0.95), rng.uniform(0.3, 0.95), rng.uniform(0.3, 0.95))
        m = rng.below(10)
        emit = (1.0, 0.9, 0.7) if rng.below(100) == 0 else (0, 0, 0)
        if m == 0:
            o = make_object(pos, dim, gtype, rot=q, color=col, refl=1.0, emit=emit)
        elif m == 1:
            o = make_object(pos, dim, gtype, rot=q, color=col, ior=1.5, emit=emit)
        else:
            o = make_object(pos, dim, gtype, rot=q, color=col, emit=emit)
        objs.append(o)
    return world, objs


def mixed_scene(seed=7, n=40, glass_share=0.3, mirror_share=0.2, rough_share=0.2, spread=18.0):
    """Small all-types test scene over the demo ground box (used by the parity tests)."""
    rng = PCG32(0, seed)
    world = make_world()
    objs = [make_object((0, 0, 0.1), (50, 50, 0.01), PT_GEOM_BOX, color=(0.5, 0.5, 0.5))]
    for i in range(n):
        gtype = (PT_GEOM_SPHERE, PT_GEOM_BOX, PT_GEOM_CYLINDER, PT_GEOM_CONE)[i % 4]
        pos = (round(rng.uniform(-spread, spread), 2), round(rng.uniform(-spread, spread), 2),
               round(rng.uniform(1, 12), 2))
        even = rng.below(4) == 0
        d0 = rng.uniform(1.5, 5)
        dim = (d0, d0, d0) if even else (d0, rng.uniform(1.5, 5), rng.uniform(1.5, 5))
        q = (1, 0, 0, 0) if rng.below(3) == 0 else _rand_quat(rng)
        col = (rng.uniform(0.4, 0.95), rng.uniform(0.4, 0.95), rng.uniform(0.4, 0.95))
        u = rng.uniform(0, 1)
        emit = (0.9, 0.8, 0.6) if rng.below(12) == 0 else (0, 0, 0)
        if u < glass_share:
            o = make_object(pos, dim, gtype, rot=q, color=col, ior=rng.uniform(1.2, 2.4), emit=emit)
        elif u < glass_share + mirror_share:
            o = make_object(pos, dim, gtype, rot=q, color=col, refl=1.0 if rng.below(2) else 0.6, emit=emit)
        elif u < glass_share + mirror_share + rough_share:
            o = make_object(pos, dim, gtype, rot=q, color=col, rough=rng.uniform(0.2, 1.0), emit=emit)
        else:
            o = make_object(pos, dim, gtype, rot=q, color=col, emit=emit)
        objs.append(o)
    return world, objs


def plain_scene(seed=5, n=8):
    """Even spheres and unrotated boxes with opaque, smooth materials over the demo ground box: the kind of
    scene (like the demo scene itself) that the device renders with its leanest kernel variant."""
    rng = PCG32(0, seed)
    world = make_world()
    objs = [make_object((0, 0, 0.1), (50, 50, 0.01), PT_GEOM_BOX, color=(0.5, 0.5, 0.5))]
    for i in range(n):
        pos = (round(rng.uniform(-12, 12), 2), round(rng.uniform(-12, 12), 2), round(rng.uniform(1, 10), 2))
        col = (rng.uniform(0.4, 0.95), rng.uniform(0.4, 0.95), rng.uniform(0.4, 0.95))
        refl = (0.0, 1.0, 0.6)[rng.below(3)]
        emit = (0.9, 0.8, 0.6) if rng.below(6) == 0 else (0, 0, 0)
        if i % 2 == 0:
            d = rng.uniform(1.5, 5)
            objs.append(make_object(pos, (d, d, d), PT_GEOM_SPHERE, color=col, refl=refl, emit=emit))
        else:
            objs.append(make_object(pos, (rng.uniform(1.5, 5), rng.uniform(1.5, 5), rng.uniform(1.5, 5)), PT_GEOM_BOX, color=col, refl=refl, emit=emit))
    return world, objs


def aligned_boxes_scene():
    """Sixteen upright boxes on a lattice whose faces lie on multiples of 6 in x and y and on integers in z.  With
    grid_hint 14 the grid's cells are 6 x 6 x 1 (code/raycast.cpp:595-617: cell size = extent / hint, truncated), so every
    face lies on cell bounds: hit points sit exactly on a cell's lower or upper bound, or an ulp beside it, which is where
    the rule "the hit counts for the first visited cell whose closed box holds it" (code/raycast.cpp:957-983) is decided."""
    world = make_world(cam_pos=(3, -70, 22), cam_rot=(0, -0.3, 0), fov=80.0)
    objs = []
    for i in range(4):
        for j in range(4):
            x0, y0 = 24 * i - 42, 24 * j - 42
            top = 4 * (1 + (i + 2 * j) % 3)
            shade = 0.35 + 0.04 * (i + 4 * j)
            refl = 1.0 if (i + j) % 5 == 0 else 0.0
            objs.append(make_object((x0 + 6, y0 + 6, top / 2), (12, 12, top), PT_GEOM_BOX, color=(shade, 0.6, 1 - shade), refl=refl))
    return world, objs


def _axis_quat(axis, degrees):
    a = math.radians(degrees) / 2
    return (math.cos(a), axis[0] * math.sin(a), axis[1] * math.sin(a), axis[2] * math.sin(a))


def planes_small_scene():
    """The demo scene with the reference's fifth primitive added (GEOM_TYPE_PLANE, code/raycast.cpp:885-889): a ground
    plane under the ground box and a tilted mirror plane behind the sphere.  Fewer than ten entities: every cast
    tests every entity, the planes without a bounding-sphere test, and rays that miss the grid test the planes only."""
    world, objs = default_scene()
    objs.append(make_object((0, 0, -1.5), (1, 1, 1), PT_GEOM_PLANE, color=(0.6, 0.5, 0.4)))
    objs.append(make_object((0, 30, 10), (1, 1, 1), PT_GEOM_PLANE, rot=_axis_quat((1, 0, 0), 70.0), color=(0.8, 0.8, 0.9), refl=0.85))
    return world, objs


def planes_grid_scene(seed=13, n=30):
    """A grid scene (every kind and material) with three planes: one below the grid (its list entries are clamped into
    the bottom layer of cells, and it is what rays that miss the grid hit), one cutting through the grid at a slant
    (a mirror), one upright behind the cluster (diffuse, rough)."""
    world, objs = mixed_scene(seed=seed, n=n)
    objs.insert(3, make_object((0, 0, -2.0), (2, 2, 2), PT_GEOM_PLANE, color=(0.55, 0.6, 0.5)))
    objs.insert(11, make_object((2, 3, 4.0), (1, 1, 1), PT_GEOM_PLANE, rot=_axis_quat((0.6, 0.8, 0), 38.0), color=(0.85, 0.85, 0.85), refl=1.0))
    objs.append(make_object((0, 26, 0), (1, 1, 1), PT_GEOM_PLANE, rot=_axis_quat((1, 0, 0), 90.0), color=(0.4, 0.5, 0.7), rough=0.4))
    return world, objs
