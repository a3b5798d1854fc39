"""Seeded ray sets for the intersection tests: a mix of rays aimed at the targets (so most of them
hit), axis-parallel rays, rays with a zero component, and a few un-normalised directions (those occur
in the reference after a partial-mirror bounce, code/raycast.cpp:1189-1191)."""
import numpy as np


def _unit(d):
    d = np.ascontiguousarray(d, np.float32)
    ln = np.sqrt((d * d).sum(1, dtype=np.float32)).astype(np.float32)
    ln[ln == 0] = 1
    return np.ascontiguousarray(d / ln[:, None], np.float32)


def rays(n, seed, spread=25.0, aim=None, aim_jitter=1.0, unnormalised=0):
    """n rays; `aim` (m,3) are target points, 3 of every 4 rays point at a jittered target."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    if aim is not None:
        aim = np.asarray(aim, np.float32).reshape(-1, 3)
        t = aim[rng.integers(0, aim.shape[0], n)] if aim.shape[0] != n else aim
        t = t + rng.uniform(-aim_jitter, aim_jitter, (n, 3)).astype(np.float32)
        sel = (np.arange(n) % 4) != 3
        d[sel] = (t - o)[sel]
    d = _unit(d)
    k = n // 32
    d[0 * k:1 * k, 0] = 0.0
    d[1 * k:2 * k, 1] = 0.0
    d[2 * k:3 * k, 2] = 0.0
    d[3 * k:3 * k + 8] = [0, 0, -1]
    d[3 * k + 8:3 * k + 16] = [1, 0, 0]
    d[3 * k + 16:3 * k + 24] = [0, -1, 0]
    d = _unit(d)
    if unnormalised:
        d[n // 2:n // 2 + unnormalised] *= np.float32(0.7)
    return o, d


def scene_rays(n, seed, objs, spread=25.0):
    centres = np.array([[o.pos[0], o.pos[1], o.pos[2]] for o in objs], np.float32)
    return rays(n, seed, spread, aim=centres, aim_jitter=2.0, unnormalised=n // 16)
