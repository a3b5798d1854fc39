"""The multi-process path on CPU: world_size 2 and 3 over gloo.  Each rank lays a known image out as
its tile-major partition buffer (the layout the trace kernel writes), the ranks gather on rank 0
through pathtracer_amd.dist.gather_parts - the function bench.py uses over RCCL - and rank 0 puts the
image back together.  Covers partition sizes (from the C library), strides, gather and tile order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def synthetic_image(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x * 1.0 + y * 0.001, y * 1.0 + 0.5, (x * 31 + y * 17) % 97], axis=-1).astype(np.float32)


def to_partition(img, part, parts):
    """Tile-major buffer of one partition: [local tile][8x8 pixels][3], tiles t with t % parts == part."""
    h, w, _ = img.shape
    tx, ty = (w + 7) // 8, (h + 7) // 8
    tiles = [t for t in range(tx * ty) if t % parts == part]
    buf = np.zeros((len(tiles), 64, 3), np.float32)
    for lt, t in enumerate(tiles):
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        for j in range(64):
            x, y = x0 + (j & 7), y0 + (j >> 3)
            if x < w and y < h:
                buf[lt, j] = img[y, x]
    return buf.reshape(-1)


def from_partitions(gathered, stride, parts, w, h):
    img = np.zeros((h, w, 3), np.float32)
    tx = (w + 7) // 8
    for y in range(h):
        for x in range(w):
            t = (y >> 3) * tx + (x >> 3)
            part, local = t % parts, t // parts
            at = part * stride + (local * 64 + (y & 7) * 8 + (x & 7)) * 3
            img[y, x] = gathered[at:at + 3]
    return img


def traced_tiles(w, h, part, parts):
    """What a rank of a partitioned render produces, from the CPU oracle: the demo scene's pixels of the tiles t with
    t % parts == part (per-sample streams, so a pixel does not depend on who traces it), each tile rendered on its own;
    every other pixel stays zero.  parts = 1: the whole image."""
    import ctypes as C
    from oracle.oraclebind import OracleScene
    from pathtracer_amd import api, scenes
    world, objs = scenes.default_scene()
    st = api.default_settings(w, h, 4, 4, seed=99)
    sc = OracleScene(world, objs, st)
    img = np.zeros((h, w, 3), np.float32)
    tx, ty = (w + 7) // 8, (h + 7) // 8
    for t in range(tx * ty):
        if t % parts == part:
            x0, y0 = (t % tx) * 8, (t // tx) * 8
            sc.lib.pto_render_rect(sc.h, 0, api.PT_SEED_PER_SAMPLE, x0, y0, min(8, w - x0), min(8, h - y0), img.ctypes.data_as(C.POINTER(C.c_float)))
    sc.close()
    return img


def worker(rank, world, port, w, h, out, traced=False):
    sys.path.insert(0, ROOT)
    from pathtracer_amd import api
    from pathtracer_amd import dist as ptdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = api.load_core()
        # synthetic: every rank knows the whole image and cuts its share out; traced: a rank only has the tiles it rendered
        img = traced_tiles(w, h, rank, world) if traced else synthetic_image(w, h)
        stride = ptdist.partition_stride(w, h, world)
        mine_np = to_partition(img, rank, world)
        assert mine_np.size == lib.pt_partition_tiles(w, h, rank, world) * ptdist.TILE_FLOATS
        mine = torch.zeros(stride)
        mine[:mine_np.size] = torch.from_numpy(mine_np)
        gathered = torch.zeros(world * stride) if rank == 0 else None
        ptdist.gather_parts(mine, gathered, stride, rank, world)
        # timing protocol of bench.py: barrier, max over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        if rank == 0:
            back = from_partitions(gathered.numpy(), stride, world, w, h)
            whole = traced_tiles(w, h, 0, 1) if traced else img
            same = (back.view(np.uint32) == whole.view(np.uint32)).all() and (not traced or 0.0 < float(whole.mean()) < 1.0)
            np.save(out, np.array([float(same)]))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,w,h", [(2, 64, 40), (3, 100, 37), (2, 24, 24)])
def test_gather_of_tile_partitions(tmp_path, world, w, h):
    out = str(tmp_path / "ok.npy")
    mp.spawn(worker, args=(world, free_port(), w, h, out), nprocs=world, join=True)
    assert np.load(out)[0] == 1.0


@pytest.mark.parametrize("world,w,h", [(2, 64, 40), (3, 52, 37)])
def test_gather_of_traced_tile_partitions(tmp_path, world, w, h):
    """The same gather with rendered content: every rank traces its own tiles of the demo scene (CPU oracle), rank 0 puts the
    gathered partitions together and finds the image of an unpartitioned render, bit for bit."""
    out = str(tmp_path / "ok.npy")
    mp.spawn(worker, args=(world, free_port(), w, h, out, True), nprocs=world, join=True)
    assert np.load(out)[0] == 1.0


def test_partition_tiles_cover_the_image(core_lib):
    for w, h in [(1920, 1080), (3840, 2160), (640, 360), (100, 37), (8, 8), (1, 1)]:
        total = ((w + 7) // 8) * ((h + 7) // 8)
        for parts in (1, 2, 3, 4, 8):
            counts = [core_lib.pt_partition_tiles(w, h, p, parts) for p in range(parts)]
            assert sum(counts) == total and max(counts) - min(counts) <= 1
    assert core_lib.pt_partition_tiles(64, 64, 2, 2) < 0
