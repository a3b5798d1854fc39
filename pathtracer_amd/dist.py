"""Multi-GPU plumbing: one process per GPU, torch.distributed (RCCL on the GPU box, gloo in the CPU
tests).  The render path has exactly one inter-GPU step, the gather of the partitions' tile buffers
on the root rank; everything here is about that step."""
import torch
import torch.distributed as dist

from . import api

TILE_FLOATS = 8 * 8 * 3


def partition_stride(width, height, part_count):
    """Floats of the largest partition buffer: the common stride of the gathered buffers."""
    lib = api.load_core()
    return max(lib.pt_partition_tiles(width, height, p, part_count) for p in range(part_count)) * TILE_FLOATS


def gather_parts(mine, gathered, stride, rank, world_size, root=0, group=None):
    """Every rank contributes `mine` (stride floats); root receives them side by side in `gathered`
    (world_size * stride floats).  One collective, no other data-path communication."""
    if world_size == 1:
        gathered[:stride].copy_(mine)
        return
    if rank == root:
        views = [gathered[i * stride:(i + 1) * stride] for i in range(world_size)]
        dist.gather(mine, views, dst=root, group=group)
    else:
        dist.gather(mine, None, dst=root, group=group)
