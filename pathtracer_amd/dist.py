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
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        # several ranks sharing one GPU in a rehearsal: gloo moves host tensors; the device tensors are staged through
        # the host in stream order (the copies below run on the current stream and wait for it)
        host = mine.cpu()
        if rank == root:
            parts = [torch.empty(stride, dtype=mine.dtype) for _ in range(world_size)]
            dist.gather(host, parts, dst=root, group=group)
            for i, part in enumerate(parts):
                gathered[i * stride:(i + 1) * stride].copy_(part)
        else:
            dist.gather(host, None, dst=root, group=group)
        return
    if rank == root:
        views = [gathered[i * stride:(i + 1) * stride] for i in range(world_size)]
        dist.gather(mine, views, dst=root, group=group)
    else:
        dist.gather(mine, None, dst=root, group=group)


class FramePipeline:
    """Frames of a partitioned render with the gather off the critical path: the tile buffers come in a ring of
    `depth`, the traces of the following frames run on the main stream while the gather and the assemble of
    frame i run on a side stream.  Three buffers, not two: the trace kernel is persistent and holds every wavefront
    slot of its GPU until its frame ends, so the collective of frame i only gets onto the GPU in the tail of trace
    i+1; with two buffers trace i+2 would have to wait for it (and for the assemble) right there, with three it
    waits for the collective of frame i-1, which had a whole frame to finish.  `renderers` are this process's contexts
    (one per owned partition; one in the one-process-per-GPU case), `parts` their partition indices."""

    def __init__(self, renderers, parts, part_count, width, height, rank=0, world_size=1, group=None, keep_frames=0, depth=3):
        self.renderers, self.parts, self.part_count = renderers, parts, part_count
        self.width, self.height, self.rank, self.world_size, self.group = width, height, rank, world_size, group
        self.stride = partition_stride(width, height, part_count)
        # Two streams of the pipeline's own.  Not torch's current stream: the default stream has handle 0, which
        # pt_trace takes for "the context's own stream", and events recorded on the default stream would then
        # not cover the trace kernels at all.
        torch.cuda.init()                # Stream() does not initialise torch's device state by itself
        self.main = torch.cuda.Stream()
        self.side = torch.cuda.Stream()
        local = len(parts)
        self.depth = depth
        self.mine = [torch.zeros(local * self.stride, dtype=torch.float32, device="cuda") for _ in range(depth)]
        root = rank == 0
        self.gathered = [torch.empty(part_count * self.stride, dtype=torch.float32, device="cuda") if root else None for _ in range(depth)]
        self.image = torch.empty(height * width * 3, dtype=torch.float32, device="cuda") if root else None
        self.traced = [torch.cuda.Event() for _ in range(depth)]
        self.drained = [torch.cuda.Event() for _ in range(depth)]
        self.frame = 0
        # tests: a copy of every assembled frame, taken on the side stream without stopping the pipeline
        self.kept = [torch.empty(height * width * 3, dtype=torch.float32, device="cuda") for _ in range(keep_frames)] if root else []
        torch.cuda.synchronize()         # the buffers above were filled on the default stream

    def step(self):
        b = self.frame % self.depth
        if self.frame >= self.depth:
            self.main.wait_event(self.drained[b])          # the gather that read mine[b] `depth` frames ago
        for k, r in enumerate(self.renderers):
            r.trace(d_rgb=self.mine[b][k * self.stride:].data_ptr(), stream=self.main.cuda_stream, sync=False)
        self.traced[b].record(self.main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.traced[b])
            if self.world_size > 1:
                gather_parts(self.mine[b], self.gathered[b], self.stride, self.rank, self.world_size, group=self.group)
            else:                                           # one process owns several partitions: place them by index
                for k, p in enumerate(self.parts):
                    self.gathered[b][p * self.stride:(p + 1) * self.stride].copy_(self.mine[b][k * self.stride:(k + 1) * self.stride])
            if self.rank == 0:
                self.renderers[0].assemble(self.gathered[b].data_ptr(), self.stride, self.part_count, self.width, self.height,
                                           self.image.data_ptr(), self.side.cuda_stream)
                if self.frame < len(self.kept):
                    self.kept[self.frame].copy_(self.image, non_blocking=True)
            self.drained[b].record(self.side)
        self.frame += 1

    def finish(self):
        torch.cuda.synchronize()
        return self.image
