"""One rank of the multi-process GPU test (tests/test_gpu_multiproc.py): prepares its partition of a frame on the one GPU,
traces it, and takes part in the gather on rank 0 - over gloo with host tensors here, because several ranks cannot share
one GPU under RCCL; the rest (partition settings, tile buffers, gather_parts, pt_assemble) is what bench.py runs per GPU.
usage: python multiproc_worker.py <rank> <world> <port> <out.npy>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import api, scenes  # noqa: E402
from pathtracer_amd import dist as ptdist  # noqa: E402

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
W, H = 200, 117
world_scene, objs = scenes.mixed_scene(seed=7, n=40)
st = api.default_settings(W, H, 16, 8, seed=4242, part_index=rank, part_count=world)
r = api.Renderer(0)
r.prepare(world_scene, objs, st)
stride = ptdist.partition_stride(W, H, world)
stream = torch.cuda.Stream()
frames = []
for frame in range(3):                      # the same frame three times: buffers are reused, the gather must not mix them
    mine = torch.full((stride,), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    r.trace(d_rgb=mine.data_ptr(), stream=stream.cuda_stream, sync=False)
    stream.synchronize()
    host = mine.cpu()
    gathered = torch.empty(world * stride, dtype=torch.float32) if rank == 0 else None
    ptdist.gather_parts(host, gathered, stride, rank, world)
    if rank == 0:
        dev = gathered.cuda()
        image = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        r.assemble(dev.data_ptr(), stride, world, W, H, image.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        frames.append(image.cpu().numpy().reshape(H, W, 3))
r.sync()
paths = torch.tensor([float(r.stats().paths)], dtype=torch.float64)
dist.all_reduce(paths)
if rank == 0:
    np.save(out, np.stack(frames + [np.full((H, W, 3), paths.item(), np.float32)]))
r.close()
dist.barrier()
dist.destroy_process_group()
