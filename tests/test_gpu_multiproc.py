"""One process per partition, as bench.py runs one per GPU: three processes share the one GPU, each traces its partition of
a frame, the tile buffers are gathered on rank 0 (gloo, host tensors: several ranks cannot share a GPU under RCCL) and
assembled there on the device.  The image must be the single-process image bit for bit, and the ranks' path counts must
add up to the frame's."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from pathtracer_amd import api, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partitions_traced_by_separate_processes(tmp_path):
    world = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frames.npy")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multiproc_worker.py"), str(rank), str(world), str(port), out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for rank in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    w_scene, objs = scenes.mixed_scene(seed=7, n=40)
    r = api.Renderer(0)
    whole = r.render(w_scene, objs, api.default_settings(200, 117, 16, 8, seed=4242))
    paths = r.stats().paths
    r.close()
    for frame in got[:3]:
        assert (frame.view(np.uint32) == whole.view(np.uint32)).all()
    assert got[3, 0, 0, 0] == np.float32(paths)


def test_bench_line_of_two_ranks_on_the_one_gpu():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per partition), rehearsed with two ranks
    on the one GPU over gloo: the line must come from rank 0 alone, count the whole frame's paths and name both ranks."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PT_BENCH_BACKEND="gloo", PT_BENCH_HANG_TRACE="150")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-secondary", "--no-prewarm"]
    try:
        done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=200)
    except subprocess.TimeoutExpired:
        pytest.fail("the two-rank bench hung")
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [json.loads(l) for l in done.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong"
    assert line["config"]["paths_per_frame"] == 139622400          # both partitions of the 1080p / 64 spp frame
    assert line["value"] > 0 and line["roofline"]["kernel_ms"] > 0
