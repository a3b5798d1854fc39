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
