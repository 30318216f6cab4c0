"""SURVEY.md section 8e on the device: the reconstruction loop with the rays sharded by theta-block over two
ranks (both on this box's one GPU, gloo over host tensors -- the control flow of the RCCL run) gives the
wavespeed of the one-rank loop up to summation order.  examples/distributed_reconstruction.py is the caller."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env, timeout):
    r = subprocess.run(cmd, cwd=ROOT, env=env, timeout=timeout, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_two_rank_reconstruction_equals_one_rank(tmp_path):
    script = os.path.join(ROOT, "examples", "distributed_reconstruction.py")
    args = ["--ns", "48", "--nd", "64", "--N", "45", "--iters", "3", "--rank-thres", "2000"]
    env = dict(os.environ, RAY_SHARE_GPU="1", RAY_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    o1, o2 = str(tmp_path / "w1.npy"), str(tmp_path / "w2.npy")
    out1 = _run([sys.executable, script, *args, "--out", o1], env, 200)
    out2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                 "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script, *args, "--out", o2], env, 200)
    a, b = np.load(o1), np.load(o2)
    assert np.max(np.abs(a - b)) <= 1e-10 * np.max(np.abs(a))
    import json
    l1 = [json.loads(x) for x in out1.splitlines() if x.startswith("{")]
    l2 = [json.loads(x) for x in out2.splitlines() if x.startswith("{")]
    assert len(l1) == len(l2) == 3
    assert l2[-1]["world"] == 2 and l1[-1]["selected_rays"] == l2[-1]["selected_rays"]
    assert abs(l1[-1]["obj"] - l2[-1]["obj"]) <= 1e-12 * l1[-1]["obj"]
    assert l1[-1]["obj"] < l1[0]["obj"]
