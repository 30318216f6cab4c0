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


def _lines(out):
    import json
    return [json.loads(x) for x in out.splitlines() if x.startswith("{")]


def test_matrix_free_update_equals_dense(tmp_path):
    """NonObstacle.jl:351-356 at the shipped size (100 x 100 rays, N = 75): the matrix-free solve -- action of M'M
    from the device CSR, Jacobi-PCG on device vectors -- gives the update of the dense normal-matrix solve to 1e-8
    (measured 1e-13), on one rank and sharded over two (one all-reduce per CG iteration)."""
    script = os.path.join(ROOT, "examples", "distributed_reconstruction.py")
    args = ["--ns", "100", "--nd", "100", "--N", "75", "--iters", "2"]
    env = dict(os.environ, RAY_SHARE_GPU="1", RAY_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    od, oc, o2 = (str(tmp_path / n) for n in ("dense.npy", "cg.npy", "cg2.npy"))
    _run([sys.executable, script, *args, "--solver", "dense", "--out", od], env, 300)
    lc = _lines(_run([sys.executable, script, *args, "--solver", "cg", "--out", oc], env, 300))
    a, b = np.load(od), np.load(oc)
    assert np.max(np.abs(a - b)) <= 1e-8 * np.max(np.abs(a))
    assert all(l["cg_relres"] <= 1e-12 and l["allreduces"] == l["cg_iterations"] + 1 for l in lc)
    l2 = _lines(_run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                      "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script, *args,
                      "--solver", "cg", "--out", o2], env, 400))
    c2 = np.load(o2)
    assert np.max(np.abs(a - c2)) <= 1e-8 * np.max(np.abs(a))
    assert l2[-1]["world"] == 2 and abs(l2[-1]["obj"] - lc[-1]["obj"]) <= 1e-10 * lc[-1]["obj"]


def test_c5_grid_1024_loop_runs_matrix_free():
    """BASELINE config 5's grid (1024^2), one rank's shard on one GPU: a few iterations of the full loop --
    forward operator with the Frechet matrix on the device, dofs / gradient / fidelity passes, the matrix-free
    normal-equation solve -- reduce the objective and the interior error (a 64 x 64 ray subset keeps it short;
    profiles/r2/c5_grid1024_matrix_free_256x256rays.jsonl is the larger run)."""
    script = os.path.join(ROOT, "examples", "distributed_reconstruction.py")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = _run([sys.executable, script, "--ns", "64", "--nd", "64", "--N", "1024", "--iters", "3", "--solver", "cg",
                "--cg-rtol", "1e-6", "--cg-maxit", "1500", "--time-step", "1e-2"], env, 600)
    l = _lines(out)
    assert len(l) == 3 and l[0]["nnz"] > 1e7
    assert l[2]["obj"] < l[1]["obj"] < l[0]["obj"] and l[2]["err"] < l[0]["err"]
