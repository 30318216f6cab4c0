"""world_size-2 gloo test (CPU) of the multi-GPU path: theta-block sharding and the single
all-reduce of the misfit sums.  The per-block compute is the CPU oracle here (the product has no
CPU path); on the GPU box the same functions run over RCCL with ray_amd.ScatterRelation blocks
(bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_theta_block_partitions():
    from ray_amd.dist import theta_block
    for ns in (1, 7, 50, 2048, 8192):
        for world in (1, 2, 3, 4, 8):
            blocks = [theta_block(ns, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == ns
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        theta_block(8, 2, 2)


def test_theta_comb_partitions():
    from ray_amd.dist import theta_comb
    for ns in (1, 7, 50, 2048, 8192):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                b, e, stride, count = theta_comb(ns, r, world)
                own = list(range(b, e, stride))
                assert len(own) == count and stride == world
                seen += own
            assert sorted(seen) == list(range(ns))
    with pytest.raises(ValueError):
        theta_comb(8, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import orc
    from ray_amd.dist import sharded_scatter_relation, allreduce_misfit
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ns, nd, dt = 9, 20, 1e-2                               # ragged: 9 sources over 2 ranks -> 5 + 4
    shoot = lambda b, e: orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, dt, src_begin=b, src_end=e)[0]
    m, r0, r1 = sharded_scatter_relation(shoot, ns, nd, rank, world)
    obs = np.asfortranarray(m.copy())
    obs[:, 4:8] *= 1.0 + 1e-3 * (1 + rank)
    _, out, _ = orc.misfit(m, obs, 0.2)
    sums = torch.tensor(out, dtype=torch.float64)
    objective = allreduce_misfit(sums)
    q.put((rank, r0, r1, m, obs, sums.tolist(), objective))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sweep_matches_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ns, nd, dt = 9, 20, 1e-2
    full = orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, dt)[0]
    got = np.vstack([r[3] for r in res])
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == ns * nd
    assert np.array_equal(got, full)                       # slabs tile the global m exactly
    obs = np.asfortranarray(np.vstack([r[4] for r in res]))
    _, out, _ = orc.misfit(np.asfortranarray(got), obs, 0.2)
    for r in res:                                           # every rank holds the global sums
        assert abs(r[5][0] - out[0]) <= 1e-12 * out[0] and abs(r[5][1] - out[1]) <= 1e-12 * out[1]
        assert abs(r[6] - np.sqrt(out[0] / out[1])) <= 1e-12


def _worker_loop(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scipy.sparse as sps
    import torch
    import torch.distributed as dist
    import orc
    from ray_amd.dist import theta_block, allreduce_loop_sums, allreduce_fidelity
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ns, nd, N = 7, 12, 21
    b0, e0 = theta_block(ns, rank, world)
    m = np.asfortranarray(orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, 1e-2, src_begin=b0, src_end=e0)[0])
    c0 = np.asfortranarray(1.0 + 0.5 * (orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.5) - 1.0))
    X, rho, M, st, _ = orc.forward_operator(orc.GridTables(c0, 1.5), m, 5e-2)
    mism = (m[:, 4:8] - X).reshape(-1)
    sums = torch.tensor([float(mism @ mism), float((m[:, 4:8] ** 2).sum())], dtype=torch.float64)
    b = torch.from_numpy(M.T @ mism)                       # this rank's rays only
    A = torch.from_numpy(np.ascontiguousarray(M.T @ M))
    objective = allreduce_loop_sums(sums, b, A)            # ONE collective: sums + gradient + normal matrix
    fid = orc.loop_fidelity(sps.csr_matrix(M), np.zeros(N * N), mism, np.arange(M.shape[0]), np.zeros(N * N), 1.0, 0.5)[0]
    fid = allreduce_fidelity(torch.from_numpy(fid))
    q.put((rank, sums.tolist(), b.numpy().copy(), A.numpy().copy(), fid.numpy().copy(), objective))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_sums_one_collective():
    """NEXT-2 over ranks: the gradient b = M' mismatch, the normal matrix and the misfit sums of a
    theta-sharded iteration arrive through ONE all-reduce and equal the single-process values."""
    import scipy.sparse as sps
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_loop, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ns, nd, N = 7, 12, 21
    m = np.asfortranarray(orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, 1e-2)[0])
    c0 = np.asfortranarray(1.0 + 0.5 * (orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.5) - 1.0))
    X, rho, M, st, _ = orc.forward_operator(orc.GridTables(c0, 1.5), m, 5e-2)
    mism = (m[:, 4:8] - X).reshape(-1)
    b, A = M.T @ mism, M.T @ M
    fid = orc.loop_fidelity(sps.csr_matrix(M), np.zeros(N * N), mism, np.arange(M.shape[0]), np.zeros(N * N), 1.0, 0.5)[0]
    for r in res:
        assert abs(r[1][0] - mism @ mism) <= 1e-12 * (mism @ mism)
        assert np.max(np.abs(r[2] - b)) <= 1e-12 * np.max(np.abs(b))
        assert np.max(np.abs(r[3] - A)) <= 1e-12 * np.max(np.abs(A))
        assert np.array_equal(r[4], fid)                   # max is exact
        assert abs(r[5] - np.sqrt((mism @ mism) / (m[:, 4:8] ** 2).sum())) <= 1e-12
