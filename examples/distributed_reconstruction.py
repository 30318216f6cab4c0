#!/usr/bin/env python3
"""The reference's reconstruction loop (NonObstacle.jl:283-381) with the rays sharded by theta-block over the
ranks of torch.distributed (one process per GPU; backend nccl = RCCL on a multi-GPU node):

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
        examples/distributed_reconstruction.py [--ns 100 --nd 100 --N 75 --iters 6 --out c0.npy]

Every rank shoots and traces its own sources and builds its own rows of M on its GPU (ScatterRelation,
ScatterForwardOperator, dofs, M'v, fidelity passes: the hot path).  Per iteration the ranks exchange ONE
packed SUM all-reduce -- {sum mismatch^2, sum target^2, b = M' mismatch, the normal matrix M'M over the
selected rays} (ray_amd.dist.allreduce_loop_sums) -- and one MAX all-reduce of the fidelity vector; then
every rank solves the same small dense system and applies the same update.  No ray data moves.

The ray selection `dofs <= rankThres` (NonObstacle.jl:345-348: sort, then searchsortedlast) is a per-ray
test, so it needs no exchange.  A 1-rank run of this script is the loop of
examples/nonobstacle_reconstruction.py; runs with different rank counts agree to summation order.

Rehearsal on a 1-GPU box: RAY_SHARE_GPU=1 puts every rank on device 0 and RAY_BACKEND=gloo replaces RCCL
(host tensors)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ns", type=int, default=100)
    ap.add_argument("--nd", type=int, default=100)
    ap.add_argument("--N", type=int, default=75)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--rank-thres", type=float, default=3600)
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ray_amd
    from ray_amd.dist import theta_block, allreduce_loop_sums, allreduce_fidelity
    from nonobstacle_reconstruction import regularization, interpolation

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("RAY_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RAY_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    xdev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")   # where the exchanged sums live

    ctx = ray_amd.Context(local)
    sp = ray_amd.SinProd.smooth_demo()
    ns, nd, N, ext = args.ns, args.nd, args.N, 1.5
    penalty, rejection, decay, time_step = 5e-1, 1.0, 0.0, 5e-2                      # NonObstacleDemo.jl:36-41
    b, e = theta_block(ns, rank, world)
    m = ray_amd.ScatterRelation(sp, sp, ns, nd, 1e-2, (0, np.pi), ctx=ctx, src_begin=b, src_end=e)   # this rank's rows
    num = m.shape[0]

    p = np.zeros(N)
    ray_amd._lib.check(ctx._lib.ray_linspace(-ext, ext, N, p.ctypes.data))
    h = p[1] - p[0]
    c = np.asfortranarray([[sp(x, y) for y in p] for x in p])
    r2 = p[:, None] ** 2 + p[None, :] ** 2
    Ldx = np.flatnonzero((r2 < 1).reshape(-1, order="F"))
    Idx = np.flatnonzero((r2 <= (1 + 2 * h) ** 2).reshape(-1, order="F"))
    R = regularization(h, N)
    c0 = interpolation(R, np.asfortranarray(np.where(r2 < 1, 0.0, c)), Ldx, N)
    RI = R[Idx][:, Idx].toarray()
    fidelity = np.zeros(N * N)
    med = ray_amd.GridMedium(c0, ext, ctx)
    target2 = float((m[:, 4:8] ** 2).sum())
    nI = Idx.size

    for it in range(args.iters):
        t0 = time.perf_counter()
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, ext, time_step, ctx=ctx)       # NonObstacle.jl:333
        mismatch = (m[:, 4:8] - X).reshape(-1)                                                 # :336
        dofs = F.dofs(fidelity)                                                                # :340-342
        mask = (dofs <= args.rank_thres).astype(np.uint8)                                      # :345-348
        rows = np.flatnonzero(np.repeat(mask, 4))
        b_loc = F.rmatvec(mismatch, mask)[Idx]                                                 # :354 (device)
        Msel = F.tocsr()[rows][:, Idx]                                                         # :351
        A_loc = (Msel.T @ Msel).toarray()                                                      # :353, this rank's rays
        t_gpu = time.perf_counter() - t0
        sums = torch.tensor([float(mismatch @ mismatch), target2], dtype=torch.float64, device=xdev)
        tb = torch.from_numpy(b_loc).to(xdev)
        tA = torch.from_numpy(A_loc).to(xdev)
        sel = torch.tensor([float(mask.sum())], dtype=torch.float64, device=xdev)
        objective = allreduce_loop_sums(sums, tb, tA, sel)                                     # the one SUM all-reduce
        A = tA.cpu().numpy() + penalty * RI
        corr = np.zeros(N * N)
        corr[Idx] = np.linalg.solve(A, tb.cpu().numpy())                                       # :356, same on every rank
        fidelity, _ = F.fidelity_update(corr, mismatch, fidelity, rejection, decay, mask)      # :359-367, this rank's rays
        tf = torch.from_numpy(fidelity).to(xdev)
        fidelity = allreduce_fidelity(tf).cpu().numpy()                                        # max over all rays
        F.close()
        c0 = np.asfortranarray(c0 + corr.reshape(N, N, order="F"))                             # :370
        med.update(c0)
        cv, c0v = c.reshape(-1, order="F")[Ldx], c0.reshape(-1, order="F")[Ldx]
        err = float(np.linalg.norm(cv - c0v) / np.linalg.norm(cv))                             # :371
        if rank == 0:
            print(json.dumps({"iter": it, "world": world, "obj": objective, "err": err,
                              "selected_rays": int(sel.item()), "rank": float(fidelity.sum()),
                              "t_rank_gpu_part": t_gpu}), flush=True)
    if rank == 0 and args.out:
        np.save(args.out, c0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
