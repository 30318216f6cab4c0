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

--solver cg (what a 1024^2 grid needs -- BASELINE config 5 -- where the normal matrix cannot be formed): the same
loop MATRIX-FREE on device vectors: dofs, b = M' mismatch, diag(M'M) and the action w = M'(M x) come from the device
CSR (ray_frechet_*_dev), the regulariser is applied as a stencil, (M'M + penalty R) x = b is solved by
Jacobi-preconditioned CG (examples/matrix_free.py) with ONE all-reduce (w, N^2 doubles) per CG iteration, and the
new nodal values go back with ray_medium_update_grid_dev.  --solver dense is the small-grid form above.

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
    ap.add_argument("--solver", choices=["dense", "cg"], default="dense")
    ap.add_argument("--cg-rtol", type=float, default=1e-12)
    ap.add_argument("--cg-maxit", type=int, default=20000)
    ap.add_argument("--time-step", type=float, default=5e-2)          # NonObstacleDemo.jl:29
    ap.add_argument("--shoot-dt", type=float, default=1e-2)           # NonObstacleDemo.jl:28
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ray_amd
    from ray_amd.dist import theta_comb, allreduce_loop_sums, allreduce_fidelity
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
    penalty, rejection, decay, time_step = 5e-1, 1.0, 0.0, args.time_step            # NonObstacleDemo.jl:36-41
    # this rank's rays: the comb of sources rank, rank + world, ... (equal work per rank whatever the medium; the
    # loop only ever sums over rays, so no rank needs to know where its rows sit in a global m)
    b, e, stride, _ = theta_comb(ns, rank, world)
    ctx.set_option(ray_amd._lib.OPT_SOURCE_STRIDE, stride)
    m = ray_amd.ScatterRelation(sp, sp, ns, nd, args.shoot_dt, (0, np.pi), ctx=ctx, src_begin=b, src_end=e)   # this rank's rows
    ctx.set_option(ray_amd._lib.OPT_SOURCE_STRIDE, 1)
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
    if args.solver == "cg":
        return loop_matrix_free(args, ctx, sp, m, c, c0, R, Idx, Ldx, N, ext, h, penalty, rejection, decay, time_step,
                                rank, world, backend, local)
    RI = R[Idx][:, Idx].toarray()
    fidelity = np.zeros(N * N)
    med = ray_amd.GridMedium(c0, ext, ctx)
    target2 = float((m[:, 4:8] ** 2).sum())
    nI = Idx.size

    for it in range(args.iters):
        t0 = time.perf_counter()
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, ext, time_step, ctx=ctx, want_rho=False)       # NonObstacle.jl:333
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


def loop_matrix_free(args, ctx, sp, m, c, c0, R, Idx, Ldx, N, ext, h, penalty, rejection, decay, time_step,
                     rank, world, backend, local):
    """The loop of NonObstacle.jl:331-372 with every O(nnz) and O(N^2) pass on the device and no matrix product."""
    import torch
    import torch.distributed as dist
    import ray_amd
    from ray_amd import device as rdev
    from matrix_free import regularizer_apply, pcg
    dev = torch.device("cuda", local)
    stream = torch.cuda.Stream(device=dev)                  # one stream: library kernels, torch ops, collectives
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    def allreduce(t, op=None):
        op = op or dist.ReduceOp.SUM
        if world == 1:
            return t
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:                                               # gloo rehearsal: host copy (see bench.py)
            hcopy = t.cpu()
            dist.all_reduce(hcopy, op=op)
            t.copy_(hcopy)
        return t

    n2, num = N * N, m.shape[0]
    f64 = dict(dtype=torch.float64, device=dev)
    idx_d = torch.from_numpy(Idx).to(dev)
    diagR = torch.from_numpy(R.diagonal()).to(dev)
    c0_d = torch.from_numpy(np.ascontiguousarray(c0.reshape(-1, order="F"))).to(dev)       # node order i + j N
    med = ray_amd.GridMedium(c0, ext, ctx)
    fid_d = torch.zeros(n2, **f64)
    dofs_d = torch.zeros(max(num, 1), **f64)
    b_d, d_d, w_d, xfull = (torch.zeros(n2, **f64) for _ in range(4))
    target2 = float((m[:, 4:8] ** 2).sum())
    cv = c.reshape(-1, order="F")[Ldx]
    for it in range(args.iters):
        t0 = time.perf_counter()
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, ext, time_step, ctx=ctx, want_rho=False)        # NonObstacle.jl:333
        mism_d = torch.from_numpy((m[:, 4:8] - X).reshape(-1)).to(dev)                         # :336
        rdev.frechet_dofs_dev(F, fid_d.data_ptr(), dofs_d.data_ptr())                          # :340-342
        mask_d = (dofs_d[:num] <= args.rank_thres).to(torch.uint8)                             # :345-348
        rdev.frechet_matvec_t_dev(F, mism_d.data_ptr(), mask_d.data_ptr(), b_d.data_ptr())     # :354
        rdev.frechet_normal_diag_dev(F, mask_d.data_ptr(), d_d.data_ptr())                     # diag of :353
        head = torch.stack([torch.dot(mism_d, mism_d), torch.tensor(target2, **f64), mask_d.sum().to(torch.float64)])
        packed = allreduce(torch.cat([head, b_d, d_d]))                                        # the one SUM all-reduce
        sums, bI, dI = packed[:3].tolist(), packed[3:3 + n2][idx_d], packed[3 + n2:][idx_d] + penalty * diagR[idx_d]
        n_allreduce = 1

        def apply_A(pI):                                                                       # (M'M + penalty R) restricted to Idx
            nonlocal n_allreduce
            xfull.zero_()
            xfull[idx_d] = pI
            rdev.frechet_normal_matvec_dev(F, xfull.data_ptr(), mask_d.data_ptr(), w_d.data_ptr())
            allreduce(w_d)                                                                     # one per CG iteration
            n_allreduce += 1
            return w_d[idx_d] + penalty * regularizer_apply(xfull, N, h)[idx_d]

        xI, cg_it, cg_res = pcg(apply_A, bI, dI, args.cg_rtol, args.cg_maxit)                  # :356
        corr_d = torch.zeros(n2, **f64)
        corr_d[idx_d] = xI
        rdev.frechet_fidelity_update_dev(F, corr_d.data_ptr(), mism_d.data_ptr(), mask_d.data_ptr(), rejection, decay,
                                         fid_d.data_ptr())                                     # :359-367
        allreduce(fid_d, dist.ReduceOp.MAX)
        c0_d += corr_d                                                                         # :370
        stream.synchronize()
        F.close()
        ray_amd._lib.check(ctx._lib.ray_medium_update_grid_dev(med.handle, c0_d.data_ptr()))
        c0v = c0_d[torch.from_numpy(Ldx).to(dev)].cpu().numpy()
        err = float(np.linalg.norm(cv - c0v) / np.linalg.norm(cv))                             # :371
        objective = float(np.sqrt(sums[0] / sums[1]))                                          # :372
        if rank == 0:
            print(json.dumps({"iter": it, "world": world, "solver": "cg", "obj": objective, "err": err,
                              "selected_rays": int(sums[2]), "rank": float(fid_d.sum().item()), "cg_iterations": cg_it,
                              "cg_relres": cg_res, "allreduces": n_allreduce, "nnz": int(F.nnz),
                              "t_iteration": time.perf_counter() - t0}), flush=True)
    stream.synchronize()
    if rank == 0 and args.out:
        np.save(args.out, np.asfortranarray(c0_d.cpu().numpy().reshape(N, N, order="F")))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
