#!/usr/bin/env python3
"""End-to-end check: the reference's reconstruction loop (the CALLER of the hot path, out of scope
as product code -- SURVEY.md section 8b) written out in Python on top of ray_amd, so that the whole chain
ScatterRelation -> ScatterForwardOperator (X, M on the GPU) -> dofs / M'v / fidelity passes ->
host normal-equation solve can be run and seen to converge.

Follows NonObstacleDemo.jl:26-44 and NonObstacleReconstruction (NonObstacle.jl:283-381) line by
line; plotting and timers dropped.  `A \\ b` is a dense host solve (numpy) here, CHOLMOD there.

    python examples/nonobstacle_reconstruction.py [--ns 100 --nd 100 --N 75 --iters 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def regularization(h, N):
    """NonObstacle.jl:245-274."""
    ii, jj = np.meshgrid(np.arange(3, N - 2), np.arange(3, N - 2), indexing="ij")     # i,j = 3..N-3 (1-based)
    node = (ii + (jj - 1) * N).reshape(-1) - 1                                         # 0-based
    n2 = N * N
    ones = np.full(node.size, 1.0 / h)
    gx = sps.csr_matrix((np.concatenate([ones, -ones]), (np.concatenate([node, node]), np.concatenate([node, node - 1]))), shape=(n2, n2))
    gy = sps.csr_matrix((np.concatenate([ones, -ones]), (np.concatenate([node, node]), np.concatenate([node, node + N]))), shape=(n2, n2))
    return (gx.T @ gx + gy.T @ gy).tocsr()


def interpolation(R, c0, Ldx, N):
    """NonObstacle.jl:276-281: harmonic fill of the unknown interior."""
    z = c0.reshape(-1, order="F").copy()
    b = -(R @ z)
    z[Ldx] = sps.linalg.spsolve(R[Ldx][:, Ldx].tocsc(), b[Ldx])
    return np.asfortranarray(z.reshape(N, N, order="F"))


def reconstruct(m, N, ext, penalty, rejection, decay, rank_thres, wave_speed, time_step, iters, ctx, log=print):
    import ray_amd
    num = m.shape[0]
    target = m[:, 4:8].reshape(-1)                                                   # :287 (ray-major)
    p = np.zeros(N)
    ray_amd._lib.check(ctx._lib.ray_linspace(-ext, ext, N, p.ctypes.data))           # :290
    h = p[1] - p[0]
    c = np.asfortranarray([[wave_speed(x, y) for y in p] for x in p])                # :306
    r2 = p[:, None] ** 2 + p[None, :] ** 2
    Ldx = np.flatnonzero((r2 < 1).reshape(-1, order="F"))                            # :307-308
    Idx = np.flatnonzero((r2 <= (1 + 2 * h) ** 2).reshape(-1, order="F"))            # :312-313
    c0 = np.asfortranarray(np.where(r2 < 1, 0.0, c))                                 # :310
    R = regularization(h, N)
    c0 = interpolation(R, c0, Ldx, N)                                                # :320
    fidelity = np.zeros(N * N)
    RI = R[Idx][:, Idx].toarray()
    med = ray_amd.GridMedium(c0, ext, ctx)
    hist = []
    for it in range(iters):
        t0 = time.perf_counter()
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, ext, time_step, ctx=ctx, want_rho=False)   # :333
        t_forward = time.perf_counter() - t0
        mismatch = (m[:, 4:8] - X).reshape(-1)                                       # :336
        t0 = time.perf_counter()
        dofs = F.dofs(fidelity)                                                      # :340-342
        t_dof = time.perf_counter() - t0
        perm = np.argsort(dofs, kind="stable")                                       # :345
        trunc = np.searchsorted(dofs[perm], rank_thres, side="right")                # :346
        mask = np.zeros(num, dtype=np.uint8)
        mask[perm[:trunc]] = 1
        rows = np.flatnonzero(np.repeat(mask, 4))
        t0 = time.perf_counter()
        Msel = F.tocsr()[rows][:, Idx]                                               # :351
        A = (Msel.T @ Msel).toarray() + penalty * RI                                 # :353
        b = F.rmatvec(mismatch, mask)[Idx]                                           # :354 (device)
        corr = np.zeros(N * N)
        corr[Idx] = np.linalg.solve(A, b)                                            # :356
        t_solv = time.perf_counter() - t0
        t0 = time.perf_counter()
        fidelity, _ = F.fidelity_update(corr, mismatch, fidelity, rejection, decay, mask)   # :359-367
        t_fid = time.perf_counter() - t0
        F.close()
        c0 = np.asfortranarray(c0 + corr.reshape(N, N, order="F"))                   # :370
        med.update(c0)
        cv, c0v = c.reshape(-1, order="F")[Ldx], c0.reshape(-1, order="F")[Ldx]
        error = np.linalg.norm(cv - c0v) / np.linalg.norm(cv)                        # :371
        objective = np.linalg.norm(mismatch) / np.linalg.norm(target)                # :372
        hist.append({"iter": it, "rank": float(fidelity.sum()), "obj": float(objective), "err": float(error),
                     "selected_rays": int(trunc), "t_forward": t_forward, "t_dof": t_dof, "t_solve": t_solv, "t_fid": t_fid})
        log(json.dumps(hist[-1]))
        if objective < 1e-2:                                                         # :379
            break
    return c0, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ns", type=int, default=100)
    ap.add_argument("--nd", type=int, default=100)
    ap.add_argument("--N", type=int, default=75)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rank-thres", type=float, default=3600)
    args = ap.parse_args()
    import ray_amd
    ctx = ray_amd.Context(0)
    sp = ray_amd.SinProd.smooth_demo()
    m = ray_amd.ScatterRelation(sp, sp, args.ns, args.nd, 1e-2, (0, np.pi), ctx=ctx)         # NonObstacleDemo.jl:32
    reconstruct(m, args.N, 1.5, 5e-1, 1.0, 0.0, args.rank_thres, sp, 5e-2, args.iters, ctx)    # :36-44


if __name__ == "__main__":
    main()
