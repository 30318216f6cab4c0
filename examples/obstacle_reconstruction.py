#!/usr/bin/env python3
"""End-to-end check of the obstacle workflow: ObstacleDemo.jl:36-260 (the CALLER of the hot path)
written out in Python on top of ray_amd -- ScatterRelationObstacle (reflections) -> ray classification
(orthogonal hits, unbroken rays) -> reconstruction of the wavespeed outside the obstacle from the
unbroken rays with ScatterForwardOperator, the `perc` selection mask, M'v and the fidelity update on
the GPU, and a dense host solve for A \\ b.  Plotting and timers dropped.

    python examples/obstacle_reconstruction.py [--ns 50 --nd 300 --N 45 --iters 50]
    python examples/obstacle_reconstruction.py --stage-two      # ObstacleDemo.jl:42-71 only: draw the obstacle
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nonobstacle_reconstruction import interpolation, regularization  # noqa: E402  (NonObstacle.jl:245-281)


def reconstruct(ns, nd, N, iters, ctx, log=print):
    import ray_amd
    sp, ob = ray_amd.SinProd.obstacle_demo(), ray_amd.Circle(0.3)
    fine_dt, time_step, ext = 1e-2, 5e-2, 1.5                                        # ObstacleDemo.jl:41-42,141
    m = ray_amd.ScatterRelationObstacle(sp, sp, ob, ob, ns, nd, fine_dt, ctx=ctx)    # :43
    cls = ray_amd.classify_rays(m, ns, nd, ctx=ctx)                                  # :48-67
    m[:, 8] *= fine_dt                                                               # :122
    m = np.asfortranarray(m[cls["unbroken"]])                                        # :124
    target = m[:, 4:8].reshape(-1)                                                   # :125
    penalty, rejection, decay = 5e-1, 5e-2, 10.0                                     # :142-144
    p = np.zeros(N)
    ray_amd._lib.check(ctx._lib.ray_linspace(-ext, ext, N, p.ctypes.data))
    h = p[1] - p[0]
    c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])          # :166
    r2 = p[:, None] ** 2 + p[None, :] ** 2
    inside_ob = (np.sqrt(r2) - 0.3 < 0).reshape(-1, order="F")                       # :167 (the circle is symmetric)
    Ldx = np.flatnonzero((r2 < 1).reshape(-1, order="F") & ~inside_ob)               # :170-171,185
    Idx = np.flatnonzero((r2 <= (1 + 2 * h) ** 2).reshape(-1, order="F"))            # :175-176
    R = regularization(h, N)
    RI = R[Idx][:, Idx].toarray()
    c0 = np.asfortranarray(0.8 * np.ones((N, N)))                                    # :189
    fidelity = np.zeros(N * N)
    med = ray_amd.GridMedium(c0, ext, ctx)
    hist = []
    for it in range(iters):
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, ext, time_step, ctx=ctx, want_rho=False)      # :199
        obs = np.asfortranarray(m.copy())
        obs[:, 4:8] = X
        mf = ray_amd.misfit(m, obs, 0.2, ctx=ctx)                                    # :202,217-226 on the GPU
        mismatch, mask = mf["mismatch"], mf["mask"]
        rows = np.flatnonzero(np.repeat(mask, 4))                                    # :228-229
        Msel = F.tocsr()[rows][:, Idx]                                               # :231
        A = (Msel.T @ Msel).toarray() + penalty * RI                                 # :233
        b = F.rmatvec(mismatch, mask)[Idx]                                           # :234 (device)
        corr = np.zeros(N * N)
        corr[Idx] = np.linalg.solve(A, b)                                            # :236
        fidelity, _ = F.fidelity_update(corr, mismatch, fidelity, rejection, decay, mask)   # :239-247
        F.close()
        c0 = np.asfortranarray(c0 + corr.reshape(N, N, order="F"))                   # :250
        med.update(c0)
        cv, c0v = c.reshape(-1, order="F")[Ldx], c0.reshape(-1, order="F")[Ldx]
        error = np.linalg.norm(cv - c0v) / np.linalg.norm(cv)                        # :251
        objective = np.linalg.norm(mismatch) / np.linalg.norm(target)                # :252
        hist.append({"iter": it, "rank": float(fidelity.sum()), "obj": float(objective), "err": float(error),
                     "selected_rays": int(mask.sum()), "rays": int(m.shape[0]),
                     "ceased_or_oob": int((info["status"] != 0).sum())})
        log(json.dumps(hist[-1]))
        if objective < 1e-2:                                                         # :259
            break
    return c0, hist, cls


def draw_obstacle(ns, nd, ctx, N=257):
    """Stage two of ObstacleDemo.jl (:42-71): the rays that return to their source with reversed momentum hit the
    obstacle at normal incidence half way through their time; traced that long without the obstacle, their end points
    draw it.  Returns the end points (k, 2) and their level-set values."""
    import ray_amd
    sp, ob = ray_amd.SinProd.obstacle_demo(), ray_amd.Circle(0.3)
    dt, ext = 5e-3, 1.5                                                              # accurateTimeStep, :42
    m = ray_amd.ScatterRelationObstacle(sp, sp, ob, ob, ns, nd, dt, ctx=ctx)         # :43 (t counted in steps)
    ortho = ray_amd.classify_rays(m, ns, nd, ctx=ctx)["ortho_index"]                 # :48-49
    m[:, 8] *= dt / 2                                                                # :50: half the time, in seconds
    p = np.zeros(N)
    ray_amd._lib.check(ctx._lib.ray_linspace(-ext, ext, N, p.ctypes.data))
    c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])
    ends = ray_amd.NonReflectionTrace(c, np.asfortranarray(m[ortho]), ext, dt, ctx=ctx)[:, 4:6]   # :51 (the plot's end points)
    return ends, np.hypot(ends[:, 0], ends[:, 1]) - 0.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ns", type=int, default=50)
    ap.add_argument("--nd", type=int, default=300)
    ap.add_argument("--N", type=int, default=45)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--stage-two", action="store_true", help="only draw the obstacle from the orthogonal hits")
    args = ap.parse_args()
    import ray_amd
    ctx = ray_amd.Context(0)
    if args.stage_two:
        ends, lev = draw_obstacle(args.ns, args.nd, ctx)
        print(json.dumps({"orthogonal_hits": int(ends.shape[0]), "max_abs_level_set_at_end_points": float(np.abs(lev).max()),
                          "median_abs_level_set": float(np.median(np.abs(lev)))}))
        return
    c0, hist, cls = reconstruct(args.ns, args.nd, args.N, args.iters, ctx)
    print(json.dumps({"orthogonal_hits": int(cls["ortho_index"].size), "unbroken_rays": int(cls["unbroken"].size)}))


if __name__ == "__main__":
    main()
