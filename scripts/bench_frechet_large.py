#!/usr/bin/env python3
"""ScatterForwardOperator at a size that goes through the chunked assembly: ns x nd rays on an N^2 grid,
M left on the device (its CSR would be tens of GB on the host).  usage: bench_frechet_large.py [ns nd N dt]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ray_amd
    ns, nd, N = (int(v) for v in (sys.argv[1:4] or (512, 512, 513)))
    dt = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-2
    ctx = ray_amd.Context(0)
    sp = ray_amd.SinProd.smooth_demo()
    p = np.zeros(N)
    ray_amd._lib.check(ctx._lib.ray_linspace(-1.5, 1.5, N, p.ctypes.data))
    sx, sy = np.sin(1.5 * np.pi * p), np.sin(1.5 * np.pi * p)
    c = np.asfortranarray(1 + 0.3 * sx[:, None] * sy[None, :])
    med = ray_amd.GridMedium(c, 1.5, ctx)
    m = ray_amd.ScatterRelation(med, None, ns, nd, 1e-2, ctx=ctx)
    for rep in range(2):
        ctx.reset_counters(); ctx.enable_timing(True)
        t0 = time.perf_counter()
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, 1.5, dt, ctx=ctx)
        t_all = time.perf_counter() - t0
        cnt = ctx.counters(); ctx.enable_timing(False)
        rows, cols, nnz = F.shape[0], F.shape[1], F.nnz
        # RAY_BENCH_NOMATVEC: ablation builds may leave the matrix unwritten; nothing may read it then
        y = np.zeros(1) if os.environ.get("RAY_BENCH_NOMATVEC") else F.matvec(np.ones(cols))
        F.close()
        if hasattr(ctx._lib, "ray_debug_fr_timeline"):        # instrumentation build (-DRAY_TIMELINE): k_frechet occupancy
            import ctypes as C
            nblk = (int(m.shape[0]) + 63) // 64
            tl = np.zeros(3 * nblk, dtype=np.uint64)
            ctx._lib.ray_debug_fr_timeline(tl.ctypes.data_as(C.c_void_p), nblk)
            tl = tl.reshape(nblk, 3).astype(np.int64)
            t0, t1 = tl[:, 0].min(), tl[:, 1].max()
            dur = tl[:, 1] - tl[:, 0]
            edges = np.linspace(t0, t1, 21)
            occ = [float(np.mean([(np.minimum(tl[:, 1], hi) - np.maximum(tl[:, 0], lo)).clip(0).sum() / (hi - lo)]))
                   for lo, hi in zip(edges[:-1], edges[1:])]
            print(json.dumps({"k_frechet_timeline": {"workgroups": nblk, "span_ticks": int(t1 - t0), "mean_resident": float(dur.sum() / (t1 - t0)),
                                                      "duration_ticks_min_mean_max": [int(dur.min()), float(dur.mean()), int(dur.max())],
                                                      "resident_by_twentieth": [round(o) for o in occ],
                                                      "start_ticks_percentiles_0_50_100": [int(np.percentile(tl[:, 0] - t0, q)) for q in (0, 50, 100)]}}), flush=True)
        if hasattr(ctx._lib, "ray_debug_fr_phases"):          # instrumentation build (-DRAY_FR_PHASES)
            import ctypes as C
            ph = (C.c_ulonglong * 16)()
            ctx._lib.ray_debug_fr_phases(ph)
            n = max(ph[0], 1)
            print(json.dumps({"merge_emit_phases_cycles_per_workgroup": {"workgroups": ph[0], "meta": ph[1] / n, "copy": ph[2] / n,
                                                                          "tiles": ph[3] / n, "tail": ph[4] / n},
                              "merge_sort_phases_cycles_per_workgroup": {"workgroups": ph[8], "perm_nsteps": ph[13] / max(ph[8], 1), "base_cells": ph[9] / max(ph[8], 1),
                                                                         "keys_first16": ph[10] / max(ph[8], 1), "rounds": ph[11] / max(ph[8], 1),
                                                                         "ranking": ph[12] / max(ph[8], 1)}}), flush=True)
        print(json.dumps({"workload": f"ScatterForwardOperator {ns}x{nd} rays, N={N}, dt={dt}", "call": rep,
                          "rays": int(m.shape[0]), "ray_steps": int(info["nsteps"].sum()), "nnz": int(nnz),
                          "csr_GB": nnz * 12 / 1e9, "kernel_ms": cnt["kernel_ms"], "launches": cnt["launches"],
                          "call_s": t_all, "ray_steps_per_s_call": float(info["nsteps"].sum() / t_all),
                          "bad_status": int((info["status"] & ~16).astype(bool).sum()),
                          # k_frechet: a wave of 64 consecutive rays steps until its longest ray is done
                          "live_lane_fraction_in_ray_order": float(info["nsteps"].sum() / (64.0 * np.add.reduceat(
                              np.maximum.reduceat(info["nsteps"], np.arange(0, info["nsteps"].size, 64)), [0])[0])),
                          "matvec_finite": bool(np.isfinite(y).all())}), flush=True)


if __name__ == "__main__":
    main()
