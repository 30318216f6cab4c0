#!/usr/bin/env python3
"""Timing of the complete ScatterForwardOperator (SURVEY.md section 8f NEXT-1) on the shipped
reconstruction size (NonObstacleDemo.jl: 100 x 100 rays, N = 75, timeStep 5e-2) and a larger
one; the CPU oracle (C port of ChunkProcessing!, dense M like the reference) timed beside it on
a bounded sample."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import orc
    import ray_amd
    ctx = ray_amd.Context(0)
    sp = ray_amd.SinProd.smooth_demo()
    for ns, nd, N, dt in [(100, 100, 75, 5e-2), (256, 256, 257, 2e-2)]:
        m = ray_amd.ScatterRelation(sp, sp, ns, nd, 1e-2, ctx=ctx)
        c = orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.5)
        med = ray_amd.GridMedium(c, 1.5, ctx)
        ray_amd.forward_operator_dev(med, m[:256], 1.5, dt, ctx=ctx)[0].close()     # warm-up
        ctx.reset_counters(); ctx.enable_timing(True)
        t0 = time.perf_counter()
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, 1.5, dt, ctx=ctx)
        t_all = time.perf_counter() - t0
        cnt = ctx.counters(); ctx.enable_timing(False)
        t0 = time.perf_counter()
        M = F.tocsr()
        t_csr = time.perf_counter() - t0
        F.close()
        # the same call again, as the reconstruction loop makes it every iteration: workspaces and the
        # CSR arrays of the handle just closed are reused
        t0 = time.perf_counter()
        F2, _, _, _ = ray_amd.forward_operator_dev(med, m, 1.5, dt, ctx=ctx)
        t_again = time.perf_counter() - t0
        F2.close()
        out = {"workload": f"ScatterForwardOperator {ns}x{nd} rays, N={N}, dt={dt}", "rays": int(m.shape[0]),
               "ray_steps": int(info["nsteps"].sum()), "kernel_ms": cnt["kernel_ms"],
               "kernel_ray_steps_per_s": info["nsteps"].sum() / (cnt["kernel_ms"] * 1e-3),
               "call_s_incl_upload_and_allocs": t_all, "call_s_repeated": t_again, "csr_download_s": t_csr, "nnz": int(M.nnz)}
        # CPU oracle on a bounded sample (per-ray 4 x N^2 accumulator, as the reference's Phi)
        k = min(m.shape[0], 4096)
        tab = orc.GridTables(c, 1.5, 1)
        t0 = time.perf_counter()
        _, _, _, _, nso = orc.forward_operator(tab, np.asfortranarray(m[:k]), dt, dense=False, nthreads=16)
        t_cpu = time.perf_counter() - t0
        out["cpu_oracle_ray_steps_per_s"] = float(nso.sum() / t_cpu)
        out["cpu_sample"] = f"first {k} rays, 16 threads, {t_cpu:.2f} s"
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
