#!/usr/bin/env python3
"""The loop passes over a device-resident Frechet matrix (NEXT-2), for a kernel trace: builds M for ns x nd rays on
an N^2 grid and applies y = M x, b = M' v, w = M'(M x), d = diag(M'M) a few times each (host vectors: the kernels
are what the trace is for).  usage: bench_passes.py [ns nd N dt reps]"""
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
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    ctx = ray_amd.Context(0)
    p = np.zeros(N)
    ray_amd._lib.check(ctx._lib.ray_linspace(-1.5, 1.5, N, p.ctypes.data))
    s1 = np.sin(1.5 * np.pi * p)
    med = ray_amd.GridMedium(np.asfortranarray(1 + 0.3 * s1[:, None] * s1[None, :]), 1.5, ctx)
    m = ray_amd.ScatterRelation(med, None, ns, nd, 1e-2, ctx=ctx)
    F, X, _, info = ray_amd.forward_operator_dev(med, m, 1.5, dt, ctx=ctx, want_rho=False)
    rng = np.random.default_rng(3)
    x = rng.normal(size=N * N) * 1e-2
    v = rng.normal(size=F.shape[0])
    mask = (rng.uniform(size=F.shape[0] // 4) < 0.7).astype(np.uint8)
    out = {"rays": int(m.shape[0]), "entries": F.block_rows_dev()[0], "N": N}
    for name, fn in (("matvec", lambda: F.matvec(x)), ("matvec_t", lambda: F.rmatvec(v)), ("normal_matvec", lambda: F.normal_matvec(x)),
                     ("normal_matvec_masked", lambda: F.normal_matvec(x, mask)), ("normal_diag", lambda: F.normal_diag())):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        out[name + "_call_ms_host_vectors"] = (time.perf_counter() - t0) / reps * 1e3
        assert np.isfinite(r).all()
    # consistency of the passes among themselves (the GPU tests hold them against numpy)
    y = F.matvec(x)
    w = F.normal_matvec(x)
    out["normal_vs_two_passes_rel"] = float(np.max(np.abs(w - F.rmatvec(y))) / np.max(np.abs(w)))
    F.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
