#!/usr/bin/env python3
"""Cell-cache statistics of the headline workload with an instrumentation build
(hipcc -DRAY_STATS ... -o exp/STATS.so; RAY_AMD_LIB=exp/STATS.so python scripts/cache_stats.py)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_amd  # noqa: E402

ctx = ray_amd.Context(0)
sp = ray_amd.SinProd.smooth_demo()
N = 512
p = np.zeros(N)
ray_amd._lib.check(ctx._lib.ray_linspace(-1.5, 1.5, N, p.ctypes.data))
c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])
med = ray_amd.GridMedium(c, 1.5, ctx)
out = (C.c_ulonglong * 12)()
ctx._lib.ray_debug_stats(out)
m, info = ray_amd.ScatterRelation(med, None, 2048, 2048, 1e-2, ctx=ctx, info=True, src_begin=0, src_end=256)
ctx._lib.ray_debug_stats(out)
steps = int(info["nsteps"].sum())
print({"ray_steps": steps, "wave_lookups": out[0], "lookups_with_miss": out[1], "missing_lanes": out[2],
       "wide_fan_fetches": out[3], "miss_wave_frac": out[1] / max(out[0], 1),
       "miss_lane_frac": out[2] / max(64 * out[0], 1), "fetches": out[4], "mean_runs": out[10] / max(out[4], 1), "same_cell_lookups": out[11],
       "frac_runs_gt_21_32_42_48_53": [round(out[k] / max(out[4], 1), 4) for k in (5, 6, 7, 8, 9)]})
