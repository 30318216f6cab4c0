#!/usr/bin/env python3
"""BASELINE config 4, one rank's share on one GPU: theta-block [k*1024, (k+1)*1024) of the 8192 x 8192 sweep on the
512^2 grid, device-resident (ray_shoot_to_exit_dev) + misfit.  Prints one JSON line per block."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ray_amd
from ray_amd import device as rdev

NS = ND = 8192
N, EXT, DT = 512, 1.5, 1e-2
torch.cuda.set_device(0)
ctx = ray_amd.Context(0)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
ctx.set_stream(s.cuda_stream)
sp = ray_amd.SinProd.smooth_demo()
p = np.zeros(N)
ray_amd._lib.check(ctx._lib.ray_linspace(-EXT, EXT, N, p.ctypes.data))
c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])
med = ray_amd.GridMedium(c, EXT, ctx)
num = (NS // 8) * ND
m = torch.empty(9 * num, dtype=torch.float64, device="cuda")
obs = torch.zeros(9 * num, dtype=torch.float64, device="cuda")
st = torch.empty(num, dtype=torch.int32, device="cuda")
sums = torch.zeros(2, dtype=torch.float64, device="cuda")
for blk in (0, 3, 7):
    sb, se = blk * (NS // 8), (blk + 1) * (NS // 8)
    for rep in range(2):                        # first pass warms the tables / boundary grid
        ctx.reset_counters()
        ctx.enable_timing(True)
        for _ in range(5):
            rdev.shoot_to_exit_dev(ctx, med, NS, ND, DT, m.data_ptr(), num, src_begin=sb, src_end=se, status_ptr=st.data_ptr())
            rdev.misfit_dev(ctx, m.data_ptr(), obs.data_ptr(), num, num, sums.data_ptr())
        cnt = ctx.counters()
    ok = int((st == ray_amd.ST_EXITED).sum().item())
    ms = cnt["kernel_ms"] / cnt["launches"]
    print(json.dumps({"config": "C4 theta-block", "sources": [sb, se], "rays": num, "exited": ok,
                      "ray_steps_per_launch": cnt["ray_steps"] / cnt["launches"], "kernel_ms": ms,
                      "ray_steps_per_s": cnt["ray_steps"] / cnt["launches"] / ms * 1e3,
                      "frac_fp64_vector_peak": 225 * cnt["ray_steps"] / cnt["launches"] / ms * 1e3 / 78.6e12}), flush=True)
