#!/usr/bin/env python3
"""Kernel rate of the gridded shoot for other (ns, nd) shapes of the same total size: narrower or
wider direction fans per wave (64 consecutive directions).  usage: bench_shape.py ns nd [ns nd ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_amd  # noqa: E402
from ray_amd import device as rdev  # noqa: E402

ctx = ray_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
sp = ray_amd.SinProd.smooth_demo()
N = int(os.environ.get("RAY_SHAPE_N", "512"))
p = np.zeros(N)
ray_amd._lib.check(ctx._lib.ray_linspace(-1.5, 1.5, N, p.ctypes.data))
c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])
med = ray_amd.GridMedium(c, 1.5, ctx)
args = [int(a) for a in sys.argv[1:]] or [2048, 2048]
for ns, nd in zip(args[0::2], args[1::2]):
    num = ns * nd
    m = torch.empty(9 * num, dtype=torch.float64, device="cuda")
    for rep in range(2):
        ctx.reset_counters(); ctx.enable_timing(True)
        for _ in range(5):
            rdev.shoot_to_exit_dev(ctx, med, ns, nd, 1e-2, m.data_ptr(), num)
        torch.cuda.synchronize()
        cnt = ctx.counters(); ctx.enable_timing(False)
    print(json.dumps({"grid": N, "ns": ns, "nd": nd, "kernel_ms": cnt["kernel_ms"] / cnt["launches"],
                      "ray_steps_per_s": cnt["ray_steps"] / (cnt["kernel_ms"] * 1e-3)}))
