#!/usr/bin/env python3
"""The weak-scaling workload of bench.py --gpus 8, one rank's share at a time on ONE GPU: (2048 x 8) x 2048 boundary
grid on the 512^2 medium, rank r's sources as a contiguous theta-block or as a comb (RAY_OPT_SOURCE_STRIDE).
Prints one JSON line per (sharding, rank) and a summary: the step time of an 8-rank job is the slowest rank's."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ray_amd
from ray_amd import _lib, device as rdev
from ray_amd.dist import theta_block, theta_comb

WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NS1, ND, N, EXT, DT = 2048, 2048, 512, 1.5, 1e-2
torch.cuda.set_device(0)
ctx = ray_amd.Context(0)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
ctx.set_stream(s.cuda_stream)
sp = ray_amd.SinProd.smooth_demo()
p = np.zeros(N)
_lib.check(ctx._lib.ray_linspace(-EXT, EXT, N, p.ctypes.data))
c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])
med = ray_amd.GridMedium(c, EXT, ctx)
ns_total = NS1 * WORLD
num = NS1 * ND
m = torch.empty(9 * num, dtype=torch.float64, device="cuda")
st = torch.empty(num, dtype=torch.int32, device="cuda")
summary = {}
for shard in ("block", "comb"):
    res = []
    for r in range(WORLD):
        if shard == "comb":
            sb, se, stride, _ = theta_comb(ns_total, r, WORLD)
        else:
            (sb, se), stride = theta_block(ns_total, r, WORLD), 1
        ctx.set_option(_lib.OPT_SOURCE_STRIDE, stride)
        for rep in range(2):                    # first pass warms the boundary grid of this share
            ctx.reset_counters()
            ctx.enable_timing(True)
            for _ in range(5):
                rdev.shoot_to_exit_dev(ctx, med, ns_total, ND, DT, m.data_ptr(), num, src_begin=sb, src_end=se, status_ptr=st.data_ptr())
            cnt = ctx.counters()
        ms = cnt["kernel_ms"] / cnt["launches"]
        steps = cnt["ray_steps"] / cnt["launches"]
        res.append((ms, steps))
        print(json.dumps({"sharding": shard, "world": WORLD, "rank": r, "exited": int((st == ray_amd.ST_EXITED).sum().item()),
                          "ray_steps": steps, "kernel_ms": ms, "ray_steps_per_s": steps / ms * 1e3}), flush=True)
    ms = np.array([a for a, _ in res]); steps = np.array([b for _, b in res])
    summary[shard] = {"ray_steps_max_over_mean": float(steps.max() / steps.mean()), "kernel_ms_max": float(ms.max()),
                      "kernel_ms_mean": float(ms.mean()), "job_ray_steps_per_s_at_slowest_rank": float(steps.sum() / ms.max() * 1e3),
                      "efficiency_vs_mean_rank": float(ms.mean() / ms.max())}
ctx.set_option(_lib.OPT_SOURCE_STRIDE, 1)
print(json.dumps({"summary": summary}), flush=True)
