#!/usr/bin/env python3
"""Occupancy timeline of the headline k_shoot launch (instrumentation build: hipcc -DRAY_TIMELINE -> exp/lib_tl.so,
RAY_AMD_LIB=exp/lib_tl.so): start / end clock of every workgroup -> gpurun_out/r2/timeline.npy"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ray_amd
from ray_amd import device as rdev

NS = ND = 2048
N = 512
torch.cuda.set_device(0)
ctx = ray_amd.Context(0)
sp = ray_amd.SinProd.smooth_demo()
p = np.zeros(N)
ray_amd._lib.check(ctx._lib.ray_linspace(-1.5, 1.5, N, p.ctypes.data))
c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])
med = ray_amd.GridMedium(c, 1.5, ctx)
num = NS * ND
m = torch.empty(9 * num, dtype=torch.float64, device="cuda")
for _ in range(3):
    rdev.shoot_to_exit_dev(ctx, med, NS, ND, 1e-2, m.data_ptr(), num)
ctx.synchronize()
nblk = int(os.environ.get("RAY_TL_BLOCKS", "16384"))
out = np.zeros(3 * nblk, dtype=np.uint64)
lib = C.CDLL(os.environ["RAY_AMD_LIB"])
assert lib.ray_debug_timeline(out.ctypes.data_as(C.c_void_p), nblk) == 0
np.save(os.path.join("gpurun_out", "r2", "timeline.npy"), out.reshape(nblk, 3))
t = out.reshape(nblk, 3)
print("span (clock ticks)", int(t[:, 1].max() - t[:, 0].min()), "blocks", nblk)
