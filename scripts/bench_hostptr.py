#!/usr/bin/env python3
"""BASELINE config 2 through the HOST-pointer entry point (what a blocking Julia ccall uses): ray_shoot_to_exit
into a caller-owned column-major m -- pageable, page-locked (ray_host_register), and with columns 5..9 only --
beside the HBM-resident rate of ray_shoot_to_exit_dev.  One JSON line per variant."""
import ctypes as C
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ray_amd
from ray_amd import _lib

NS = ND = 2048
N, EXT, DT = 512, 1.5, 1e-2
ctx = ray_amd.Context(0)
lib = _lib.load()
sp = ray_amd.SinProd.smooth_demo()
p = np.zeros(N)
_lib.check(lib.ray_linspace(-EXT, EXT, N, p.ctypes.data))
c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])
med = ray_amd.GridMedium(c, EXT, ctx)
num = NS * ND
prm = _lib.ShootParams(NS, ND, 0, NS, 0.0, math.pi, DT, 0, 0)
m = np.zeros((num, 9), order="F")


def run(tag, reps=4):
    ts = []
    for _ in range(reps + 1):
        ctx.reset_counters()
        t0 = time.perf_counter()
        _lib.check(lib.ray_shoot_to_exit(ctx.handle, med.handle, None, C.byref(prm), m.ctypes.data, num, None, None, None))
        ts.append(time.perf_counter() - t0)
    cnt = ctx.counters()
    best = min(ts[1:])
    print(json.dumps({"variant": tag, "call_ms_first": ts[0] * 1e3, "call_ms": best * 1e3, "ray_steps": cnt["ray_steps"],
                      "ray_steps_per_s_through_host_pointers": cnt["ray_steps"] / best,
                      "bytes_over_pcie": int(m.nbytes * (5 / 9 if "5..9" in tag else 1))}), flush=True)


run("pageable m, 9 columns")
ctx.set_option(_lib.OPT_SKIP_INIT_COLUMNS, 1)
run("pageable m, columns 5..9")
ctx.set_option(_lib.OPT_SKIP_INIT_COLUMNS, 0)
t0 = time.perf_counter()
try:
    ray_amd.host_register(m)
except ray_amd.RayError as e:
    print(json.dumps({"variant": "page-locked m", "unmeasured": "hipHostRegister refused on this box: " + str(e)}))
    sys.exit(0)
treg = time.perf_counter() - t0
run("page-locked m (ray_host_register, %.0f ms once), 9 columns" % (treg * 1e3))
ctx.set_option(_lib.OPT_SKIP_INIT_COLUMNS, 1)
run("page-locked m, columns 5..9")
ray_amd.host_unregister(m)
