#!/usr/bin/env python3
"""Secondary timings (not the headline bench): every kernel of the hot path on a 2048 x 2048 sweep,
device-resident, HIP-event kernel time from the library's counters.  One JSON line per workload.
usage: python3 scripts/bench_workloads.py [--reps 5] [--ns 2048 --nd 2048]"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ns", type=int, default=2048)
    ap.add_argument("--nd", type=int, default=2048)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import ray_amd
    from ray_amd import device as rdev

    dev = torch.device("cuda", 0)
    ctx = ray_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ns, nd, N, ext = args.ns, args.nd, 512, 1.5
    num = ns * nd
    m = torch.empty(9 * num, dtype=torch.float64, device=dev)
    X = torch.empty(4 * num, dtype=torch.float64, device=dev)
    st = torch.empty(num, dtype=torch.int32, device=dev)
    nst = torch.empty(num, dtype=torch.int32, device=dev)
    nrf = torch.empty(num, dtype=torch.int32, device=dev)
    p = np.zeros(N)
    ctx._lib.ray_linspace(-ext, ext, N, p.ctypes.data)

    def grid_of(sp):
        return ray_amd.GridMedium(np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p]), ext, ctx)

    smooth, obst = ray_amd.SinProd.smooth_demo(), ray_amd.SinProd.obstacle_demo()
    gsmooth, gobst = grid_of(smooth), grid_of(obst)
    circle = ray_amd.Circle(0.3)
    trefoil = ray_amd.Trefoil.demo()

    def shoot(med, ob, tu, dt=1e-2):
        return lambda: rdev.shoot_to_exit_dev(ctx, med, ns, nd, dt, m.data_ptr(), num, ob=ob, time_unit=tu,
                                              status_ptr=st.data_ptr(), nsteps_ptr=nst.data_ptr(),
                                              nreflect_ptr=nrf.data_ptr())

    def run(name, fn, arith="fused"):
        if args.only and args.only not in name:
            return
        ctx.set_arith(ray_amd.ARITH_FUSED if arith == "fused" else ray_amd.ARITH_EXACT)
        fn()
        ctx.synchronize()
        ctx.reset_counters()
        ctx.enable_timing(True)
        for _ in range(args.reps):
            fn()
        c = ctx.counters()
        ctx.enable_timing(False)
        ms = c["kernel_ms"] / max(c["launches"], 1)
        steps = c["ray_steps"] / max(c["launches"], 1)
        # live lanes: a wave (2 sources x 32 directions) steps until its last lane has left
        live = None
        if "shoot" in name and ns % 2 == 0 and nd % 32 == 0:
            w = nst.view(ns // 2, 2, nd // 32, 32).to(torch.int64)
            live = float(w.sum().item()) / float(64 * w.amax(dim=(1, 3)).sum().item())
        print(json.dumps({"workload": name, "arith": arith, "rays": num, "kernel_ms": round(ms, 3),
                          "ray_steps": int(steps), "ray_steps_per_s": steps / (ms * 1e-3),
                          "rays_per_s": num / (ms * 1e-3), "steps_per_ray": steps / num, "live_lane_fraction": live,
                          "reflections": int(nrf.sum().item()) if "obstacle" in name else 0,
                          "not_exited": int((st & 1).eq(0).sum().item()) if "trace" not in name else 0}), flush=True)

    run("shoot grid512 smooth", shoot(gsmooth, None, 0), "fused")
    run("shoot grid512 smooth", shoot(gsmooth, None, 0), "exact")
    # SURVEY.md section 8d's secondary point: dt = h/2 (half a cell per step)
    run("shoot grid512 smooth dt=h/2", shoot(gsmooth, None, 0, (p[1] - p[0]) / 2), "fused")
    run("shoot analytic smooth (NonObstacleDemo medium)", shoot(smooth, None, 0))
    run("shoot analytic obstacle circle0.3 (ObstacleDemo workload)", shoot(obst, circle, 1))
    run("shoot analytic radial sine sum, 4 terms (NonObstacleDemo.jl:7-11 alternate)", shoot(ray_amd.RadialSineSum.smooth_alt(), None, 0))
    run("shoot grid512 obstacle circle0.3", shoot(gobst, circle, 1))
    run("shoot analytic obstacle trefoil 0.4/0.3/0.2 (ObstacleDemo.jl:23-24 alternate, cavities)", shoot(obst, trefoil, 1))
    run("shoot grid512 obstacle trefoil", shoot(gobst, trefoil, 1))
    # a9/a8: trace the smooth data on the grid for its own traveltimes at the reconstruction step 5e-2
    ctx.set_arith(ray_amd.ARITH_FUSED)
    shoot(gsmooth, None, 0)()
    ctx.synchronize()
    for tdt, fan in ((5e-2, 0), (1e-2, 0), (5e-2, nd), (1e-2, nd)):   # the reconstruction's step and the shooting step; rows as given / fan hint
      ctx.set_option(ray_amd._lib.OPT_FAN_DIRECTIONS, fan)
      for _ in (0,):
        for mode, nm in ((0, f"trace a9 shorten+cease dt={tdt:g} fan={fan}"), (1, f"trace a8 overshoot dt={tdt:g} fan={fan}")):
            run(nm, lambda: rdev.trace_fixed_time_dev(ctx, gsmooth, m.data_ptr(), num, num, tdt, mode, X.data_ptr(), num,
                                                      status_ptr=st.data_ptr(), nsteps_ptr=nst.data_ptr()))


if __name__ == "__main__":
    main()
