#!/usr/bin/env python3
"""bench.py -- headline benchmark of the forward ray-shooting core (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input:
  shoot-to-exit of a 2048 x 2048 (theta, phi) boundary phase-space grid per GPU on a 512^2 fp64
  wavespeed grid (BASELINE.json configs[1]; init generated on device, outputs stay in HBM)
  + the misfit reduction against an observation resident in HBM
  + (N > 1) ONE all-reduce (RCCL) of {sum mismatch^2, sum target^2}.
N > 1: rays shard by source angle, rank r shoots the comb of sources r, r + N, ... of a (2048*N) x 2048 grid
(2048 sources per GPU -- weak scaling: per-GPU work fixed, and equal whatever the medium's asymmetry;
RAY_BENCH_SHARD=block selects contiguous theta-blocks instead); no ray data is exchanged.

Prints ONE JSON line (rank 0).  `value` = ray-steps/s of the whole job; `roofline` prices the
shoot kernel at SURVEY.md section 8(d)'s 225 algorithmic flop per ray-step against the fp64 VECTOR peak
(78.6 TFLOP/s; there is no dense contraction, so neither the HBM nor the MFMA roofline binds);
`cpu_baseline` times the CPU oracle (a C port of the reference's Julia path: Julia is not
installed and the reference is Julia-0.6-only) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_STEP_GRID = 225            # algorithmic fp64 flop per ray-step, SURVEY.md section 8(d)
FP64_VECTOR_PEAK = 78.6      # TFLOP/s: 256 CU x 4 SIMD x 16 lanes x 2 (FMA) x 2.4 GHz
NS_PER_GPU, ND, NGRID, EXT, DT = 2048, 2048, 512, 1.5, 1e-2


def cpu_baseline(seconds_target=12.0):
    """The oracle (oracle/ray_oracle.c, a C port of the reference's CPU path) on a bounded
    sample of the same workload: the first S sources x all 2048 directions."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    c = orc.sinprod_grid(orc.SMOOTH_DEMO, NGRID, EXT)
    tab = orc.GridTables(c, EXT)
    t0 = time.perf_counter()
    _, _, nst, _ = orc.scatter_relation(tab, NS_PER_GPU, ND, DT, src_begin=0, src_end=2, nthreads=threads)
    rate = nst.sum() / (time.perf_counter() - t0)
    S = int(max(2, min(NS_PER_GPU, rate * seconds_target / (127.0 * ND))))
    t0 = time.perf_counter()
    _, _, nst, _ = orc.scatter_relation(tab, NS_PER_GPU, ND, DT, src_begin=0, src_end=S, nthreads=threads)
    el = time.perf_counter() - t0
    # the same at one thread (SURVEY.md section 8d asks for both), on a sample ~3 s long
    S1 = max(1, min(S, int(S * 3.0 / seconds_target / threads) + 1))
    t1 = time.perf_counter()
    _, _, nst1, _ = orc.scatter_relation(tab, NS_PER_GPU, ND, DT, src_begin=0, src_end=S1, nthreads=1)
    el1 = time.perf_counter() - t1
    return {"value": float(nst.sum() / el), "unit": "ray-steps/s", "cores": threads, "kind": "port",
            "single_thread": {"value": float(nst1.sum() / el1), "cores": 1,
                              "sample": f"sources [0,{S1}): {int(nst1.sum())} ray-steps in {el1:.2f} s"},
            "sample": f"sources [0,{S}) x {ND} directions of the same 2048x2048 / 512^2 / dt=1e-2 "
                      f"workload: {int(nst.sum())} ray-steps in {el:.2f} s (C port of the "
                      "reference's Julia CPU path, -O2 -ffp-contract=off, OpenMP over rays)"}


def recorded_traffic():
    """HBM bytes per launch of the shoot kernel from the newest committed PMC profile
    (profiles/rN/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected
    as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process, so
    this is the RECORDED figure of the same command, not a live one -- the bench line says so in
    `traffic_source`.  Returns (bytes, l2 hit rate, profile's kernel average in ms, source) or Nones."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return None, None, None, None
    d = json.load(open(files[-1]))
    src = os.path.relpath(files[-1], ROOT) + " (recorded by rocprofv3 --pmc on an earlier run of this command, not live)"
    return d.get("traffic_bytes_per_launch"), d.get("l2_hit_rate"), d.get("kernel_avg_ms"), src


def secondary_timings(ctx, torch, np, ray_amd, rdev, dev, m, status, nsteps, num, p, reps=4, hmed=None):
    """The other kernels of the hot path on the driver's clock: timed in this process after the
    headline loop, HIP events around every launch (the library's counters), few launches each, clocks already warm.
    `value` / `roofline` of the line are untouched by this.  Every entry: kernel_ms (average), ray_steps per launch,
    ray_steps_per_s; `frac` where SURVEY.md section 8(d) defines a flop count (gridded RK4: 225 per ray-step)."""
    out = {}
    nrf = torch.empty(num, dtype=torch.int32, device=dev)
    X = torch.empty(4 * num, dtype=torch.float64, device=dev)

    def timed(fn, n=reps):
        fn()
        ctx.synchronize()
        ctx.reset_counters()
        ctx.enable_timing(True)
        for _ in range(n):
            fn()
        c = ctx.counters()
        ctx.enable_timing(False)
        ms = c["kernel_ms"] / max(c["launches"], 1)
        steps = c["ray_steps"] / max(c["launches"], 1)
        return {"kernel_ms": ms, "launches": int(c["launches"]), "ray_steps": int(steps), "ray_steps_per_s": steps / (ms * 1e-3),
                "rays_per_s": num / (ms * 1e-3)}

    def shoot(med, ob, tu):
        return lambda: rdev.shoot_to_exit_dev(ctx, med, NS_PER_GPU, ND, DT, m.data_ptr(), num, ob=ob, time_unit=tu,
                                              status_ptr=status.data_ptr(), nsteps_ptr=nsteps.data_ptr(), nreflect_ptr=nrf.data_ptr())

    # BASELINE configs[2] at its timing size: ObstacleDemo's analytic medium around the circle r0 = 0.3, t in steps
    e = timed(shoot(ray_amd.SinProd.obstacle_demo(), ray_amd.Circle(0.3), ray_amd.TIME_STEPS))
    e.update(workload="config 3: analytic SinProd(0.2,2pi,pi) + circle 0.3, 2048x2048 rays, dt=1e-2 (k_shoot<sinprod,circle>)",
             reflections=int(nrf.sum().item()), not_exited=int((status != ray_amd.ST_EXITED).sum().item()))
    out["config3_analytic_circle"] = e
    # the same workload around the obstacle WITH CAVITIES (the trefoil of ObstacleDemo.jl:23-24,30-31: up to four reflections per ray)
    e = timed(shoot(ray_amd.SinProd.obstacle_demo(), ray_amd.Trefoil.demo(), ray_amd.TIME_STEPS))
    e.update(workload="config 3 with the cavity obstacle: analytic SinProd(0.2,2pi,pi) + trefoil(0.4,0.3,0.2), 2048x2048 rays, dt=1e-2 "
                      "(k_shoot<sinprod,trefoil>)",
             reflections=int(nrf.sum().item()), not_exited=int((status != ray_amd.ST_EXITED).sum().item()))
    out["config3_analytic_trefoil"] = e
    # config 1's medium at the same size (what ScatterRelation runs in NonObstacleDemo.jl)
    e = timed(shoot(ray_amd.SinProd.smooth_demo(), None, ray_amd.TIME_SECONDS))
    e.update(workload="analytic SinProd(0.3,1.5pi,1.5pi), 2048x2048 rays, dt=1e-2 (k_shoot<sinprod,none>)",
             not_exited=int((status != ray_amd.ST_EXITED).sum().item()))
    out["analytic_smooth"] = e
    # SURVEY.md section 8(d)'s secondary point: the headline sweep at dt = h/2 (453 steps per ray; same medium handle `hmed`)
    if hmed is not None:
        hh = 0.5 * (2 * EXT / (NGRID - 1))
        e = timed(lambda: rdev.shoot_to_exit_dev(ctx, hmed, NS_PER_GPU, ND, hh, m.data_ptr(), num,
                                                 status_ptr=status.data_ptr(), nsteps_ptr=nsteps.data_ptr()), n=2)
        e.update(workload="headline sweep at dt = h/2 (2048x2048 rays, N=512; k_shoot<grid,none>)",
                 frac=F_STEP_GRID * e["ray_steps_per_s"] / 1e12 / FP64_VECTOR_PEAK)
        out["grid_dt_half_h"] = e
        timed(lambda: rdev.shoot_to_exit_dev(ctx, hmed, NS_PER_GPU, ND, DT, m.data_ptr(), num,                # m back to the dt = 1e-2 rows
                                             status_ptr=status.data_ptr(), nsteps_ptr=nsteps.data_ptr()), n=1)
    # a9: the trajectory part of ScatterForwardOperator for the rays just shot, on a perturbed gridded model, fan hint
    c0 = np.asfortranarray([[0.85 * ray_amd.SinProd.smooth_demo()(float(x), float(y)) + 0.1 for y in p] for x in p])
    med0 = ray_amd.GridMedium(c0, EXT, ctx)
    ctx.set_option(ray_amd._lib.OPT_FAN_DIRECTIONS, ND)
    try:
        e = timed(lambda: rdev.trace_fixed_time_dev(ctx, med0, m.data_ptr(), num, num, DT, ray_amd.TRACE_SHORTEN_CEASE, X.data_ptr(), num,
                                                    status_ptr=status.data_ptr(), nsteps_ptr=nsteps.data_ptr()))
    finally:
        ctx.set_option(ray_amd._lib.OPT_FAN_DIRECTIONS, 0)
    e.update(workload="a9 ChunkProcessing! trajectories, 2048x2048 rays, N=512, dt=1e-2, fan hint (k_trace)",
             frac=F_STEP_GRID * e["ray_steps_per_s"] / 1e12 / FP64_VECTOR_PEAK)
    out["trace_a9_fan"] = e
    # NEXT-1: the complete ScatterForwardOperator build (X, rho, Frechet block rows) at 512 x 512 rays, N = 513
    ns2 = nd2 = 512
    mh = ray_amd.ScatterRelation(ray_amd.SinProd.smooth_demo(), None, ns2, nd2, DT, ctx=ctx)
    p2 = np.zeros(513)
    ray_amd._lib.check(ctx._lib.ray_linspace(-EXT, EXT, 513, p2.ctypes.data))
    med2 = ray_amd.GridMedium(np.asfortranarray([[ray_amd.SinProd.smooth_demo()(float(x), float(y)) for y in p2] for x in p2]), EXT, ctx)
    ray_amd.forward_operator_dev(med2, mh, EXT, DT, ctx=ctx, want_rho=False)[0].close()         # sizes the workspaces
    ctx.reset_counters()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    F, _, _, info = ray_amd.forward_operator_dev(med2, mh, EXT, DT, ctx=ctx, want_rho=False)
    call_s = time.perf_counter() - t0
    c = ctx.counters()
    ctx.enable_timing(False)
    steps = int(info["nsteps"].astype(np.int64).sum())
    out["frechet_build"] = {"workload": "ScatterForwardOperator build (k_frechet + merge -> device block rows), 512x512 rays, N=513, dt=1e-2",
                            "kernel_ms": c["kernel_ms"], "ray_steps": steps, "ray_steps_per_s": steps / (c["kernel_ms"] * 1e-3),
                            "nnz": int(F.nnz), "call_ms_host_pointers": call_s * 1e3}
    # NEXT-2: the inner loop of the matrix-free solve on that matrix, w = M'(M x)
    x = torch.randn(513 * 513, dtype=torch.float64, device=dev) * 1e-2
    w = torch.empty_like(x)
    n_entries = F.block_rows_dev()[0]
    rdev.frechet_normal_matvec_dev(F, x.data_ptr(), 0, w.data_ptr())
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rdev.frechet_normal_matvec_dev(F, x.data_ptr(), 0, w.data_ptr())
    e1.record()
    e1.synchronize()
    nm_ms = e0.elapsed_time(e1) / reps
    out["normal_matvec"] = {"workload": "w = M'(M x) on the block rows of that matrix: y = M x (k_br_matvec, ray-major rows) then "
                                        "w = M'y (k_br_band_scatter, band-major copy); 36 B per entry and pass (4 f64 + one index word)",
                            "kernel_ms": nm_ms,
                            "block_row_entries": int(n_entries), "algorithmic_bytes": int(n_entries) * 36 * 2,
                            "hbm_frac": int(n_entries) * 36 * 2 / (nm_ms * 1e-3) / 8.0e12}
    F.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    # the clocks of an idle GPU take about ten 4-ms launches to ramp (profiles/r2/README.md: 4.6 -> 3.66 ms per launch)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary kernels (config 3, analytic smooth, a9 trace, Frechet build) timed after the headline loop")
    ap.add_argument("--arith", choices=["fused", "exact"], default="fused",
                    help="arithmetic flavour of the gridded kernels (include/ray_amd.h); fused is the library default")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # rehearsal knobs (1-GPU box only): RAY_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # RAY_BENCH_BACKEND=gloo replaces RCCL, so the N>1 control flow can be exercised without N GPUs
    if os.environ.get("RAY_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("RAY_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # RAY_BENCH_FORCE_PG=1: build the process group (and run the collective) even with one rank -- exercises the
    # RCCL code path of the N > 1 run on a one-GPU box
    force_pg = os.environ.get("RAY_BENCH_FORCE_PG") == "1" and world == 1
    if force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, device_id=dev if backend == "nccl" else None, rank=0, world_size=1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import ray_amd
    from ray_amd import device as rdev

    ctx = ray_amd.Context(local)
    # ONE explicit stream for everything: the library's kernels (borrowed handle), torch's own ops and the
    # collective (RCCL orders itself against torch's current stream), so that the all-reduce of the misfit sums
    # is stream-ordered after k_misfit_final and the next shoot after the all-reduce.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_arith(ray_amd.ARITH_FUSED if args.arith == "fused" else ray_amd.ARITH_EXACT)

    # synthetic medium: c = 1 + 0.3 sin(1.5 pi x) sin(1.5 pi y) on linspace(-1.5,1.5,512)^2
    sp = ray_amd.SinProd.smooth_demo()
    p = np.zeros(NGRID)
    ray_amd._lib.check(ctx._lib.ray_linspace(-EXT, EXT, NGRID, p.ctypes.data))
    c = np.asfortranarray([[sp(float(x), float(y)) for y in p] for x in p])
    if os.environ.get("RAY_BENCH_CONST") == "1":                 # ablation runs: data-independent trajectories
        c = np.asfortranarray(np.ones_like(c))
    med = ray_amd.GridMedium(c, EXT, ctx)

    # sharding of the (2048 world) x 2048 boundary grid: a COMB of the source circle per rank (sources rank,
    # rank + world, ...: every rank sees the whole medium, equal work) or a contiguous theta-block (an arc per rank:
    # the arcs' ray-step counts differ by 16 % on this medium and the step time is the slowest rank's)
    ns_total = NS_PER_GPU * world
    shard = os.environ.get("RAY_BENCH_SHARD", "comb")
    if shard == "comb":
        from ray_amd.dist import theta_comb
        sb, se, stride, cnt_src = theta_comb(ns_total, rank, world)
        assert cnt_src == NS_PER_GPU
        ctx.set_option(ray_amd._lib.OPT_SOURCE_STRIDE, stride)
    else:
        sb, se = rank * NS_PER_GPU, (rank + 1) * NS_PER_GPU
    num = NS_PER_GPU * ND
    m = torch.empty(9 * num, dtype=torch.float64, device=dev)
    obs = torch.zeros(9 * num, dtype=torch.float64, device=dev)
    status = torch.empty(num, dtype=torch.int32, device=dev)
    nsteps = torch.empty(num, dtype=torch.int32, device=dev)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)

    trace = os.environ.get("RAY_BENCH_TRACE") == "1"            # progress lines on stderr (rehearsals)
    if trace:                                                   # and the Python stacks if nothing has finished by then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("RAY_BENCH_TRACE_DUMP_S", "45")), exit=False, file=sys.stderr)

    def say(msg):
        if trace:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    # RCCL (the real run): the collective on the DEVICE tensor, ordered on `stream`.  The gloo rehearsal (two ranks
    # time-slicing one GPU) reduces a host copy: ProcessGroupGloo's own device-tensor all_reduce blocks both ranks
    # for good on its third call there -- with the stream ordering fixed and with or without HIP timing events
    # (gpurun_out/r2/bench_2r*.err) -- and is not a path the product uses.
    # RAY_BENCH_HOST_REDUCE=0 / 1 overrides.
    host_reduce = os.environ.get("RAY_BENCH_HOST_REDUCE", "1" if backend == "gloo" else "0") == "1"
    if backend == "gloo" and not host_reduce:
        raise SystemExit("RAY_BENCH_BACKEND=gloo with RAY_BENCH_HOST_REDUCE=0 is refused: ProcessGroupGloo's all_reduce of a "
                         "DEVICE tensor blocks both ranks for good when they share one GPU "
                         "(profiles/r2/gloo_device_allreduce_hang_trace.txt; DESIGN.md section 6).  The rehearsal reduces a host "
                         "copy; the real run uses RCCL on the device tensor.")

    def allreduce(t, op=None):
        """One all-reduce of a small device tensor (see host_reduce above)."""
        op = op or dist.ReduceOp.SUM
        if not host_reduce:
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    def step():
        rdev.shoot_to_exit_dev(ctx, med, ns_total, ND, DT, m.data_ptr(), num, src_begin=sb, src_end=se,
                               status_ptr=status.data_ptr(), nsteps_ptr=nsteps.data_ptr())
        rdev.misfit_dev(ctx, m.data_ptr(), obs.data_ptr(), num, num, sums.data_ptr())
        if world > 1 or force_pg:
            allreduce(sums)                # the one collective per iteration (misfit sums)

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # the observation = the data itself from a first pass, perturbed (misfit cost is data-independent)
    say("medium built")
    step()
    torch.cuda.synchronize()
    say("first step done")
    obs.copy_(m)
    obs[4 * num:8 * num] *= 1.001
    for _ in range(args.warmup):
        step()
    fence()
    say("warm-up done")
    ctx.reset_counters()
    notiming = os.environ.get("RAY_BENCH_NOTIMING") == "1"      # rehearsal knob: no HIP events around the launches
    ctx.enable_timing(not notiming)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
        say(f"timed step {k} enqueued")
    fence()
    elapsed = time.perf_counter() - t0
    say("timed region done")
    cnt = ctx.counters()
    ctx.enable_timing(False)

    # the collective really carried every rank's sums: `sums` (reduced in the last timed step) must equal the
    # rank-ordered sum of the ranks' own misfit sums, recomputed here and gathered
    if world > 1 or force_pg:
        own = torch.zeros(2, dtype=torch.float64, device=dev)
        rdev.misfit_dev(ctx, m.data_ptr(), obs.data_ptr(), num, num, own.data_ptr())
        parts = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        stream.synchronize()
        if backend == "nccl":
            dparts = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(dparts, own)
            parts = [t.cpu() for t in dparts]
        else:
            dist.all_gather(parts, own.cpu())
        want = sum(parts[1:], parts[0].clone())
        got = sums.cpu()
        if not torch.allclose(got, want, rtol=1e-12, atol=0):
            raise SystemExit(f"rank {rank}: all-reduced misfit sums {got.tolist()} != sum over ranks {want.tolist()}")
        say("reduced sums verified against the ranks' own")

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(cnt["ray_steps"]), float(cnt["rays"])], dtype=torch.float64, device=dev)
    most = torch.tensor([float(cnt["ray_steps"])], dtype=torch.float64, device=dev)      # the busiest rank's ray-steps
    if world > 1 or force_pg:
        allreduce(el, dist.ReduceOp.MAX)
        allreduce(tot)
        allreduce(most, dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ray_steps, rays = float(tot[0].item()), float(tot[1].item())

    bad = int((status != ray_amd.ST_EXITED).sum().item())
    if bad and os.environ.get("RAY_BENCH_NOCHECK") != "1":      # (ablation builds compute garbage on purpose)
        raise SystemExit(f"{bad} rays did not exit cleanly")

    if rank == 0:
        kern_s = cnt["kernel_ms"] * 1e-3 / max(cnt["launches"], 1)
        steps_per_launch = cnt["ray_steps"] / max(cnt["launches"], 1)
        achieved = F_STEP_GRID * steps_per_launch / kern_s / 1e12
        traffic, l2_hit, prof_ms, traffic_src = recorded_traffic()
        out = {
            "metric": "fp64 RK4 ray-steps/sec", "value": ray_steps / elapsed, "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rays_per_s": rays / elapsed,
            "config": {"workload": "forward shooting to exit, 2048x2048 (theta,phi) rays per GPU, "
                                   "512^2 fp64 wavespeed grid, dt=1e-2 (BASELINE configs[1]) + misfit "
                                   "reduction" + (" + 1 all-reduce" if world > 1 else ""),
                       "arith": args.arith, "ns_total": ns_total, "nd": ND, "grid": NGRID, "dt": DT,
                       "rays_per_gpu": num, "mean_steps_per_ray": steps_per_launch / num,
                       "parallelism": f"theta-{shard} x{world}",
                       "busiest_rank_over_mean_ray_steps": float(most.item()) * world / ray_steps},
            "roofline": {"bound": "fp64-vector", "kernel": "k_shoot<grid,no-obstacle>",
                         "achieved": achieved, "peak": FP64_VECTOR_PEAK, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK, "traffic": traffic,
                         "traffic_source": traffic_src, "profile_kernel_ms": prof_ms,
                         # the same launch against the other roof: recorded HBM bytes / kernel time / 8 TB/s
                         "hbm_frac": (traffic / kern_s / 8.0e12) if traffic else None,
                         "l2_hit_rate": l2_hit,
                         "kernel_ms": kern_s * 1e3, "kernel_ms_source": "HIP events around every launch of the timed region, on the launch stream (live)",
                         "ray_steps_per_launch": steps_per_launch,
                         "flop_per_ray_step": F_STEP_GRID,
                         "kernel_ray_steps_per_s": steps_per_launch / kern_s},
        }
        out["runtime"] = ray_amd.runtime_info()        # which HIP runtime / RCCL this process bound
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_timings(ctx, torch, np, ray_amd, rdev, dev, m, status, nsteps, num, p, hmed=med)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1 or force_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
