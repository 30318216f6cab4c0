"""SURVEY.md section 8f NEXT-4: ray polylines (the data ScatterRelationObstaclePlot collects, Obstacle.jl:98-221)
from the PATH kernels against the pure-Python restatement (oracle/ray_oracle.py shoot(paths=...))."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ray_oracle
    return ray_oracle


def test_paths_obstacle_analytic(ctx):
    import ray_amd
    P = _py()
    so = P.SinProd.obstacle_demo()
    ns, nd = 6, 40
    rows = P.init_states(so, ns, nd, 0.0, np.pi)
    pick = [3, 17, 19, 20, 21, 77, 100, 139, 200, 239]            # includes rays that hit the circle
    ref_paths = []
    ref = P.shoot(so, [rows[k] for k in pick], 1e-2, obstacle=P.Circle(0.3), time_unit=1, paths=ref_paths)
    assert any(o[4] > 0 for o in ref)                             # reflections present
    m, paths = ray_amd.ScatterRelationObstaclePlot(ray_amd.SinProd.obstacle_demo(), None, ray_amd.Circle(0.3), None,
                                                   ns, nd, 1e-2, pick, ctx=ctx)
    for k, (p, rp, o) in enumerate(zip(paths, ref_paths, ref)):
        rp = np.array(rp)
        assert p.shape == rp.shape, (k, p.shape, rp.shape)
        assert np.max(np.abs(p - rp)) <= 1e-10
        assert p.shape[0] == o[3] + 1 + o[4]                      # start + one per step (+1 per reflection)
        assert np.max(np.abs(m[pick[k], 4:8] - np.array(o[0]))) <= 1e-10
        assert abs(np.hypot(*p[-1]) - 1.0) < 1e-12                # ends on the unit circle
    untouched = np.setdiff1d(np.arange(ns * nd), pick)
    assert not m[untouched, 4:].any()


def test_paths_grid_exact_bits(ctx, orc):
    """gridded medium, explicit states, exact flavour: every recorded point is bit-identical to the
    restatement's (no transcendental on this path)."""
    import ray_amd
    P = _py()
    from conftest import golden
    g = golden("grid_n75_ns8_nd16.npz")
    N = int(g["N"])
    grid = P.Grid(list(np.asarray(g["c"]).flatten(order="F")), N, float(g["ext"]))
    rows = [list(r) for r in g["m"][:24, 0:4]]
    ref_paths = []
    ref = P.shoot(grid, rows, 1e-2, paths=ref_paths)
    m = np.zeros((24, 9), order="F")
    m[:, 0:4] = g["m"][:24, 0:4]
    med = ray_amd.GridMedium(np.asfortranarray(g["c"]), float(g["ext"]), ctx)
    ctx.set_arith(ray_amd.ARITH_EXACT)
    try:
        paths, info = ray_amd.shoot_paths(med, m, 1e-2, ctx=ctx, max_pts=512)
    finally:
        ctx.set_arith(ray_amd.ARITH_FUSED)
    for p, rp in zip(paths, ref_paths):
        assert np.array_equal(p, np.array(rp))
    assert np.array_equal(m[:, 4:8], np.array([o[0] for o in ref]))
    with pytest.raises(ray_amd.RayError):
        ray_amd.shoot_paths(med, m, 1e-2, ctx=ctx, max_pts=8)


def test_nonreflection_plot_exact_bits(ctx, orc):
    """NonReflectionPlot's polylines (Obstacle.jl:222-274): the overshoot trace, start + every step,
    bit-identical to the restatement in the exact flavour."""
    import ray_amd
    P = _py()
    from conftest import golden
    g = golden("grid_n75_ns8_nd16.npz")
    N = int(g["N"])
    grid = P.Grid(list(np.asarray(g["c"]).flatten(order="F")), N, float(g["ext"]))
    rows = [list(r) for r in g["m"][:32, 0:4]]
    T = [float(t) for t in g["m"][:32, 8]]
    ref_paths = []
    ref = P.trace_fixed_time(grid, rows, T, 5e-2, 1, paths=ref_paths)
    m = np.asfortranarray(g["m"][:32].copy())
    ctx.set_arith(ray_amd.ARITH_EXACT)
    try:
        paths, info = ray_amd.NonReflectionPlot(np.asfortranarray(g["c"]), m, float(g["ext"]), 5e-2, ctx=ctx,
                                                max_pts=128, info=True)
    finally:
        ctx.set_arith(ray_amd.ARITH_FUSED)
    assert len(paths) == 32
    for p, rp in zip(paths, ref_paths):
        assert np.array_equal(p, np.array(rp))
    assert np.array_equal(info["X"], np.array([o[0] for o in ref]))
    assert np.array_equal(info["npts"], info["nsteps"] + 1)
    assert np.array_equal(info["X"], g["X_overshoot"][:32])
    with pytest.raises(ray_amd.RayError):
        ray_amd.NonReflectionPlot(np.asfortranarray(g["c"]), m, float(g["ext"]), 5e-2, ctx=ctx, max_pts=4)


def test_paths_trefoil_radial(ctx):
    """polylines around the trefoil obstacle in the radial-sine-sum medium (both commented alternates of
    ObstacleDemo.jl:8-34 at once): rays with up to several reflections, reflection points on the level set."""
    import ray_amd
    P = _py()
    med, tf = P.RadialSineSum.obstacle_alt(), P.Trefoil(0.4, 0.3, 0.2)
    ns, nd = 8, 16
    rows = P.init_states(med, ns, nd, 0.0, np.pi)
    ref_all = P.shoot(med, rows, 1e-2, obstacle=tf, time_unit=1)
    pick = [k for k, o in enumerate(ref_all) if o[4] >= 1][:12]
    assert len(pick) >= 6
    ref_paths = []
    ref = P.shoot(med, [rows[k] for k in pick], 1e-2, obstacle=tf, time_unit=1, paths=ref_paths)
    m, paths = ray_amd.ScatterRelationObstaclePlot(ray_amd.RadialSineSum.obstacle_alt(), None, ray_amd.Trefoil.demo(),
                                                   None, ns, nd, 1e-2, pick, ctx=ctx)
    for k, (p, rp, o) in enumerate(zip(paths, ref_paths, ref)):
        rp = np.array(rp)
        assert p.shape == rp.shape, (k, p.shape, rp.shape)
        assert np.max(np.abs(p - rp)) <= 1e-10
        assert np.max(np.abs(m[pick[k], 4:8] - np.array(o[0]))) <= 1e-10
    # every reflection point sits on the level set r = R - rho cos 3 theta
    hits = 0
    for p, rp in zip(paths, ref_paths):
        for q in p[1:-1]:
            if abs(tf.ob(q[0], q[1])) < 1e-14:
                hits += 1
    assert hits >= len(pick)
