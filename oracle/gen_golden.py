"""Generates tests/golden/*.npz from the pure-Python restatement (oracle/ray_oracle.py).

Run from the repo root:  python oracle/gen_golden.py            (all fixtures)
                         python oracle/gen_golden.py trefoil    (only the trefoil one, added later)
                         python oracle/gen_golden.py radial     (only the radial-sine-sum ones, added later)
                         python oracle/gen_golden.py trefoil_grid   (grid + trefoil "identity" flavour, round 2)
                         python oracle/gen_golden.py rep        (analytic media, reproducible sin/cos flavour, round 2)
                         python oracle/gen_golden.py fused      (analytic media, the kernels' default fused flavour, round 3)
The fixtures are data only (inputs + expected outputs).  They pin the C oracle and the HIP
path to the Python reading of the reference; the reference itself ships no vectors
("parity unpinned" by the reference).
"""
import os
import sys
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ray_oracle as P  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def pack(rows, out):
    m = np.array([list(r) + list(o[0]) + [o[1]] for r, o in zip(rows, out)], dtype=np.float64)
    return dict(m=m, status=np.array([o[2] for o in out], dtype=np.int32),
                nsteps=np.array([o[3] for o in out], dtype=np.int32),
                nreflect=np.array([o[4] for o in out], dtype=np.int32))


def trefoil():
    # G1d: obstacle demo medium + the commented trefoil obstacle (ObstacleDemo.jl:23-24,30-31): cavities,
    #      several reflections per ray
    ns, nd, dt = 8, 16, 1e-2
    so = P.SinProd.obstacle_demo()
    rows = P.init_states(so, ns, nd, 0.0, pi)
    tf = P.Trefoil(0.4, 0.3, 0.2)
    np.savez(os.path.join(OUT, "scatter_trefoil_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt,
             trefoil=np.array([tf.R, tf.rho, tf.rho_n]),
             params=np.array([so.A, so.wx, so.wy, so.gx, so.gy]),
             **pack(rows, P.shoot(so, rows, dt, obstacle=tf, time_unit=1)))


def trefoil_grid():
    # G1f: a7 loop over a GRIDDED medium around the trefoil, "identity" flavour of its cos/sin 3 theta
    #      (x/r, y/r + triple-angle identities: no transcendental anywhere on this path, so the C oracle and
    #      the HIP kernels must reproduce it bit for bit, events included)
    ns, nd, dt, N, ext = 8, 16, 1e-2, 33, 1.5
    so = P.SinProd.obstacle_demo()
    p = P.linspace(-ext, ext, N)
    c = [so.c(p[i], p[j]) for j in range(N) for i in range(N)]
    g = P.Grid(c, N, ext)
    rows = P.init_states(g, ns, nd, 0.0, pi)
    tf = P.Trefoil(0.4, 0.3, 0.2, trig="identity")
    np.savez(os.path.join(OUT, "grid_trefoil_identity_n33_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt, N=N, ext=ext,
             c=np.array(c).reshape(N, N, order="F"), trefoil=np.array([tf.R, tf.rho, tf.rho_n]),
             **pack(rows, P.shoot(g, rows, dt, obstacle=tf, time_unit=1)))


def radial():
    # G1e: the commented alternate media (NonObstacleDemo.jl:7-11,15-22 four terms, ScatterRelation;
    #      ObstacleDemo.jl:9-11,15-17 two terms with the shipped circle, ScatterRelationObstacle)
    ns, nd, dt = 8, 16, 1e-2
    for name, med, ob, tu in (("smooth", P.RadialSineSum.smooth_alt(), None, 0),
                              ("obstacle", P.RadialSineSum.obstacle_alt(), P.Circle(0.3), 1)):
        rows = P.init_states(med, ns, nd, 0.0, pi)
        np.savez(os.path.join(OUT, "scatter_radial_%s_ns8_nd16.npz" % name), ns=ns, nd=nd, dt=dt,
                 terms=np.array(med.terms), r0=0.3 if ob else 0.0, time_unit=tu,
                 **pack(rows, P.shoot(med, rows, dt, obstacle=ob, time_unit=tu)))


def rep():
    # G1g: the shipped analytic media under the reproducible sin/cos flavour (ray_oracle.py rep_sincos / rep_shift:
    #      the sequence of roundings the HIP kernels and the C oracle's trig = 1 evaluate) -- the smooth demo, and the
    #      obstacle demo around the trefoil (identity flavour): no math library anywhere on these paths, so the C
    #      oracle and the HIP kernels must reproduce every bit and every event
    ns, nd, dt = 8, 16, 1e-2
    sp = P.SinProd.smooth_demo()
    sp.trig = "rep"
    rows = P.init_states(sp, ns, nd, 0.0, pi)
    np.savez(os.path.join(OUT, "scatter_smooth_rep_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt,
             params=np.array([sp.A, sp.wx, sp.wy, sp.gx, sp.gy]), **pack(rows, P.shoot(sp, rows, dt)))
    so = P.SinProd.obstacle_demo()
    so.trig = "rep"
    rows = P.init_states(so, ns, nd, 0.0, pi)
    tf = P.Trefoil(0.4, 0.3, 0.2, trig="identity")
    np.savez(os.path.join(OUT, "scatter_trefoil_rep_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt,
             trefoil=np.array([tf.R, tf.rho, tf.rho_n]),
             params=np.array([so.A, so.wx, so.wy, so.gx, so.gy]),
             **pack(rows, P.shoot(so, rows, dt, obstacle=tf, time_unit=1)))


def fused():
    # G1h (round 3): the shipped analytic media under the kernels' DEFAULT flavour of Hamilton -- multiply-adds contracted,
    #      sin/cos anchors shared by consecutive steps (ray_oracle.py hamilton_fused / _rk4_fused, exact-rational fma) --
    #      the smooth demo, and the obstacle demo around the shipped circle: the C oracle (arith = 1, trig = 1) and the
    #      HIP kernels must reproduce every bit and every event
    ns, nd, dt = 8, 16, 1e-2
    sp = P.SinProd.smooth_demo()
    sp.trig = "fused"
    rows = P.init_states(sp, ns, nd, 0.0, pi)
    np.savez(os.path.join(OUT, "scatter_smooth_fused_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt,
             params=np.array([sp.A, sp.wx, sp.wy, sp.gx, sp.gy]), **pack(rows, P.shoot(sp, rows, dt)))
    so = P.SinProd.obstacle_demo()
    so.trig = "fused"
    rows = P.init_states(so, ns, nd, 0.0, pi)
    np.savez(os.path.join(OUT, "scatter_obstacle_fused_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt, r0=0.3,
             params=np.array([so.A, so.wx, so.wy, so.gx, so.gy]),
             **pack(rows, P.shoot(so, rows, dt, obstacle=P.Circle(0.3), time_unit=1)))


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["fused"]:
        return fused()
    if sys.argv[1:] == ["rep"]:
        return rep()
    if sys.argv[1:] == ["trefoil"]:
        return trefoil()
    if sys.argv[1:] == ["radial"]:
        return radial()
    if sys.argv[1:] == ["trefoil_grid"]:
        return trefoil_grid()
    trefoil()
    trefoil_grid()
    radial()
    rep()
    fused()
    ns, nd, dt = 8, 16, 1e-2

    # G1a: smooth demo medium, analytic, ScatterRelation (NonObstacleDemo.jl:32)
    sp = P.SinProd.smooth_demo()
    rows = P.init_states(sp, ns, nd, 0.0, pi)
    np.savez(os.path.join(OUT, "scatter_smooth_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt,
             params=np.array([sp.A, sp.wx, sp.wy, sp.gx, sp.gy]), **pack(rows, P.shoot(sp, rows, dt)))

    # G1b: obstacle demo medium + circle 0.3, ScatterRelationObstacle (ObstacleDemo.jl:43)
    so = P.SinProd.obstacle_demo()
    rows = P.init_states(so, ns, nd, 0.0, pi)
    np.savez(os.path.join(OUT, "scatter_obstacle_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt, r0=0.3,
             params=np.array([so.A, so.wx, so.wy, so.gx, so.gy]),
             **pack(rows, P.shoot(so, rows, dt, obstacle=P.Circle(0.3), time_unit=1)))

    # G1c: gridded N=75, ext=1.5 (NonObstacleDemo.jl:36): tables, shoot-to-exit on the grid,
    #      then both fixed-time traces at the reconstruction step 5e-2 (NonObstacleDemo.jl:29)
    N, ext = 75, 1.5
    p = P.linspace(-ext, ext, N)
    c = [sp.c(p[i], p[j]) for j in range(N) for i in range(N)]       # column-major, first index x
    g = P.Grid(c, N, ext)
    rows = P.init_states(g, ns, nd, 0.0, pi)
    shot = P.shoot(g, rows, dt)
    pk = pack(rows, shot)
    T = [o[1] for o in shot]
    tr0 = P.trace_fixed_time(g, rows, T, 5e-2, 0)
    tr1 = P.trace_fixed_time(g, rows, T, 5e-2, 1)
    ev = np.array([[g.ev[I, J] for J in range(1, N)] for I in range(1, N)])          # (N-1,N-1,4)
    gr = np.array([[g.gr[I, J] for J in range(1, N - 1)] for I in range(1, N - 1)])  # (N-2,N-2,8)
    np.savez(os.path.join(OUT, "grid_n75_ns8_nd16.npz"), ns=ns, nd=nd, dt=dt, N=N, ext=ext,
             c=np.array(c).reshape(N, N, order="F"), p=np.array(p), eval=ev, grad=gr,
             trace_dt=5e-2,
             X_shorten=np.array([o[0] for o in tr0]), st_shorten=np.array([o[1] for o in tr0], dtype=np.int32),
             ns_shorten=np.array([o[2] for o in tr0], dtype=np.int32),
             X_overshoot=np.array([o[0] for o in tr1]), st_overshoot=np.array([o[1] for o in tr1], dtype=np.int32),
             ns_overshoot=np.array([o[2] for o in tr1], dtype=np.int32), **pk)

    # G2: a5 tables for the N=8 grid of the same medium (SURVEY.md section 8c pin G2)
    N2 = 8
    p2 = P.linspace(-ext, ext, N2)
    c2 = [sp.c(p2[i], p2[j]) for j in range(N2) for i in range(N2)]
    g2 = P.Grid(c2, N2, ext)
    ev2 = np.array([[g2.ev[I, J] for J in range(1, N2)] for I in range(1, N2)])
    gr2 = np.array([[g2.gr[I, J] for J in range(1, N2 - 1)] for I in range(1, N2 - 1)])
    np.savez(os.path.join(OUT, "tables_n8.npz"), N=N2, ext=ext, c=np.array(c2).reshape(N2, N2, order="F"),
             p=np.array(p2), eval=ev2, grad=gr2)

    # a11 misfit on the gridded case: data = shot, observation = SHORTEN_CEASE trace
    m58 = [list(o[0]) for o in shot]
    o58 = [list(o[0]) for o in tr0]
    sm, st, mask = P.misfit(m58, o58, 0.2)
    np.savez(os.path.join(OUT, "misfit_grid_n75.npz"), sum_mismatch2=sm, sum_target2=st,
             mask=np.array(mask, dtype=np.uint8))
    # NEXT-1: complete ScatterForwardOperator on an N=31 grid: X, rho, M (non-zeros of the dense result)
    N3, ns3, nd3 = 31, 4, 8
    p3 = P.linspace(-ext, ext, N3)
    c3 = [sp.c(p3[i], p3[j]) for j in range(N3) for i in range(N3)]
    g3 = P.Grid(c3, N3, ext)
    rows3 = P.init_states(g3, ns3, nd3, 0.0, pi)
    shot3 = P.shoot(g3, rows3, dt)
    T3 = [o[1] * (1.0 + 0.4 * k / len(shot3)) for k, o in enumerate(shot3)]      # late rays take the cease branch
    M3, X3, rho3 = P.forward_operator(g3, rows3, T3, 5e-2)
    r_, c_ = np.nonzero(M3)
    np.savez(os.path.join(OUT, "frechet_n31_ns4_nd8.npz"), N=N3, ext=ext, trace_dt=5e-2,
             c=np.array(c3).reshape(N3, N3, order="F"), rows=np.array([list(r) for r in rows3]), T=np.array(T3),
             X=np.array(X3), rho=np.array(rho3), M_rows=r_.astype(np.int32), M_cols=c_.astype(np.int32),
             M_vals=M3[r_, c_])
    print("golden fixtures written to", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
