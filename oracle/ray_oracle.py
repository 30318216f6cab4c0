"""Independent pure-Python restatement of lowrank/ray's forward ray-shooting path.

TEST INFRASTRUCTURE ONLY: imported by tests/ and oracle/gen_golden.py, never by ray_amd/.
"parity unpinned" by the reference (no tests, no golden vectors, Julia 0.6 not runnable here;
SURVEY.md section 8c).  This file exists so that the C oracle (ray_oracle.c) is checked against a
second, separately written reading of the same Julia source; the two must agree bit for bit
(Python floats are IEEE binary64, CPython never fuses a*b+c, math.sin/cos/sqrt are the libm
the C oracle links).  It also generates tests/golden/*.npz (oracle/gen_golden.py).

Citations are file:line under /root/reference.  Indices below are 1-based where the Julia code's
are, to keep the reading literal; arrays are plain Python lists (column lists for m).
"""
from fractions import Fraction
from math import sin, cos, sqrt, floor, pi, atan2

ST_EXITED, ST_STEP_CAP, ST_BISECT_CAP, ST_OOB, ST_CEASED = 1, 2, 4, 8, 16
BISECT_CAP = 200


def linspace(a, b, n):
    """Julia 0.6 linspace(a,b,n) with TwicePrecision indexing == correctly rounded exact value."""
    fa, fb = Fraction(float(a)), Fraction(float(b))
    if n == 1:
        return [float(a)]
    return [float(fa + (fb - fa) * (k - 1) / (n - 1)) for k in range(1, n + 1)]


# ---- the "reproducible" sin/cos flavour (ray_oracle.c, med->trig == 1), read a second time ---------------
# Every operation one IEEE rounding; fma through exact rationals (CPython has no fma before 3.13).
def _fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))       # Fraction -> float rounds correctly


_S = [-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
      2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10]
_C = [4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
      -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11]
_TS = [-1 / 6, 1 / 120, -1 / 5040, 1 / 362880, -1 / 39916800]
_TC = [0.5, -1 / 24, 1 / 720, -1 / 40320, 1 / 3628800, -1 / 479001600]
REP_BOUND, REP_SHIFT_MAX = 262144.0, 0.25


def rep_sincos(a):
    """(sin a, cos a): Cody-Waite reduction by pi/2 (33-bit head) + fdlibm's polynomials on [-pi/4, pi/4]."""
    if not abs(a) < REP_BOUND:
        return sin(a), cos(a)
    fn = float(round(Fraction(a * 6.36619772367581382433e-01)))     # to nearest, ties to even
    n = int(fn)
    r = _fma(-fn, 1.57079632673412561417e+00, a)
    w = fn * 6.07710050650619224932e-11
    y = r - w
    t = (r - y) - w
    z = y * y
    v = z * y
    p = _S[5]
    for k in (4, 3, 2, 1):
        p = _fma(z, p, _S[k])
    u = _fma(-v, p, 0.5 * t)
    u = _fma(z, u, -t)
    u = _fma(-v, _S[0], u)
    ks = y - u
    q = _C[5]
    for k in (4, 3, 2, 1, 0):
        q = _fma(z, q, _C[k])
    q = z * q
    hz = 0.5 * z
    ww = 1.0 - hz
    kc = ww + (((1.0 - ww) - hz) + _fma(z, q, -(y * t)))
    ss, cc = (kc, ks) if n & 1 else (ks, kc)
    return (-ss if n & 2 else ss), (-cc if (n + 1) & 2 else cc)


def rep_shift(s0, c0, d):
    """(sin(a0 + d), cos(a0 + d)) from (sin a0, cos a0), |d| <= 1/4."""
    z = d * d
    ps = _TS[4]
    for k in (3, 2, 1, 0):
        ps = _fma(z, ps, _TS[k])
    sd = d + (d * z) * ps
    pc = _TC[5]
    for k in (4, 3, 2, 1, 0):
        pc = _fma(z, pc, _TC[k])
    v = z * pc
    return s0 + _fma(c0, sd, -(s0 * v)), c0 - _fma(s0, sd, c0 * v)


def rep_shift_fused(s0, c0, d):
    """rep_shift with its multiply-adds contracted (ray_trig.h shift_fused; ray_oracle.c orc_rep_shift_fused)."""
    z = d * d
    ps = _TS[4]
    for k in (3, 2, 1, 0):
        ps = _fma(z, ps, _TS[k])
    sd = _fma(d * z, ps, d)
    pc = _TC[5]
    for k in (4, 3, 2, 1, 0):
        pc = _fma(z, pc, _TC[k])
    v = z * pc
    return _fma(c0, sd, _fma(-s0, v, s0)), _fma(-s0, sd, _fma(-c0, v, c0))


def anchor_period(w, dt, cmax):
    """steps one full sin/cos evaluation anchors in the fused analytic flavour (ray_trig.h anchor_period)."""
    k = floor(0.24 / (abs(w) * dt * cmax))
    return 64 if k >= 64.0 else (int(k) if k >= 1.0 else 1)


REP_TINY_MAX = 0.0009765625


def rep_shift_tiny(s0, c0, d):
    """the advance by a tiny difference (ray_trig.h shift_tiny; ray_oracle.c orc_rep_shift_tiny)."""
    z = d * d
    sd = _fma(d * z, -1.66666666666666666667e-01, d)
    v = z * _fma(z, -4.16666666666666666667e-02, 0.5)
    return _fma(c0, sd, _fma(-s0, v, s0)), _fma(-s0, sd, _fma(-c0, v, c0))


def fused_stage(stage, anchor, w, x, base, slot):
    """sin/cos of w x in the fused analytic flavour: a full evaluation when this step re-anchors the argument (first
    stage only), else one contracted advance from the anchor -- or, for stage 3 and a later step's stage 1, the tiny
    advance from the stage before where the two differences from the anchor are within REP_TINY_MAX; beyond the
    advance's range: in full (a first stage re-anchors).  base[slot] = the anchor, base[("prev", slot)] = the stage
    before (set by stages 2 and 4)."""
    if stage == 0 and anchor:
        a = w * x
        sn, cs = rep_sincos(a)
        base[slot] = (a, sn, cs)
        return sn, cs
    a0, s0, c0 = base[slot]
    d = _fma(w, x, -a0)
    pd, ps, pc = base.get(("prev", slot), (0.0, 0.0, 0.0))
    dl = d - pd
    if stage in (0, 2) and abs(dl) <= REP_TINY_MAX:
        sn, cs = rep_shift_tiny(ps, pc, dl)
    elif abs(d) <= REP_SHIFT_MAX:
        sn, cs = rep_shift_fused(s0, c0, d)
    else:
        a = w * x
        sn, cs = rep_sincos(a)
        if stage == 0:
            base[slot] = (a, sn, cs)
    if stage in (1, 3):
        base[("prev", slot)] = (d, sn, cs)
    return sn, cs


def rep_stage(stage, a, base, slot):
    """sin/cos of a at RK4 stage 0..3: stage 0 evaluates and remembers, the others advance from it."""
    if stage == 0:
        sn, cs = rep_sincos(a)
        base[slot] = (a, sn, cs)
        return sn, cs
    a0, s0, c0 = base[slot]
    d = a - a0
    if abs(d) <= REP_SHIFT_MAX:
        return rep_shift(s0, c0, d)
    return rep_sincos(a)


# ---- media (NonObstacleDemo.jl:12,23; ObstacleDemo.jl:12,18) --------------------------------
class SinProd:
    """c = 1 + A sin(wx x) sin(wy y);  grad c = [gx cos sin, gy sin cos].
    trig = "libm": math.sin/cos (as the reference calls its own); trig = "rep": the reproducible flavour above, the
    sequence the HIP kernels' exact flavour evaluates (stages of an RK4 step through `hamilton_staged`); trig = "fused":
    the kernels' default -- Hamilton (@fastmath in the reference, NonObstacle.jl:1) with its multiply-adds contracted,
    sin/cos anchors shared by consecutive steps (`hamilton_fused`, `_rk4_fused`)."""

    def __init__(self, A, wx, wy, gx, gy, trig="libm"):
        self.A, self.wx, self.wy, self.gx, self.gy, self.trig = A, wx, wy, gx, gy, trig

    def hamilton_staged(self, X, stage, base):
        sx, cx = rep_stage(stage, self.wx * X[0], base, 0)
        sy, cy = rep_stage(stage, self.wy * X[1], base, 1)
        speed = 1 + self.A * sx * sy
        g = [self.gx * (cx * sy), self.gy * (sx * cy)]
        tau = X[2] * X[2] + X[3] * X[3]
        s2 = speed * speed
        return [s2 * X[2], s2 * X[3], ((-tau) * g[0]) * speed, ((-tau) * g[1]) * speed], 0

    def hamilton_fused(self, X, stage, anchor, base):
        sx, cx = fused_stage(stage, anchor & 1, self.wx, X[0], base, 0)
        sy, cy = fused_stage(stage, anchor & 2, self.wy, X[1], base, 1)
        speed = _fma(self.A * sx, sy, 1.0)
        tau = _fma(X[3], X[3], X[2] * X[2])
        s2, w = speed * speed, speed * tau
        return [s2 * X[2], s2 * X[3], ((-self.gx) * w) * (cx * sy), ((-self.gy) * w) * (sx * cy)]

    def anchor_periods(self, dt):
        cmax = 1 + abs(self.A)
        return [anchor_period(self.wx, dt, cmax), anchor_period(self.wy, dt, cmax)]

    @staticmethod
    def smooth_demo():  # NonObstacleDemo.jl:12,23
        return SinProd(0.3, 1.5 * pi, 1.5 * pi, 0.45 * pi, 0.45 * pi)

    @staticmethod
    def obstacle_demo():  # ObstacleDemo.jl:12,18
        return SinProd(0.2, 2 * pi, pi, 2 * (0.2 * pi), 0.2 * pi)

    def c(self, x, y):
        if self.trig in ("rep", "fused"):
            return 1 + self.A * rep_sincos(self.wx * x)[0] * rep_sincos(self.wy * y)[0]
        return 1 + self.A * sin(self.wx * x) * sin(self.wy * y)

    def gradc(self, x, y):
        return [self.gx * (cos(self.wx * x) * sin(self.wy * y)),
                self.gy * (sin(self.wx * x) * cos(self.wy * y))]

    def hamilton(self, X):  # NonObstacle.jl:1-4
        speed = self.c(X[0], X[1])
        g = self.gradc(X[0], X[1])
        tau = X[2] * X[2] + X[3] * X[3]
        s2 = speed * speed
        return [s2 * X[2], s2 * X[3], ((-tau) * g[0]) * speed, ((-tau) * g[1]) * speed], 0


class RadialSineSum:
    """The commented alternate media (NonObstacleDemo.jl:7-11,15-22; ObstacleDemo.jl:9-11,15-17):
    c = 1 + sum a_k sin(pi r_k), grad c = sum a_k pi cos(pi r_k)/r_k [(x - x_k), (y - y_k)]."""

    def __init__(self, terms):
        self.terms = [tuple(t) for t in terms]          # (a_k, x_k, y_k)

    @staticmethod
    def smooth_alt():    # NonObstacleDemo.jl:7-11
        return RadialSineSum([(0.2, 0.5, 0.2), (0.4, -0.4, -0.3), (0.2, 0.3, -0.4), (0.3, -0.2, 0.1)])

    @staticmethod
    def obstacle_alt():  # ObstacleDemo.jl:9-11
        return RadialSineSum([(0.4, 0.5, 0.2), (0.4, -0.4, -0.3)])

    def c(self, x, y):
        v = 1.0
        for a, xk, yk in self.terms:
            r = sqrt((x - xk) * (x - xk) + (y - yk) * (y - yk))
            v = v + a * sin(pi * r)
        return v

    def gradc(self, x, y):
        g = None
        for a, xk, yk in self.terms:
            dx, dy = x - xk, y - yk
            r = sqrt(dx * dx + dy * dy)
            f = ((a * pi) * cos(pi * r)) / r
            g = [f * dx, f * dy] if g is None else [g[0] + f * dx, g[1] + f * dy]
        return g

    def hamilton(self, X):  # NonObstacle.jl:1-4
        speed = self.c(X[0], X[1])
        g = self.gradc(X[0], X[1])
        tau = X[2] * X[2] + X[3] * X[3]
        s2 = speed * speed
        return [s2 * X[2], s2 * X[3], ((-tau) * g[0]) * speed, ((-tau) * g[1]) * speed], 0


class Circle:  # ObstacleDemo.jl:20-34
    def __init__(self, r0):
        self.r0 = r0

    def ob(self, x, y):
        return sqrt(x * x + y * y) - self.r0

    def normal(self, x, y):
        r = sqrt(x * x + y * y)
        n = [x / r, y / r]
        nn = sqrt(n[0] * n[0] + n[1] * n[1])
        return [n[0] / nn, n[1] / nn]


class Trefoil:  # the commented alternate of ObstacleDemo.jl:20-34
    """trig = "libm": atan2 / cos / sin as the reference writes them; trig = "identity": cos(theta) = x/r,
    sin(theta) = y/r and cos 3t = c (4 c^2 - 3), sin 3t = s (3 - 4 s^2) -- the evaluation the HIP kernels use
    (only +,-,*,/,sqrt: the same bits on every IEEE machine)."""

    def __init__(self, R=0.4, rho=0.3, rho_n=0.2, trig="libm"):
        self.R, self.rho, self.rho_n, self.trig = R, rho, rho_n, trig

    def ob(self, x, y):          # :21-24  r - (0.4 - rho * cos(3 * theta))
        r = sqrt(x * x + y * y)
        if self.trig == "identity":
            c = x / r if r > 0 else 1.0
            return r - (self.R - self.rho * (c * (4 * (c * c) - 3)))
        theta = atan2(y, x)
        return r - (self.R - self.rho * cos(3 * theta))

    def normal(self, x, y):      # :28-33  n = [x, y]/r - 3 * rho * sin(3 * theta)/r * [-y, x]/r; n/norm(n)
        r = sqrt(x * x + y * y)
        if self.trig == "identity":
            sn = y / r
            sin3 = sn * (3 - 4 * (sn * sn))
        else:
            sin3 = sin(3 * atan2(y, x))
        s = ((3 * self.rho_n) * sin3) / r
        n = [x / r - (s * (-y)) / r, y / r - (s * x) / r]
        nn = sqrt(n[0] * n[0] + n[1] * n[1])
        return [n[0] / nn, n[1] / nn]


class Grid:
    """Bilinear-table medium: tables built as NonObstacle.jl:191-210; RHS as :30-52."""

    def __init__(self, c, N, ext):
        # c: list of columns? no -- flat column-major list, c[(I-1) + (J-1)*N], first index = x
        self.N, self.ext = N, ext
        self.p = linspace(-ext, ext, N)
        dx = 2 * ext / (N - 1)
        C = lambda I, J: c[(I - 1) + (J - 1) * N]
        ev = {}
        for I in range(1, N):
            for J in range(1, N):  # Q4, NonObstacle.jl:103-105
                ev[I, J] = [C(I, J), (C(I + 1, J) - C(I, J)) / dx, (C(I, J + 1) - C(I, J)) / dx,
                            (C(I + 1, J + 1) + C(I, J) - C(I + 1, J) - C(I, J + 1)) / (dx * dx)]
        gr = {}
        for I in range(1, N - 1):
            for J in range(1, N - 1):
                gr[I, J] = [0.0] * 8          # zero-filled SharedArray, index 1 never written
        for I in range(2, N - 1):
            for J in range(2, N - 1):
                gr[I, J] = [(ev[I + 1, J][k] - ev[I - 1, J][k]) / (2 * dx) for k in range(4)] + \
                           [(ev[I, J + 1][k] - ev[I, J - 1][k]) / (2 * dx) for k in range(4)]
        self.ev, self.gr = ev, gr

    def _cell(self, x):
        p, N = self.p, self.N
        h = p[1] - p[0]
        q = (x - p[0]) / h
        if not (q == q):
            return 1, 1
        q = floor(q)
        if q < 0:
            return 1, 1
        if q > N - 3:
            return N - 2, 1
        return int(q) + 1, 0

    def c(self, x, y):
        I, _ = self._cell(x)
        J, _ = self._cell(y)
        dx, dy = x - self.p[I - 1], y - self.p[J - 1]
        z = [1.0, dx, dy, dx * dy]
        e = self.ev[I, J]
        s = z[0] * e[0]
        for k in range(1, 4):
            s = s + z[k] * e[k]
        return s

    def hamilton(self, X):  # NonObstacle.jl:30-42
        I, o1 = self._cell(X[0])
        J, o2 = self._cell(X[1])
        dx, dy = X[0] - self.p[I - 1], X[1] - self.p[J - 1]
        z = [1.0, dx, dy, dx * dy]
        tau = X[2] * X[2] + X[3] * X[3]

        def dot(v):
            s = z[0] * v[0]
            for k in range(1, 4):
                s = s + z[k] * v[k]
            return s
        c = dot(self.ev[I, J])
        g = self.gr[I, J]
        gcX, gcY = dot(g[0:4]), dot(g[4:8])
        c2 = c * c
        return [c2 * X[2], c2 * X[3], ((-gcX) * c) * tau, ((-gcY) * c) * tau], (o1 | o2)


# ---- a6 init (NonObstacle.jl:55-65) ---------------------------------------------------------
def init_states(med, ns, nd, dir_lo=0.0, dir_hi=pi, src_begin=0, src_end=None):
    src_end = ns if src_end is None else src_end
    source = linspace(0.0, 2 * pi, ns + 1)[:ns]
    direct = linspace(dir_lo, dir_hi, nd + 2)[1:nd + 1]
    rows = []
    rep = getattr(med, "trig", "libm") in ("rep", "fused")
    for i in range(src_begin, src_end):
        sx, sy = cos(source[i]), sin(source[i])
        if rep:
            sy, sx = rep_sincos(source[i])
        cs = med.c(sx, sy)
        for j in range(nd):
            theta = source[i] + pi * 0.5 + direct[j]
            st, ct = rep_sincos(theta) if rep else (sin(theta), cos(theta))
            rows.append([sx, sy, ct / cs, st / cs])
    return rows


def _rk4_fused(med, X, dt, state):
    """One RK4 step of the fused analytic flavour: Y = fma(H, dt/2, X), Xn = fma(sum, dt RN(1/6), X) (the gridded fused
    flavour's combination); `state` carries the ray's sin/cos anchors and countdowns from step to step."""
    if not state:
        state.update(base={}, acnt=[0, 0], K=med.anchor_periods(dt))
    anchor = 0
    for k in range(2):
        if state["acnt"][k] == 0:
            anchor |= 1 << k
            state["acnt"][k] = state["K"][k]
        state["acnt"][k] -= 1
    hdt, dt6 = 0.5 * dt, dt * 0.16666666666666666
    H = med.hamilton_fused(X, 0, anchor, state["base"])
    acc = list(H)
    H = med.hamilton_fused([_fma(H[k], hdt, X[k]) for k in range(4)], 1, anchor, state["base"])
    acc = [_fma(2.0, H[k], acc[k]) for k in range(4)]
    H = med.hamilton_fused([_fma(H[k], hdt, X[k]) for k in range(4)], 2, anchor, state["base"])
    acc = [_fma(2.0, H[k], acc[k]) for k in range(4)]
    H = med.hamilton_fused([_fma(H[k], dt, X[k]) for k in range(4)], 3, anchor, state["base"])
    return [_fma(acc[k] + H[k], dt6, X[k]) for k in range(4)], 0


def _rk4_kdt(med, X, dt, state=None):  # NonObstacle.jl:72-76
    oob = 0
    if getattr(med, "trig", "libm") == "fused":
        return _rk4_fused(med, X, dt, state)
    if getattr(med, "trig", "libm") == "rep" and hasattr(med, "hamilton_staged"):
        base, stage = {}, iter(range(4))
        rhs = lambda Y: med.hamilton_staged(Y, next(stage), base)
    else:
        rhs = med.hamilton
    H, o = rhs(X); oob |= o
    k1 = [h * dt for h in H]
    H, o = rhs([X[k] + k1[k] / 2 for k in range(4)]); oob |= o
    k2 = [h * dt for h in H]
    H, o = rhs([X[k] + k2[k] / 2 for k in range(4)]); oob |= o
    k3 = [h * dt for h in H]
    H, o = rhs([X[k] + k3[k] for k in range(4)]); oob |= o
    k4 = [h * dt for h in H]
    return [X[k] + (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) / 6.0 for k in range(4)], oob


def shoot(med, rows, dt, obstacle=None, time_unit=0, max_steps=100000, paths=None):
    """a6 (NonObstacle.jl:67-97) / a7 (Obstacle.jl:23-95).  rows: list of [x,y,px,py].
    Returns list of (X1[4], t, status, nsteps, nreflect).  paths (a list) receives, per ray, the
    polyline `res` of the *Plot tracers (Obstacle.jl:114,142,175,211): start, every step's position,
    reflection points, the exit point."""
    out = []
    for X0 in rows:
        X = list(X0)
        res = [(X[0], X[1])]
        t = 0
        rfl = st = steps = nref = 0
        state = {}                                       # the fused analytic flavour's per-ray sin/cos anchors
        while True:
            if steps >= max_steps:
                st |= ST_STEP_CAP
                break
            if obstacle is not None and obstacle.ob(X[0], X[1]) > 0:
                rfl = 0
            prev = X
            X, o = _rk4_kdt(med, X, dt, state)
            if o:
                st |= ST_OOB
            steps += 1
            t = t + 1 if time_unit else t + dt
            if sqrt(X[0] * X[0] + X[1] * X[1]) >= 1:
                lo, hi, it = 0.0, 1.0, 0
                mid = (lo + hi) / 2
                E = [(1 - mid) * prev[k] + mid * X[k] for k in range(4)]
                e = sqrt(E[0] * E[0] + E[1] * E[1]) - 1
                while abs(e) > 1e-15:
                    it += 1
                    if it > BISECT_CAP:
                        st |= ST_BISECT_CAP
                        break
                    if e > 0:
                        hi = mid
                    else:
                        lo = mid
                    mid = (lo + hi) / 2
                    E = [(1 - mid) * prev[k] + mid * X[k] for k in range(4)]
                    e = sqrt(E[0] * E[0] + E[1] * E[1]) - 1
                t = t - (1 - mid) if time_unit else t - (1 - mid) * dt
                res.append((E[0], E[1]))
                X = E
                st |= ST_EXITED
                break
            if obstacle is not None and obstacle.ob(X[0], X[1]) <= 0 and rfl == 0:
                lo, hi, it = 0.0, 1.0, 0
                rfl = 1
                mid = (lo + hi) / 2
                R = [(1 - mid) * prev[k] + mid * X[k] for k in range(4)]
                e = obstacle.ob(R[0], R[1])
                while abs(e) > 1e-15:
                    it += 1
                    if it > BISECT_CAP:
                        st |= ST_BISECT_CAP
                        break
                    if e > 0:
                        lo = mid
                    else:
                        hi = mid
                    mid = (lo + hi) / 2
                    R = [(1 - mid) * prev[k] + mid * X[k] for k in range(4)]
                    e = obstacle.ob(R[0], R[1])
                res.append((R[0], R[1]))
                n = obstacle.normal(R[0], R[1])
                tn = [2 * n[0], 2 * n[1]]
                M = [[1 - tn[0] * n[0], 0 - tn[0] * n[1]], [0 - tn[1] * n[0], 1 - tn[1] * n[1]]]
                X = [R[0], R[1], M[0][0] * R[2] + M[0][1] * R[3], M[1][0] * R[2] + M[1][1] * R[3]]
                nref += 1
            res.append((X[0], X[1]))
        out.append((X, float(t), st, steps, nref))
        if paths is not None:
            paths.append(res)
    return out


def _rk4_k(med, X, dt):  # NonObstacle.jl:161-164,171 ; Obstacle.jl:312-317
    oob = 0
    k1, o = med.hamilton(X); oob |= o
    k2, o = med.hamilton([X[k] + k1[k] / 2 * dt for k in range(4)]); oob |= o
    k3, o = med.hamilton([X[k] + k2[k] / 2 * dt for k in range(4)]); oob |= o
    k4, o = med.hamilton([X[k] + k3[k] * dt for k in range(4)]); oob |= o
    return [X[k] + (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) * dt / 6.0 for k in range(4)], oob


def trace_fixed_time(grid, rows, T, timeStep, mode, step_cap=10000000, paths=None):
    """mode 0 = a9 X-only (NonObstacle.jl:146-157,161,171,177-181,185);
    mode 1 = a8 (Obstacle.jl:306-326).  Returns list of (X[4], status, nsteps).  paths (a list) receives
    the polyline `res` of NonReflectionPlot per ray (Obstacle.jl:256,266): start, then every step."""
    p = grid.p
    cease_r = 1 + 2 * (p[1] - p[0])
    out = []
    for X0, Ti in zip(rows, T):
        X = list(X0)
        t, dt, st, steps = 0, timeStep, 0, 0
        res = [(X[0], X[1])]
        while t < Ti:
            if steps >= step_cap:
                st |= ST_STEP_CAP
                break
            if mode == 0 and (t + dt) > Ti:
                dt = Ti - t
            t = t + dt
            X, o = _rk4_k(grid, X, dt)
            if o:
                st |= ST_OOB
            steps += 1
            if mode == 0 and sqrt(X[0] * X[0] + X[1] * X[1]) > cease_r:
                t = t + (Ti - t)
                st |= ST_CEASED
            res.append((X[0], X[1]))
        out.append((X, st, steps))
        if paths is not None:
            paths.append(res)
    return out


def misfit(m58, obs58, thresh):
    """a11.  m58/obs58: lists of 4-lists (columns 5..8).  Returns (sum d^2, sum target^2, mask)."""
    sm = st = 0.0
    mask = []
    for a, b in zip(m58, obs58):
        d = [a[k] - b[k] for k in range(4)]
        for k in range(4):
            sm = sm + d[k] * d[k]
            st = st + a[k] * a[k]
        nd_ = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3])
        na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3])
        mask.append(1 if nd_ / na < thresh else 0)
    return sm, st, mask


# ---- NEXT-1: Jacobian rho and Frechet matrix M (NonObstacle.jl:6-28,107-139,146-185,212-218) ----------
# Written against the reference text independently of ray_oracle.c, with numpy's LAPACK-backed
# inv / matmul for the 4x4 algebra (what the reference's `inv(rho) * ...` and `t1 * rho` call).
# The C oracle uses its own elimination and loop order, so the two agree to rounding (~1e-13
# relative), not bit for bit; tests/test_oracle.py compares them at 1e-10.
def _hess_table(grid):
    """hess[I,J] (12 numbers) for I,J in 1..N-3; rows 1,2 stay zero (NonObstacle.jl:194,212-218)."""
    N, gr = grid.N, grid.gr
    dx = 2 * grid.ext / (N - 1)
    he = {(I, J): [0.0] * 12 for I in range(1, N - 2) for J in range(1, N - 2)}
    for I in range(3, N - 2):
        for J in range(3, N - 2):
            he[I, J] = [(gr[I + 1, J][k] - gr[I - 1, J][k]) / (2 * dx) for k in range(4)] + \
                       [(gr[I, J + 1][k] - gr[I, J - 1][k]) / (2 * dx) for k in range(4)] + \
                       [(gr[I, J + 1][4 + k] - gr[I, J - 1][4 + k]) / (2 * dx) for k in range(4)]
    return he


def _point(grid, X):
    import numpy as np
    I, _ = grid._cell(X[0])
    J, _ = grid._cell(X[1])
    h = grid.p[1] - grid.p[0]
    dx, dy = X[0] - grid.p[I - 1], X[1] - grid.p[J - 1]
    z = np.array([1.0, dx, dy, dx * dy])
    tau = X[2] * X[2] + X[3] * X[3]
    c = float(z @ np.array(grid.ev[I, J]))
    g = grid.gr[I, J]
    return I, J, h, dx, dy, z, tau, c, float(z @ np.array(g[0:4])), float(z @ np.array(g[4:8]))


def hamilton_jacobian(grid, he, X):
    """NonObstacle.jl:20-25: the 4x4 matrix the reference calls M (here Jm)."""
    import numpy as np
    I, J, h, dx, dy, z, tau, c, gcX, gcY = _point(grid, X)
    Ih, Jh = min(I, grid.N - 3), min(J, grid.N - 3)
    hs = he[Ih, Jh]
    hXX, hXY, hYY = float(z @ np.array(hs[0:4])), float(z @ np.array(hs[4:8])), float(z @ np.array(hs[8:12]))
    hh = np.array([[hXX, hXY], [hXY, hYY]])
    g = np.array([gcX, gcY])
    P = np.array([X[2], X[3]])
    return np.block([[2 * c * np.outer(P, g), c * c * np.eye(2)],
                     [-(c * hh + np.outer(g, g)) * tau, -2 * c * np.outer(g, P)]])


def dV(grid, X):
    """NonObstacle.jl:107-139 as a dense 4 x N^2 array (duplicates summed like sparse(), :137)."""
    import numpy as np
    N = grid.N
    I, J, h, dx, dy, z, tau, c, gcX, gcY = _point(grid, X)
    u, v = dx / h, dy / h
    phi = np.array([(1 - u) * (1 - v), u * (1 - v), (1 - u) * v, u * v])
    psi = np.array([u * (1 - v), -(1 - u) * (1 - v), u * v, -(1 - u) * v])
    gam = np.array([(1 - u) * v, u * v, -(1 - u) * (1 - v), -u * (1 - v)])
    cell = [I + (J - 1) * N, I + 1 + (J - 1) * N, I + J * N, I + 1 + J * N]
    Rind = [1] * 4 + [2] * 4 + [3] * 8 + [4] * 8
    Cind = cell + cell + cell + [I + 2 + (J - 1) * N, I - 1 + (J - 1) * N, I + 2 + J * N, I - 1 + J * N] + \
        cell + [I + (J + 1) * N, I + 1 + (J + 1) * N, I + (J - 2) * N, I + 1 + (J - 2) * N]
    Vind = np.concatenate([2 * c * X[2] * phi, 2 * c * X[3] * phi,
                           -tau * gcX * phi + c * tau * psi / (2 * h), -c * tau * psi / (2 * h),
                           -tau * gcY * phi + c * tau * gam / (2 * h), -c * tau * gam / (2 * h)])
    A = np.zeros((4, N * N))
    for r, cc, vv in zip(Rind, Cind, Vind):
        A[r - 1, cc - 1] += vv
    return A


def forward_operator(grid, rows, T, timeStep):
    """ChunkProcessing! (NonObstacle.jl:146-185) for every ray: returns (M (4num, N^2) dense,
    X list, rho list).  X follows the same expressions as trace_fixed_time(mode 0)."""
    import numpy as np
    he = _hess_table(grid)
    N = grid.N
    cease_r = 1 + 2 * (grid.p[1] - grid.p[0])
    M = np.zeros((4 * len(rows), N * N))
    Xs, rhos = [], []
    for i, (X0, Ti) in enumerate(zip(rows, T)):
        X = list(X0)
        t, dt = 0, timeStep
        rho = np.eye(4)
        Phi = np.zeros((4, N * N))
        while t < Ti:
            if (t + dt) > Ti:
                dt = Ti - t
            t = t + dt
            Phi += np.linalg.inv(rho) @ dV(grid, X) * dt / 2
            k1, _ = grid.hamilton(X); t1 = hamilton_jacobian(grid, he, X)
            Y = [X[k] + k1[k] / 2 * dt for k in range(4)]
            k2, _ = grid.hamilton(Y); t2 = hamilton_jacobian(grid, he, Y)
            Y = [X[k] + k2[k] / 2 * dt for k in range(4)]
            k3, _ = grid.hamilton(Y); t3 = hamilton_jacobian(grid, he, Y)
            Y = [X[k] + k3[k] * dt for k in range(4)]
            k4, _ = grid.hamilton(Y); t4 = hamilton_jacobian(grid, he, Y)
            v1 = t1 @ rho
            v2 = t2 @ (rho + v1 * dt / 2)
            v3 = t3 @ (rho + v2 * dt / 2)
            v4 = t4 @ (rho + v3 * dt)
            X = [X[k] + (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) * dt / 6.0 for k in range(4)]
            rho = rho + (v1 + 2 * v2 + 2 * v3 + v4) * dt / 6.0
            Phi += np.linalg.inv(rho) @ dV(grid, X) * dt / 2
            if sqrt(X[0] * X[0] + X[1] * X[1]) > cease_r:
                Phi += np.linalg.inv(rho) @ dV(grid, X) * (Ti - t)
                t = t + (Ti - t)
        M[4 * i:4 * i + 4, :] = rho @ Phi
        Xs.append(X)
        rhos.append(rho)
    return M, Xs, rhos
