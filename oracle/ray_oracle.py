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
from math import sin, cos, sqrt, floor, pi

ST_EXITED, ST_STEP_CAP, ST_BISECT_CAP, ST_OOB, ST_CEASED = 1, 2, 4, 8, 16
BISECT_CAP = 200


def linspace(a, b, n):
    """Julia 0.6 linspace(a,b,n) with TwicePrecision indexing == correctly rounded exact value."""
    fa, fb = Fraction(float(a)), Fraction(float(b))
    if n == 1:
        return [float(a)]
    return [float(fa + (fb - fa) * (k - 1) / (n - 1)) for k in range(1, n + 1)]


# ---- media (NonObstacleDemo.jl:12,23; ObstacleDemo.jl:12,18) --------------------------------
class SinProd:
    """c = 1 + A sin(wx x) sin(wy y);  grad c = [gx cos sin, gy sin cos]."""

    def __init__(self, A, wx, wy, gx, gy):
        self.A, self.wx, self.wy, self.gx, self.gy = A, wx, wy, gx, gy

    @staticmethod
    def smooth_demo():  # NonObstacleDemo.jl:12,23
        return SinProd(0.3, 1.5 * pi, 1.5 * pi, 0.45 * pi, 0.45 * pi)

    @staticmethod
    def obstacle_demo():  # ObstacleDemo.jl:12,18
        return SinProd(0.2, 2 * pi, pi, 2 * (0.2 * pi), 0.2 * pi)

    def c(self, x, y):
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
    for i in range(src_begin, src_end):
        sx, sy = cos(source[i]), sin(source[i])
        cs = med.c(sx, sy)
        for j in range(nd):
            theta = source[i] + pi * 0.5 + direct[j]
            rows.append([sx, sy, cos(theta) / cs, sin(theta) / cs])
    return rows


def _rk4_kdt(med, X, dt):  # NonObstacle.jl:72-76
    oob = 0
    H, o = med.hamilton(X); oob |= o
    k1 = [h * dt for h in H]
    H, o = med.hamilton([X[k] + k1[k] / 2 for k in range(4)]); oob |= o
    k2 = [h * dt for h in H]
    H, o = med.hamilton([X[k] + k2[k] / 2 for k in range(4)]); oob |= o
    k3 = [h * dt for h in H]
    H, o = med.hamilton([X[k] + k3[k] for k in range(4)]); oob |= o
    k4 = [h * dt for h in H]
    return [X[k] + (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) / 6.0 for k in range(4)], oob


def shoot(med, rows, dt, obstacle=None, time_unit=0, max_steps=100000):
    """a6 (NonObstacle.jl:67-97) / a7 (Obstacle.jl:23-95).  rows: list of [x,y,px,py].
    Returns list of (X1[4], t, status, nsteps, nreflect)."""
    out = []
    for X0 in rows:
        X = list(X0)
        t = 0
        rfl = st = steps = nref = 0
        while True:
            if steps >= max_steps:
                st |= ST_STEP_CAP
                break
            if obstacle is not None and obstacle.ob(X[0], X[1]) > 0:
                rfl = 0
            prev = X
            X, o = _rk4_kdt(med, X, dt)
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
                n = obstacle.normal(R[0], R[1])
                tn = [2 * n[0], 2 * n[1]]
                M = [[1 - tn[0] * n[0], 0 - tn[0] * n[1]], [0 - tn[1] * n[0], 1 - tn[1] * n[1]]]
                X = [R[0], R[1], M[0][0] * R[2] + M[0][1] * R[3], M[1][0] * R[2] + M[1][1] * R[3]]
                nref += 1
        out.append((X, float(t), st, steps, nref))
    return out


def _rk4_k(med, X, dt):  # NonObstacle.jl:161-164,171 ; Obstacle.jl:312-317
    oob = 0
    k1, o = med.hamilton(X); oob |= o
    k2, o = med.hamilton([X[k] + k1[k] / 2 * dt for k in range(4)]); oob |= o
    k3, o = med.hamilton([X[k] + k2[k] / 2 * dt for k in range(4)]); oob |= o
    k4, o = med.hamilton([X[k] + k3[k] * dt for k in range(4)]); oob |= o
    return [X[k] + (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) * dt / 6.0 for k in range(4)], oob


def trace_fixed_time(grid, rows, T, timeStep, mode, step_cap=10000000):
    """mode 0 = a9 X-only (NonObstacle.jl:146-157,161,171,177-181,185);
    mode 1 = a8 (Obstacle.jl:306-326).  Returns list of (X[4], status, nsteps)."""
    p = grid.p
    cease_r = 1 + 2 * (p[1] - p[0])
    out = []
    for X0, Ti in zip(rows, T):
        X = list(X0)
        t, dt, st, steps = 0, timeStep, 0, 0
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
        out.append((X, st, steps))
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
