"""ctypes/numpy front-end of the C oracle (oracle/libray_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by ray_amd/.  Arrays are numpy float64, Fortran (column-major) order,
laid out exactly like the reference's m / s / c / eval / grad (see ray_oracle.h).
"""
import ctypes as C
import os
import subprocess
import math
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libray_oracle.so")

ST_EXITED, ST_STEP_CAP, ST_BISECT_CAP, ST_OOB, ST_CEASED = 1, 2, 4, 8, 16


class SinProd(C.Structure):
    _fields_ = [("A", C.c_double), ("wx", C.c_double), ("wy", C.c_double),
                ("gx", C.c_double), ("gy", C.c_double)]


class Obstacle(C.Structure):
    _fields_ = [("kind", C.c_int), ("r0", C.c_double), ("rho_ob", C.c_double), ("rho_n", C.c_double),
                ("trig", C.c_int)]


def _obstacle(ob):
    """None, a circle radius, or ("trefoil", R, rho, rho_n[, trig]) with trig 0 = libm atan2/cos/sin as the
    reference writes them (default), 1 = the triple-angle identities on x/r, y/r the HIP kernels evaluate."""
    if ob is None:
        return Obstacle(0, 0.0, 0.0, 0.0, 0)
    if isinstance(ob, tuple):
        assert ob[0] == "trefoil"
        return Obstacle(2, float(ob[1]), float(ob[2]), float(ob[3]), int(ob[4]) if len(ob) > 4 else 0)
    return Obstacle(1, float(ob), 0.0, 0.0, 0)


def ob_val(ob, x, y):
    f = lib().orc_ob_val
    f.restype = C.c_double
    o = _obstacle(ob)
    return f(C.byref(o), C.c_double(x), C.c_double(y))


def ob_normal(ob, x, y):
    n = (C.c_double * 2)()
    o = _obstacle(ob)
    lib().orc_ob_normal(C.byref(o), C.c_double(x), C.c_double(y), n)
    return n[0], n[1]


class Radial(C.Structure):
    _fields_ = [("K", C.c_int), ("a", C.c_double * 4), ("x", C.c_double * 4), ("y", C.c_double * 4)]


class Medium(C.Structure):
    _fields_ = [("kind", C.c_int), ("sp", SinProd), ("N", C.c_int), ("ext", C.c_double),
                ("eval", C.c_void_p), ("grad", C.c_void_p), ("p", C.c_void_p), ("arith", C.c_int),
                ("rad", Radial), ("trig", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_shoot.restype = None
        _lib.orc_trace_fixed_time.restype = None
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


def linspace(a, b, n):
    out = np.empty(n, dtype=np.float64)
    lib().orc_linspace(C.c_double(a), C.c_double(b), C.c_int(n), _dp(out))
    return out


SMOOTH_DEMO = (0.3, 1.5 * np.pi, 1.5 * np.pi, 0.45 * np.pi, 0.45 * np.pi)   # NonObstacleDemo.jl:12,23
OBSTACLE_DEMO = (0.2, 2 * np.pi, np.pi, 2 * (0.2 * np.pi), 0.2 * np.pi)      # ObstacleDemo.jl:12,18


def sinprod_grid(params, N, ext):
    """c[i,j] = waveSpeed(p[i], p[j]) (NonObstacle.jl:306), Fortran order (N,N)."""
    A, wx, wy, _, _ = params
    p = linspace(-ext, ext, N)
    sx = np.array([math.sin(wx * float(v)) for v in p])
    sy = np.array([math.sin(wy * float(v)) for v in p])
    # 1 + A*sin(wx x)*sin(wy y): (A*sx)*sy then 1 + .
    c = 1 + (A * sx)[:, None] * sy[None, :]
    return np.asfortranarray(c)


class GridTables:
    def __init__(self, c, ext, arith=0):
        """arith: 0 = as-written (un-fused) arithmetic, 1 = the "fused" flavour (ray_oracle.c)."""
        self.arith = int(arith)
        c = np.asfortranarray(c, dtype=np.float64)
        N = c.shape[0]
        assert c.shape == (N, N)
        self.N, self.ext, self.c = N, float(ext), c
        self.p = linspace(-ext, ext, N)
        self.eval = np.zeros((N - 1, N - 1, 4), dtype=np.float64, order="F")
        self.grad = np.zeros((N - 2, N - 2, 8), dtype=np.float64, order="F")
        lib().orc_build_tables(_dp(c), C.c_int(N), C.c_double(ext), _dp(self.eval), _dp(self.grad))

    def hess(self):
        """hess (N-3,N-3,12), NonObstacle.jl:212-218 (built on first use)."""
        if getattr(self, "_hess", None) is None:
            N = self.N
            self._hess = np.zeros((N - 3, N - 3, 12), dtype=np.float64, order="F")
            lib().orc_build_hess(_dp(self.grad), C.c_int(N), C.c_double(self.ext), _dp(self._hess))
        return self._hess

    def medium(self):
        m = Medium()
        m.kind = 1
        m.N, m.ext = self.N, self.ext
        m.eval, m.grad, m.p = _dp(self.eval), _dp(self.grad), _dp(self.p)
        m.arith = self.arith
        return m

    def with_arith(self, arith):
        """Same tables, other arithmetic flavour (shares the arrays)."""
        import copy
        t = copy.copy(self)
        t.arith = int(arith)
        return t


def analytic_medium(params):
    m = Medium()
    m.kind = 0
    m.sp = SinProd(*params)
    return m


# the commented alternates: (a_k, x_k, y_k) terms of c = 1 + sum a_k sin(pi |x - x_k|)
SMOOTH_ALT = ((0.2, 0.5, 0.2), (0.4, -0.4, -0.3), (0.2, 0.3, -0.4), (0.3, -0.2, 0.1))     # NonObstacleDemo.jl:7-11
OBSTACLE_ALT = ((0.4, 0.5, 0.2), (0.4, -0.4, -0.3))                                       # ObstacleDemo.jl:9-11


def radial_medium(terms):
    m = Medium()
    m.kind = 2
    m.rad.K = len(terms)
    for k, (a, x, y) in enumerate(terms):
        m.rad.a[k], m.rad.x[k], m.rad.y[k] = a, x, y
    return m


def radial_grid(terms, N, ext):
    """c[i,j] = waveSpeed(p[i], p[j]) for the radial family, Fortran order (N,N)."""
    p = linspace(-ext, ext, N)
    c = np.empty((N, N), order="F")
    for i in range(N):
        for j in range(N):
            v = 1.0
            for a, xk, yk in terms:
                dx, dy = float(p[i]) - xk, float(p[j]) - yk
                v = v + a * math.sin(math.pi * math.sqrt(dx * dx + dy * dy))
            c[i, j] = v
    return c


def _medium(med, trig=0, arith=None):
    """trig: 0 = sin/cos of the host libm (as the reference calls its own), 1 = the reproducible flavour of
    ray_oracle.c, the sequence the HIP kernels evaluate (analytic media and the initial directions).
    arith (analytic media; a GridTables carries its own): 0 = Hamilton as written, 1 = the "fused" flavour of
    ray_oracle.c (multiply-adds contracted, sin/cos anchors shared by consecutive steps; needs trig = 1)."""
    if isinstance(med, tuple) and med and isinstance(med[0], tuple):
        m = radial_medium(med)
    elif isinstance(med, GridTables):
        m = med.medium()
    elif isinstance(med, Medium):
        m = med
    else:
        m = analytic_medium(med)
    m.trig = int(trig)
    if arith is not None and m.kind != 1:
        assert not arith or trig, "the fused analytic flavour is defined on the reproducible sin/cos (trig = 1)"
        m.arith = int(arith)
    return m


def rep_sincos(a):
    s, c = C.c_double(), C.c_double()
    lib().orc_rep_sincos(C.c_double(a), C.byref(s), C.byref(c))
    return s.value, c.value


def rep_shift_fused(s0, c0, d):
    s, c = C.c_double(), C.c_double()
    lib().orc_rep_shift_fused(C.c_double(s0), C.c_double(c0), C.c_double(d), C.byref(s), C.byref(c))
    return s.value, c.value


def rep_shift_tiny(s0, c0, d):
    s, c = C.c_double(), C.c_double()
    lib().orc_rep_shift_tiny(C.c_double(s0), C.c_double(c0), C.c_double(d), C.byref(s), C.byref(c))
    return s.value, c.value


def anchor_period(w, dt, cmax):
    return int(lib().orc_anchor_period(C.c_double(w), C.c_double(dt), C.c_double(cmax)))


def rep_shift(s0, c0, d):
    s, c = C.c_double(), C.c_double()
    lib().orc_rep_shift(C.c_double(s0), C.c_double(c0), C.c_double(d), C.byref(s), C.byref(c))
    return s.value, c.value


def init_states(med, ns, nd, dir_lo=0.0, dir_hi=np.pi, src_begin=0, src_end=None, trig=0):
    src_end = ns if src_end is None else src_end
    num = (src_end - src_begin) * nd
    m = np.zeros((num, 9), dtype=np.float64, order="F")
    mm = _medium(med, trig)
    lib().orc_init_states(C.byref(mm), C.c_int(ns), C.c_int(nd), C.c_double(dir_lo),
                          C.c_double(dir_hi), C.c_int(src_begin), C.c_int(src_end), _dp(m),
                          C.c_int64(num))
    return m


def shoot(med, m, dt, obstacle_r0=None, time_unit=0, max_steps=100000, nthreads=1, trig=0, arith=None):
    """In-place on m (Fortran (num,9)); returns (status, nsteps, nreflect)."""
    assert m.flags.f_contiguous and m.dtype == np.float64 and m.shape[1] == 9
    num = m.shape[0]
    st = np.zeros(num, dtype=np.int32)
    ns_ = np.zeros(num, dtype=np.int32)
    nr = np.zeros(num, dtype=np.int32)
    mm = _medium(med, trig, arith)
    ob = _obstacle(obstacle_r0)
    lib().orc_shoot(C.byref(mm), C.byref(ob), C.c_double(dt), C.c_int(time_unit),
                    C.c_int(max_steps), _dp(m), C.c_int64(num), C.c_int64(num), _dp(st), _dp(ns_),
                    _dp(nr), C.c_int(nthreads))
    return st, ns_, nr


def scatter_relation(med, ns, nd, dt, dir_lo=0.0, dir_hi=np.pi, obstacle_r0=None, time_unit=0,
                     src_begin=0, src_end=None, max_steps=100000, nthreads=1, trig=0, arith=None):
    m = init_states(med, ns, nd, dir_lo, dir_hi, src_begin, src_end, trig)
    st, nsteps, nref = shoot(med, m, dt, obstacle_r0, time_unit, max_steps, nthreads, trig, arith)
    return m, st, nsteps, nref


def trace_fixed_time(tab, m_in, dt, mode, nthreads=1):
    assert m_in.flags.f_contiguous and m_in.shape[1] == 9
    num = m_in.shape[0]
    X = np.zeros((num, 4), dtype=np.float64, order="F")
    st = np.zeros(num, dtype=np.int32)
    ns_ = np.zeros(num, dtype=np.int32)
    lib().orc_trace_fixed_time(_dp(tab.eval), _dp(tab.grad), _dp(tab.p), C.c_int(tab.N),
                               C.c_double(dt), C.c_int(mode), C.c_int(tab.arith), _dp(m_in), C.c_int64(num),
                               C.c_int64(num), _dp(X), C.c_int64(num), _dp(st), _dp(ns_),
                               C.c_int(nthreads))
    return X, st, ns_


def forward_operator(tab, m_in, dt, dense=True, nthreads=1):
    """Complete ScatterForwardOperator (NonObstacle.jl:189-243): returns (X (num,4), rho (num,4,4),
    M_dense ((4num, N^2), Fortran) or None, status, nsteps)."""
    assert m_in.flags.f_contiguous and m_in.shape[1] == 9
    num, N = m_in.shape[0], tab.N
    X = np.zeros((num, 4), dtype=np.float64, order="F")
    rho = np.zeros((num, 4, 4), dtype=np.float64)
    M = np.zeros((4 * num, N * N), dtype=np.float64, order="F") if dense else None
    st = np.zeros(num, dtype=np.int32)
    ns_ = np.zeros(num, dtype=np.int32)
    f = lib().orc_forward_operator
    f.restype = None
    f(_dp(tab.eval), _dp(tab.grad), _dp(tab.hess()), _dp(tab.p), C.c_int(N), C.c_double(dt), C.c_int(tab.arith),
      _dp(m_in), C.c_int64(num), C.c_int64(num), _dp(X), C.c_int64(num), _dp(rho),
      _dp(M) if dense else None, _dp(st), _dp(ns_), C.c_int(nthreads))
    return X, rho, M, st, ns_


def misfit(m, obs, thresh=0.2):
    num = m.shape[0]
    mism = np.zeros(4 * num, dtype=np.float64)
    out = np.zeros(2, dtype=np.float64)
    mask = np.zeros(num, dtype=np.uint8)
    lib().orc_misfit(_dp(m), _dp(obs), C.c_int64(num), C.c_int64(num), _dp(mism), _dp(out),
                     _dp(mask), C.c_double(thresh))
    return mism, out, mask


# ---- NEXT-2: the reconstruction loop's passes over M (NonObstacle.jl:340-367), numpy restatement --------
def loop_dofs(M, fidelity):
    """dofs[j] = nnz(M[4j-3,:]) - sum(fidelity[find(M[4j-3,:])])  (NonObstacle.jl:340-342).
    M: scipy CSR (4num, N^2) with explicit zeros already dropped."""
    M = M.tocsr()
    out = np.zeros(M.shape[0] // 4)
    for j in range(out.size):
        cols = M.indices[M.indptr[4 * j]:M.indptr[4 * j + 1]]
        out[j] = cols.size - fidelity[cols].sum()
    return out


def loop_order(dofs, rank_thres):
    """perm / trunc / order (NonObstacle.jl:345-349) -> (ray mask, 0-based row list in the reference's order)."""
    perm = np.argsort(dofs, kind="stable")
    trunc = np.searchsorted(dofs[perm], rank_thres, side="right")
    rays = perm[:trunc]
    rows = (4 * rays[:, None] + np.arange(4)[None, :]).reshape(-1)
    mask = np.zeros(dofs.size, dtype=np.uint8)
    mask[rays] = 1
    return mask, rows


def loop_gradient(M, mismatch, rows):
    """b = M[order,:]' * mismatch[order] over all N^2 columns (NonObstacle.jl:351,354 take [Idx])."""
    return M.tocsr()[rows, :].T @ mismatch[rows]


def loop_fidelity(M, correction, mismatch, rows, fidelity, rejection, decay):
    """residual and fidelity update (NonObstacle.jl:359-367), sequential like the reference."""
    M = M.tocsr()
    fid = fidelity.copy()
    residual = np.abs(M[rows, :] @ correction - mismatch[rows])
    for k, row in enumerate(rows):
        if residual[k] < rejection:
            ids = M.indices[M.indptr[row]:M.indptr[row + 1]]
            fid[ids] = np.maximum(fid[ids], 1 - decay * residual[k])
    return fid, residual


# ---- NEXT-3: ray classification (Utility.jl:1-39; ObstacleDemo.jl:48-67), pure-Python restatement -------
def similarity(X, Y):                                            # Utility.jl:1-3
    import math
    return math.cos(math.atan2(X[0], X[1]) - math.atan2(Y[0], Y[1]))


def alignment(signal):                                           # Utility.jl:5-13 (in place on a list)
    import math
    n = len(signal)
    for i in range(1, n):
        d = signal[i] - signal[i - 1]
        if abs(d) > 1.5 * math.pi:
            sh = (1.0 if d > 0 else -1.0 if d < 0 else 0.0) * 2 * math.pi
            for k in range(i, n):
                signal[k] = signal[k] - sh
    return signal


def derivative_check(signal):                                    # Utility.jl:15-39, 1-based lo / hi
    n = len(signal)
    deriv = [signal[i + 1] - signal[i] for i in range(n - 1)]    # deriv[i-1] = Julia deriv[i]
    lo, hi = 1, n
    for i in range(1, n - 1):
        if abs(deriv[i - 1] - deriv[i]) > 0.12:
            lo = i
            break
    for i in range(n - 1, 1, -1):
        if abs(deriv[i - 1] - deriv[i - 2]) > 0.12:
            hi = i + 1
            break
    if abs(lo - hi) < n / 8:
        lo, hi = 1, n
    return lo, hi


def classify(m, ns, nd):
    """ObstacleDemo.jl:48-67: (ortho index list 0-based, lo[ns], hi[ns] 1-based, aligned arg (ns, nd))."""
    import math
    ortho = [r for r in range(ns * nd)
             if similarity(m[r, 0:2], m[r, 4:6]) > 0.995 and similarity(m[r, 2:4], m[r, 6:8]) < -0.995]
    lo, hi, args = [], [], []
    for s in range(ns):
        arg = [math.atan2(m[s * nd + j, 5], m[s * nd + j, 4]) for j in range(nd)]
        arg = alignment(arg)
        l, h = derivative_check(arg)
        lo.append(l); hi.append(h); args.append(arg)
    return np.array(ortho, dtype=np.int64), np.array(lo), np.array(hi), np.array(args)
