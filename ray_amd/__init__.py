"""ray_amd -- MI355X-native forward ray-shooting core of lowrank/ray behind the reference's names.

Host-side mirror of the reference's Julia entry points (Julia is not installed in this image;
julia/RayAMD.jl is the ccall shim a Julia caller would use, this module is the same thing over
ctypes).  Everything here calls libray_amd.so (include/ray_amd.h); nothing is computed on the
CPU and there is no fallback.

    reference (file:line)                          here
    ScatterRelation          NonObstacle.jl:54     ScatterRelation(c, gradc, ns, nd, dt, dir)
    ScatterRelationObstacle  Obstacle.jl:11        ScatterRelationObstacle(c, gradc, ob, gradob, ns, nd, dt)
    ScatterForwardOperator   NonObstacle.jl:189    ScatterForwardOperator(c, m, ext, dt) -> (M, s)
    NonReflectionTrace       Obstacle.jl:275       NonReflectionTrace(c, m, ext, dt)
    mismatch/objective/perc  NonObstacle.jl:336,372; ObstacleDemo.jl:219-226   misfit(m, obs)

Function-valued arguments cannot reach a GPU through a C ABI, so `c`/`gradc` are a medium
descriptor (SinProd) and `ob`/`gradob` an obstacle descriptor (Circle, Trefoil); pass the same object (or
None) for the gradient slot.  Arrays are numpy float64 in Fortran order, laid out like the
reference's m / s / c.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import (RayError, ST_EXITED, ST_STEP_CAP, ST_BISECT_CAP, ST_OOB, ST_CEASED,
                   TIME_SECONDS, TIME_STEPS, TRACE_SHORTEN_CEASE, TRACE_OVERSHOOT,
                   ARITH_EXACT, ARITH_FUSED)

__all__ = ["Context", "SinProd", "Circle", "GridMedium", "ScatterRelation",
           "ScatterRelationObstacle", "ScatterForwardOperator", "NonReflectionTrace", "misfit",
           "RayError", "default_context"]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One per GPU (ray_ctx)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.ray_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ray_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _lib.check(self._lib.ray_ctx_synchronize(self.handle))

    def release_workspaces(self):
        """Free what the context keeps between ScatterForwardOperator calls (workspaces, parked CSR arrays)."""
        _lib.check(self._lib.ray_ctx_release_workspaces(self.handle))

    def set_stream(self, hip_stream):
        """Borrow the caller's hipStream_t (an integer handle, e.g. torch.cuda.current_stream().cuda_stream);
        0 / None is the legacy default stream itself -- torch's default stream -- not the context's own."""
        _lib.check(self._lib.ray_ctx_set_stream(self.handle, _lib.STREAM_BORROWED, C.c_void_p(hip_stream or 0)))

    def use_own_stream(self):
        """Back to the context's own non-blocking stream."""
        _lib.check(self._lib.ray_ctx_set_stream(self.handle, _lib.STREAM_OWN, None))

    def set_arith(self, arith):
        """ARITH_EXACT (un-fused, bit-identical to the as-written CPU restatement) or
        ARITH_FUSED (default: multiply-adds contracted)."""
        _lib.check(self._lib.ray_ctx_set_arith(self.handle, int(arith)))

    def set_option(self, option, value):
        _lib.check(self._lib.ray_ctx_set_option(self.handle, int(option), int(value)))
        if int(option) == _lib.OPT_SOURCE_STRIDE:
            self.source_stride = int(value)
        if int(option) == _lib.OPT_FAN_DIRECTIONS:
            self.fan_directions = int(value)

    def sources_of_call(self, src_begin, src_end):
        """Number of sources a shoot of [src_begin, src_end) takes under the context's source stride."""
        st = getattr(self, "source_stride", 1)
        return (src_end - src_begin + st - 1) // st

    def enable_timing(self, on=True):
        _lib.check(self._lib.ray_ctx_enable_timing(self.handle, int(bool(on))))

    def counters(self):
        c = _lib.Counters()
        _lib.check(self._lib.ray_get_counters(self.handle, C.byref(c)))
        return {"rays": c.rays, "ray_steps": c.ray_steps, "launches": c.launches,
                "kernel_ms": c.kernel_ms}

    def reset_counters(self):
        _lib.check(self._lib.ray_reset_counters(self.handle))


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def host_register(a):
    """Page-lock a numpy array the caller keeps across calls (ray_host_register); host_unregister(a) undoes it."""
    _lib.check(_lib.load().ray_host_register(_ptr(a), a.nbytes))


def host_unregister(a):
    _lib.check(_lib.load().ray_host_unregister(_ptr(a)))


def runtime_info():
    """Which HIP runtime and RCCL the library's calls are bound to in this process (ray_runtime_info)."""
    buf = C.create_string_buffer(1024)
    _lib.check(_lib.load().ray_runtime_info(buf, len(buf)))
    return buf.value.decode()


def device_count():
    n = C.c_int(0)
    _lib.load().ray_device_count(C.byref(n))
    return n.value


class SinProd:
    """c = 1 + A sin(wx x) sin(wy y); grad c = [gx cos(wx x) sin(wy y), gy sin(wx x) cos(wy y)]
    -- the family both shipped demos use (NonObstacleDemo.jl:12,23; ObstacleDemo.jl:12,18)."""

    def __init__(self, A, wx, wy, gx, gy):
        self.params = (float(A), float(wx), float(wy), float(gx), float(gy))

    @classmethod
    def smooth_demo(cls):          # NonObstacleDemo.jl:12,23
        return cls(0.3, 1.5 * math.pi, 1.5 * math.pi, 0.45 * math.pi, 0.45 * math.pi)

    @classmethod
    def obstacle_demo(cls):        # ObstacleDemo.jl:12,18
        return cls(0.2, 2 * math.pi, math.pi, 2 * (0.2 * math.pi), 0.2 * math.pi)

    def __call__(self, x, y):      # waveSpeed(x, y), for filling grids on the host
        A, wx, wy, _, _ = self.params
        return 1 + A * math.sin(wx * x) * math.sin(wy * y)

    def _medium(self, ctx):
        cache = self.__dict__.setdefault("_handles", {})
        if id(ctx) not in cache:
            cache[id(ctx)] = _Medium.sinprod(ctx, self.params)
        return cache[id(ctx)]


class RadialSineSum:
    """c = 1 + sum_k a_k sin(pi |x - x_k|), grad c = sum_k a_k pi cos(pi r_k)/r_k (x - x_k): the commented
    alternate media of the demos (NonObstacleDemo.jl:7-11,15-22; ObstacleDemo.jl:9-11,15-17).  terms: up to
    four (a_k, x_k, y_k)."""

    def __init__(self, terms):
        self.terms = [(float(a), float(x), float(y)) for a, x, y in terms]

    @classmethod
    def smooth_alt(cls):           # NonObstacleDemo.jl:7-11
        return cls([(0.2, 0.5, 0.2), (0.4, -0.4, -0.3), (0.2, 0.3, -0.4), (0.3, -0.2, 0.1)])

    @classmethod
    def obstacle_alt(cls):         # ObstacleDemo.jl:9-11
        return cls([(0.4, 0.5, 0.2), (0.4, -0.4, -0.3)])

    def __call__(self, x, y):      # waveSpeed(x, y), for filling grids on the host
        v = 1.0
        for a, xk, yk in self.terms:
            v = v + a * math.sin(math.pi * math.sqrt((x - xk) * (x - xk) + (y - yk) * (y - yk)))
        return v

    def _medium(self, ctx):
        cache = self.__dict__.setdefault("_handles", {})
        if id(ctx) not in cache:
            cache[id(ctx)] = _Medium.radial(ctx, self.terms)
        return cache[id(ctx)]


class Circle:
    """ob = sqrt(x^2+y^2) - r0, normal (x,y)/r (ObstacleDemo.jl:20-34)."""

    def __init__(self, r0):
        self.r0 = float(r0)

    def _struct(self):
        return _lib.Obstacle(_lib.OB_CIRCLE, 0, self.r0, (C.c_double * 2)(0, 0))


class Trefoil:
    """ob = r - (R - rho cos 3 theta); normal [x,y]/r - 3 rho_n sin(3 theta)/r [-y,x]/r, renormalised: the
    commented alternate obstacle of ObstacleDemo.jl:23-24 (R = 0.4, rho = 0.3) and :30-31 (rho_n = 0.2)
    -- the reference's normal has a constant of its own, kept as a separate parameter."""

    def __init__(self, R=0.4, rho=0.3, rho_n=0.2):
        self.R, self.rho, self.rho_n = float(R), float(rho), float(rho_n)

    @classmethod
    def demo(cls):                 # ObstacleDemo.jl:23-24,30-31
        return cls(0.4, 0.3, 0.2)

    def _struct(self):
        return _lib.Obstacle(_lib.OB_TREFOIL, 0, self.R, (C.c_double * 2)(self.rho, self.rho_n))


class _Medium:
    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    @classmethod
    def sinprod(cls, ctx, params):
        h = C.c_void_p()
        _lib.check(ctx._lib.ray_medium_create_sinprod(ctx.handle, *params, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def radial(cls, ctx, terms):
        h = C.c_void_p()
        K = len(terms)
        a = (C.c_double * max(K, 1))(*[t[0] for t in terms])
        x = (C.c_double * max(K, 1))(*[t[1] for t in terms])
        y = (C.c_double * max(K, 1))(*[t[2] for t in terms])
        _lib.check(ctx._lib.ray_medium_create_radial(ctx.handle, K, a, x, y, C.byref(h)))
        return cls(ctx, h)

    def close(self):
        if self.handle:                       # safe after the context's own close: the library detached it
            self.ctx._lib.ray_medium_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GridMedium(_Medium):
    """Nodal wavespeed c (N,N) on linspace(-ext,ext,N)^2, first index x (NonObstacle.jl:306);
    owns the eval/grad tables of NonObstacle.jl:191-210 in HBM."""

    def __init__(self, c, ext, ctx=None):
        ctx = ctx or default_context()
        c = np.asfortranarray(c, dtype=np.float64)
        if c.ndim != 2 or c.shape[0] != c.shape[1]:
            raise ValueError("c must be square (N,N)")
        h = C.c_void_p()
        _lib.check(ctx._lib.ray_medium_create_grid(ctx.handle, _ptr(c), c.shape[0], float(ext),
                                                   C.byref(h)))
        super().__init__(ctx, h)
        self.N, self.ext = c.shape[0], float(ext)

    def update(self, c):
        c = np.asfortranarray(c, dtype=np.float64)
        if c.shape != (self.N, self.N):
            raise ValueError("shape mismatch")
        _lib.check(self.ctx._lib.ray_medium_update_grid(self.handle, _ptr(c)))

    def tables(self):
        """(eval (N-1,N-1,4), grad (N-2,N-2,8), p (N)) in the reference's layout."""
        N = self.N
        ev = np.zeros((N - 1, N - 1, 4), order="F")
        gr = np.zeros((N - 2, N - 2, 8), order="F")
        p = np.zeros(N)
        _lib.check(self.ctx._lib.ray_medium_export_tables(self.handle, _ptr(ev), _ptr(gr), _ptr(p)))
        return ev, gr, p

    def _medium(self, ctx):
        if ctx is not self.ctx:
            raise RayError("grid medium belongs to another context")
        return self


class Group:
    """All GPUs of the node behind one caller (ray_group): what the reference's callers get from addprocs(3) +
    @parallel + remotecall_wait (NonObstacleDemo.jl:1; NonObstacle.jl:59,236-240).  devices: list of device
    indices (default: every visible GPU); one device listed n times is the one-GPU rehearsal."""

    def __init__(self, devices=None):
        self._lib = _lib.load()
        if devices is None:
            devices = list(range(device_count()))
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _lib.check(self._lib.ray_group_create(arr, len(devices), C.byref(h)))
        self.handle, self.devices = h, list(devices)
        n, r = C.c_int(), C.c_int()
        _lib.check(self._lib.ray_group_size(h, C.byref(n), C.byref(r)))
        self.size, self.uses_rccl = n.value, bool(r.value)
        self._media = []

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ray_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_arith(self, arith):
        _lib.check(self._lib.ray_group_set_arith(self.handle, int(arith)))

    def grid_medium(self, c, ext):
        c = np.asfortranarray(c, dtype=np.float64)
        h = C.c_void_p()
        _lib.check(self._lib.ray_group_medium_create_grid(self.handle, _ptr(c), c.shape[0], float(ext), C.byref(h)))
        return _GroupMedium(self, h)

    def sinprod_medium(self, sp):
        h = C.c_void_p()
        _lib.check(self._lib.ray_group_medium_create_sinprod(self.handle, *sp.params, C.byref(h)))
        return _GroupMedium(self, h)

    def ScatterRelation(self, med, ns, nd, dt, dir=(0.0, math.pi), *, ob=None, time_unit=TIME_SECONDS, src_begin=0,
                        src_end=None, max_steps=0, info=False, keep_on_device=False):
        """ScatterRelation / ScatterRelationObstacle over the group: member k of G shoots the comb of sources k, k + G, ..."""
        src_end = ns if src_end is None else src_end
        num = (src_end - src_begin) * nd
        m = None if keep_on_device else np.zeros((num, 9), dtype=np.float64, order="F")
        st, nst, nrf = (np.zeros(num, dtype=np.int32) for _ in range(3))
        prm = _lib.ShootParams(ns, nd, src_begin, src_end, float(dir[0]), float(dir[1]), float(dt), time_unit, max_steps)
        obs = ob._struct() if ob is not None else None
        _lib.check(self._lib.ray_group_shoot_to_exit(self.handle, med.handle, C.byref(obs) if obs else None, C.byref(prm),
                                                     _ptr(m), max(num, 1), _ptr(st), _ptr(nst), _ptr(nrf)))
        if info:
            return m, {"status": st, "nsteps": nst, "nreflect": nrf}
        return m

    def trace(self, med, m, dt, mode=TRACE_OVERSHOOT, *, nd=0, info=False):
        """The fixed-time traces over the group (ray_group_trace_fixed_time): NonReflectionTrace's end states
        (mode TRACE_OVERSHOOT, Obstacle.jl:275-329) or ScatterForwardOperator's trajectory ends (TRACE_SHORTEN_CEASE,
        NonObstacle.jl:141-187) for every row of m; the members keep their shares for set_observation + misfit."""
        m = np.asfortranarray(m, dtype=np.float64)
        num = m.shape[0]
        X = np.zeros((num, 4), dtype=np.float64, order="F")
        st, nst = (np.zeros(num, dtype=np.int32) for _ in range(2))
        _lib.check(self._lib.ray_group_trace_fixed_time(self.handle, med.handle, _ptr(m), num, max(num, 1), int(nd),
                                                        float(dt), int(mode), _ptr(X), max(num, 1), _ptr(st), _ptr(nst)))
        return (X, {"status": st, "nsteps": nst}) if info else X

    def ScatterForwardOperator(self, med, m, dt, *, nd=0, droptol=1e-12, info=False):
        """ScatterForwardOperator (NonObstacle.jl:189-243) over the group: returns (M, s) with M a GroupFrechetMatrix
        (one device-resident part per member; `.tocsr()` is the whole matrix in m's row order) and s = m with columns
        5..8 replaced by the traced states (:196, :185)."""
        m = np.asfortranarray(m, dtype=np.float64)
        num = m.shape[0]
        X = np.zeros((num, 4), dtype=np.float64, order="F")
        st, nst = (np.zeros(num, dtype=np.int32) for _ in range(2))
        h = C.c_void_p()
        _lib.check(self._lib.ray_group_forward_operator(self.handle, med.handle, _ptr(m), num, max(num, 1), int(nd), float(dt),
                                                        float(droptol), _ptr(X), max(num, 1), _ptr(st), _ptr(nst), C.byref(h)))
        s = np.array(m, order="F", copy=True)
        s[:, 4:8] = X
        M = GroupFrechetMatrix(self, h)
        return (M, s, {"status": st, "nsteps": nst}) if info else (M, s)

    def set_observation(self, obs):
        obs = np.asfortranarray(obs, dtype=np.float64)
        _lib.check(self._lib.ray_group_set_observation(self.handle, _ptr(obs), obs.shape[0], max(obs.shape[0], 1)))
        self._num = obs.shape[0]

    def misfit(self, perc_thresh=0.2):
        num = self._num
        out = np.zeros(2)
        mask = np.zeros(num, dtype=np.uint8)
        mism = np.zeros(4 * num)
        _lib.check(self._lib.ray_group_misfit(self.handle, _ptr(out), _ptr(mask), float(perc_thresh), _ptr(mism)))
        return {"mismatch": mism, "sum_mismatch2": out[0], "sum_target2": out[1],
                "objective": math.sqrt(out[0] / out[1]) if out[1] > 0 else float("nan"), "mask": mask}

    def counters(self):
        c = _lib.Counters()
        _lib.check(self._lib.ray_group_get_counters(self.handle, C.byref(c)))
        return {"rays": c.rays, "ray_steps": c.ray_steps, "launches": c.launches, "kernel_ms": c.kernel_ms}


class _GroupMedium:
    def __init__(self, group, handle):
        self.group, self.handle = group, handle

    def update(self, c):
        c = np.asfortranarray(c, dtype=np.float64)
        _lib.check(self.group._lib.ray_group_medium_update_grid(self.handle, _ptr(c)))

    def close(self):
        if self.handle:
            self.group._lib.ray_group_medium_destroy(self.handle)
            self.handle = None


def _resolve(c, gradc, ctx):
    if gradc is not None and gradc is not c:
        raise RayError("pass the medium descriptor itself (or None) as its gradient: closures "
                       "cannot cross the C ABI onto the GPU")
    if not hasattr(c, "_medium"):
        raise RayError("c must be a SinProd, RadialSineSum or GridMedium descriptor, not a Python function")
    return c._medium(ctx)


def _shoot(med, ob, ns, nd, dt, dir_, time_unit, ctx, src_begin, src_end, max_steps, info):
    src_end = ns if src_end is None else src_end
    num = (ctx or default_context()).sources_of_call(src_begin, src_end) * nd
    m = np.zeros((num, 9), dtype=np.float64, order="F")
    st = np.zeros(num, dtype=np.int32)
    nst = np.zeros(num, dtype=np.int32)
    nrf = np.zeros(num, dtype=np.int32)
    prm = _lib.ShootParams(ns, nd, src_begin, src_end, float(dir_[0]), float(dir_[1]), float(dt),
                           time_unit, max_steps)
    obs = ob._struct() if ob is not None else None
    _lib.check(ctx._lib.ray_shoot_to_exit(ctx.handle, med.handle, C.byref(obs) if obs else None,
                                          C.byref(prm), _ptr(m), max(num, 1), _ptr(st), _ptr(nst),
                                          _ptr(nrf)))
    if info:
        return m, {"status": st, "nsteps": nst, "nreflect": nrf}
    return m


def ScatterRelation(c, gradc, ns, nd, dt, dir=(0.0, math.pi), *, ctx=None, src_begin=0,
                    src_end=None, max_steps=0, info=False):
    """NonObstacle.jl:54-99.  Returns m (ns*nd, 9), t in seconds."""
    ctx = ctx or default_context()
    return _shoot(_resolve(c, gradc, ctx), None, ns, nd, dt, dir, TIME_SECONDS, ctx, src_begin,
                  src_end, max_steps, info)


def ScatterRelationObstacle(c, gradc, ob, gradob, ns, nd, dt, *, ctx=None, src_begin=0,
                            src_end=None, max_steps=0, info=False):
    """Obstacle.jl:11-97.  Column 9 counts steps (the caller rescales, ObstacleDemo.jl:122)."""
    ctx = ctx or default_context()
    if gradob is not None and gradob is not ob:
        raise RayError("pass the obstacle descriptor itself (or None) as its normal")
    return _shoot(_resolve(c, gradc, ctx), ob, ns, nd, dt, (0.0, math.pi), TIME_STEPS, ctx,
                  src_begin, src_end, max_steps, info)


def shoot_states(c, m, dt, ob=None, time_unit=TIME_SECONDS, *, ctx=None, max_steps=0):
    """Per-ray loops of ScatterRelation / ScatterRelationObstacle from explicit m[:,1:4]
    (NonObstacle.jl:67-97; Obstacle.jl:23-95).  In place; returns info dict."""
    ctx = ctx or default_context()
    med = _resolve(c, None, ctx)
    if not (m.flags.f_contiguous and m.dtype == np.float64 and m.ndim == 2 and m.shape[1] == 9):
        raise ValueError("m must be float64 Fortran-ordered (num, 9)")
    num = m.shape[0]
    st = np.zeros(num, dtype=np.int32)
    nst = np.zeros(num, dtype=np.int32)
    nrf = np.zeros(num, dtype=np.int32)
    obs = ob._struct() if ob is not None else None
    _lib.check(ctx._lib.ray_shoot_states(ctx.handle, med.handle, C.byref(obs) if obs else None,
                                         float(dt), time_unit, max_steps, _ptr(m), num, max(num, 1),
                                         _ptr(st), _ptr(nst), _ptr(nrf)))
    return {"status": st, "nsteps": nst, "nreflect": nrf}


def shoot_paths(c, m, dt, ob=None, time_unit=TIME_SECONDS, *, ctx=None, max_steps=0, max_pts=4096):
    """What ScatterRelationObstaclePlot collects before it plots (Obstacle.jl:98-221): shoots the rays
    of m[:,1:4] (in place, like shoot_states) and returns (paths, info): paths[k] is the (npts, 2)
    polyline of ray k -- start, every step, reflection points, exit point."""
    ctx = ctx or default_context()
    med = _resolve(c, None, ctx)
    if not (m.flags.f_contiguous and m.dtype == np.float64 and m.ndim == 2 and m.shape[1] == 9):
        raise ValueError("m must be float64 Fortran-ordered (num, 9)")
    num = m.shape[0]
    st = np.zeros(num, dtype=np.int32)
    nst = np.zeros(num, dtype=np.int32)
    nrf = np.zeros(num, dtype=np.int32)
    npts = np.zeros(num, dtype=np.int32)
    path = np.zeros((num, max_pts, 2), dtype=np.float64)
    obs = ob._struct() if ob is not None else None
    _lib.check(ctx._lib.ray_shoot_paths(ctx.handle, med.handle, C.byref(obs) if obs else None, float(dt),
                                        time_unit, max_steps, _ptr(m), num, max(num, 1), max_pts,
                                        _ptr(path), _ptr(npts), _ptr(st), _ptr(nst), _ptr(nrf)))
    if num and npts.max() > max_pts:
        raise RayError(f"a ray produced {int(npts.max())} points; raise max_pts (= {max_pts})")
    return [path[k, :npts[k]].copy() for k in range(num)], {"status": st, "nsteps": nst, "nreflect": nrf, "npts": npts}


def ScatterRelationObstaclePlot(c, gradc, ob, gradob, ns, nd, dt, rayIndex, *, ctx=None):
    """Obstacle.jl:98-221 without PyPlot: returns (m, paths) -- m as the reference fills it (rows of
    rayIndex [0-based here] shot, the others carrying their initial state only) and the polylines of
    the selected rays, ready for `plot(p[:,0], p[:,1])`."""
    ctx = ctx or default_context()
    m = ScatterRelationObstacle(c, gradc, ob, gradob, ns, nd, dt, ctx=ctx)
    m[:, 4:] = 0.0
    idx = np.asarray(rayIndex, dtype=np.int64)
    sel = np.asfortranarray(m[idx])
    paths, _ = shoot_paths(c, sel, dt, ob, TIME_STEPS, ctx=ctx)
    m[idx] = sel
    return m, paths


def NonReflectionPlot(c, m, ext, dt, *, ctx=None, max_pts=4096, info=False):
    """Obstacle.jl:222-274 without PyPlot: traces every row of m from m[:,1:4] for the time m[:,9] (the
    overshooting loop of NonReflectionTrace) and returns the polylines `res` -- start and the position
    after every step -- ready for `plot(p[:,0], p[:,1])`."""
    ctx = ctx or default_context()
    med = c if isinstance(c, GridMedium) else GridMedium(c, ext, ctx)
    m = np.asfortranarray(m, dtype=np.float64)
    num = m.shape[0]
    X = np.zeros((num, 4), order="F")
    st = np.zeros(num, dtype=np.int32)
    nst = np.zeros(num, dtype=np.int32)
    npts = np.zeros(num, dtype=np.int32)
    path = np.zeros((num, max_pts, 2), dtype=np.float64)
    _lib.check(ctx._lib.ray_trace_paths(ctx.handle, med.handle, _ptr(m), num, max(num, 1), float(dt),
                                        TRACE_OVERSHOOT, max_pts, _ptr(path), _ptr(npts), _ptr(X), max(num, 1),
                                        _ptr(st), _ptr(nst)))
    if num and npts.max() > max_pts:
        raise RayError(f"a ray produced {int(npts.max())} points; raise max_pts (= {max_pts})")
    paths = [path[k, :npts[k]].copy() for k in range(num)]
    if info:
        return paths, {"X": X, "status": st, "nsteps": nst, "npts": npts}
    return paths


def _trace(c, m, ext, dt, mode, ctx, info, nd=0):
    """`nd` > 0 tells the library the rays are fans (row = source * nd + direction), so that the trace
    kernel packs neighbouring directions into one wave as the shooter does (RAY_OPT_FAN_DIRECTIONS);
    the results do not depend on it."""
    ctx = ctx or default_context()
    if isinstance(c, SinProd):
        raise RayError("fixed-time tracing is defined on gridded media only (NonObstacle.jl:189, "
                       "Obstacle.jl:275 take c::Matrix)")
    med = c if isinstance(c, GridMedium) else GridMedium(c, ext, ctx)
    m = np.asfortranarray(m, dtype=np.float64)
    num = m.shape[0]
    X = np.zeros((num, 4), dtype=np.float64, order="F")
    st = np.zeros(num, dtype=np.int32)
    nst = np.zeros(num, dtype=np.int32)
    before = getattr(ctx, "fan_directions", 0)         # a hint the caller set on the context stays in force
    if nd:
        ctx.set_option(_lib.OPT_FAN_DIRECTIONS, int(nd))
    try:
        _lib.check(ctx._lib.ray_trace_fixed_time(ctx.handle, med.handle, _ptr(m), num, max(num, 1),
                                                 float(dt), mode, _ptr(X), max(num, 1), _ptr(st),
                                                 _ptr(nst)))
    finally:
        if nd:
            ctx.set_option(_lib.OPT_FAN_DIRECTIONS, before)
    return X, {"status": st, "nsteps": nst}


class GroupFrechetMatrix:
    """The Frechet matrix of Group.ScatterForwardOperator (ray_group_frechet): member k holds the rows of its rays as an
    ordinary FrechetMatrix (`members()`), `.tocsr()` assembles the whole matrix on the host in the caller's row order."""

    def __init__(self, group, handle):
        self.group, self.handle = group, handle
        r, c, n = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(group._lib.ray_group_frechet_shape(handle, C.byref(r), C.byref(c), C.byref(n)))
        self.shape, self.nnz = (r.value, c.value), n.value

    def tocsr(self):
        import scipy.sparse as sps
        indptr = np.zeros(self.shape[0] + 1, dtype=np.int64)
        indices = np.zeros(self.nnz, dtype=np.int32)
        data = np.zeros(self.nnz, dtype=np.float64)
        _lib.check(self.group._lib.ray_group_frechet_export_csr(self.handle, _ptr(indptr), _ptr(indices), _ptr(data)))
        return sps.csr_matrix((data, indices, indptr), shape=self.shape)

    def _reduce(self, fn, vec, n, ray_mask):
        vec = np.ascontiguousarray(vec, dtype=np.float64).reshape(-1)
        if vec.size != n:
            raise ValueError(f"expected a vector of length {n}, got {vec.size}")
        mk = None
        if ray_mask is not None:
            mk = np.ascontiguousarray(ray_mask, dtype=np.uint8).reshape(-1)
            if mk.size != self.shape[0] // 4:
                raise ValueError("ray_mask must have one entry per ray")
        out = np.zeros(self.shape[1])
        _lib.check(fn(self.handle, _ptr(vec), _ptr(mk) if mk is not None else None, _ptr(out)))
        return out

    def matvec_t(self, v, ray_mask=None):
        """b = M[order,:]' v[order] summed over the members with one all-reduce (NonObstacle.jl:354)."""
        return self._reduce(self.group._lib.ray_group_frechet_matvec_t, v, self.shape[0], ray_mask)

    def normal_matvec(self, x, ray_mask=None):
        """w = M[order,:]' (M[order,:] x) summed over the members with one all-reduce (the action of :353's MtM)."""
        return self._reduce(self.group._lib.ray_group_frechet_normal_matvec, x, self.shape[1], ray_mask)

    def members(self):
        """[(FrechetMatrix, rays)]: member k's part (not owned: it lives as long as this object) and the caller's ray
        indices its local rays 0, 1, ... stand for."""
        out = []
        for k in range(self.group.size):
            f, row0, cnt, nd, stride = C.c_void_p(), C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64()
            _lib.check(self.group._lib.ray_group_frechet_member(self.handle, k, C.byref(f), C.byref(row0), C.byref(cnt),
                                                                C.byref(nd), C.byref(stride)))
            rays = (row0.value + stride.value * np.arange(cnt.value)[:, None] + np.arange(nd.value)[None, :]).reshape(-1)
            out.append((FrechetMatrix(self.group, f, owned=False), rays))
        return out

    def close(self):
        if self.handle:
            self.group._lib.ray_group_frechet_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrechetMatrix:
    """Device-resident CSR form of the Frechet matrix M (ray_frechet handle): rows 4*ray + component,
    columns i + j*N.  `.tocsr()` copies it to a scipy.sparse.csr_matrix."""

    def __init__(self, ctx, handle, owned=True):
        self.ctx, self.handle, self.owned = ctx, handle, owned
        r, c, n = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(ctx._lib.ray_frechet_shape(handle, C.byref(r), C.byref(c), C.byref(n)))
        self.shape, self.nnz = (r.value, c.value), n.value

    def tocsr(self):
        import scipy.sparse as sps
        indptr = np.zeros(self.shape[0] + 1, dtype=np.int64)
        indices = np.zeros(self.nnz, dtype=np.int32)
        data = np.zeros(self.nnz, dtype=np.float64)
        _lib.check(self.ctx._lib.ray_frechet_export_csr(self.handle, _ptr(indptr), _ptr(indices), _ptr(data)))
        return sps.csr_matrix((data, indices, indptr), shape=self.shape)

    def device_arrays(self):
        """(indptr, indices, data) device addresses (int64[4num+1], int32[nnz], float64[nnz])."""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self.ctx._lib.ray_frechet_csr_dev(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def block_rows_dev(self):
        """(n_entries, uoff, ucols, uvals) of the device-resident block rows: ray j owns entries uoff[j]..uoff[j+1],
        entry u is node ucols[u] with the values uvals[4u:4u+4] of matrix rows 4j..4j+3 (before droptol)."""
        n, a, b, c = C.c_int64(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self.ctx._lib.ray_frechet_block_rows_dev(self.handle, C.byref(n), C.byref(a), C.byref(b), C.byref(c)))
        return n.value, a.value, b.value, c.value

    # -- the reconstruction loop's passes over M (NonObstacle.jl:340-367), on the device -----------
    def _vec(self, a, n, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype).reshape(-1)
        if a.size != n:
            raise ValueError(f"expected a vector of length {n}, got {a.size}")
        return a

    def dofs(self, fidelity):
        """dofs[j] = nnz(M[4j-3,:]) - sum(fidelity[find(M[4j-3,:])])  (NonObstacle.jl:340-342)."""
        fid = self._vec(fidelity, self.shape[1])
        out = np.zeros(self.shape[0] // 4)
        _lib.check(self.ctx._lib.ray_frechet_dofs(self.handle, _ptr(fid), _ptr(out)))
        return out

    def matvec(self, x):
        x = self._vec(x, self.shape[1])
        y = np.zeros(self.shape[0])
        _lib.check(self.ctx._lib.ray_frechet_matvec(self.handle, _ptr(x), _ptr(y)))
        return y

    def rmatvec(self, v, ray_mask=None):
        """b = M[order,:]' v[order] with `order` = the four rows of every ray with ray_mask != 0
        (NonObstacle.jl:349,354; take b[Idx])."""
        v = self._vec(v, self.shape[0])
        mk = None if ray_mask is None else self._vec(ray_mask, self.shape[0] // 4, np.uint8)
        b = np.zeros(self.shape[1])
        _lib.check(self.ctx._lib.ray_frechet_matvec_t(self.handle, _ptr(v), _ptr(mk) if mk is not None else None, _ptr(b)))
        return b

    def normal_matvec(self, x, ray_mask=None):
        """w = M[order,:]' (M[order,:] x): the action of the reference's MtM (NonObstacle.jl:353) without forming it."""
        x = self._vec(x, self.shape[1])
        mk = None if ray_mask is None else self._vec(ray_mask, self.shape[0] // 4, np.uint8)
        w = np.zeros(self.shape[1])
        _lib.check(self.ctx._lib.ray_frechet_normal_matvec(self.handle, _ptr(x), _ptr(mk) if mk is not None else None, _ptr(w)))
        return w

    def normal_diag(self, ray_mask=None):
        """diag(MtM) over the selected rays."""
        mk = None if ray_mask is None else self._vec(ray_mask, self.shape[0] // 4, np.uint8)
        d = np.zeros(self.shape[1])
        _lib.check(self.ctx._lib.ray_frechet_normal_diag(self.handle, _ptr(mk) if mk is not None else None, _ptr(d)))
        return d

    def fidelity_update(self, correction, mismatch, fidelity, rejection, decay, ray_mask=None):
        """NonObstacle.jl:359-367: returns (fidelity_new, residual); residual is 0 on unselected rows."""
        x = self._vec(correction, self.shape[1])
        mm = self._vec(mismatch, self.shape[0])
        fid = self._vec(fidelity, self.shape[1]).copy()
        mk = None if ray_mask is None else self._vec(ray_mask, self.shape[0] // 4, np.uint8)
        res = np.zeros(self.shape[0])
        _lib.check(self.ctx._lib.ray_frechet_fidelity_update(self.handle, _ptr(x), _ptr(mm),
                                                             _ptr(mk) if mk is not None else None,
                                                             float(rejection), float(decay), _ptr(fid), _ptr(res)))
        return fid, res

    def close(self):
        if self.handle and self.owned:
            self.ctx._lib.ray_frechet_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def forward_operator_dev(c, m, ext, dt, *, ctx=None, droptol=1e-12, want_rho=True):
    """The complete ChunkProcessing! pass (NonObstacle.jl:146-185) on the GPU: returns
    (FrechetMatrix [device block rows], X (num, 4), rho (num, 4, 4), info).  want_rho=False skips the download of
    the Jacobians (128 B per ray; the reconstruction loop never looks at them): rho is None then."""
    ctx = ctx or default_context()
    if isinstance(c, SinProd):
        raise RayError("ScatterForwardOperator is defined on gridded media only (NonObstacle.jl:189)")
    med = c if isinstance(c, GridMedium) else GridMedium(c, ext, ctx)
    m = np.asfortranarray(m, dtype=np.float64)
    num = m.shape[0]
    X = np.zeros((num, 4), dtype=np.float64, order="F")
    rho = np.zeros((num, 4, 4), dtype=np.float64) if want_rho else None
    st = np.zeros(num, dtype=np.int32)
    nst = np.zeros(num, dtype=np.int32)
    h = C.c_void_p()
    _lib.check(ctx._lib.ray_forward_operator(ctx.handle, med.handle, _ptr(m), num, max(num, 1), float(dt),
                                             -1.0 if droptol is None else float(droptol),
                                             _ptr(X), max(num, 1), _ptr(rho) if want_rho else None, _ptr(st), _ptr(nst),
                                             C.byref(h)))
    return FrechetMatrix(ctx, h), X, rho, {"status": st, "nsteps": nst}


def ScatterForwardOperator(c, m, ext, dt, *, ctx=None, info=False, droptol=1e-12):
    """NonObstacle.jl:189-243: returns (M, s).  s[:,1:4], s[:,9] are copied from m (:196) and
    s[:,5:8] are the traced end states (:185); M is the (4 num) x N^2 Frechet matrix as a scipy
    CSR matrix (rows ray-major like reshape(M, 4num, N^2) at :242, columns i + j*N), assembled on
    the device with entries |v| <= droptol dropped (droptol!, :242; None keeps every entry)."""
    F, X, rho, inf = forward_operator_dev(c, m, ext, dt, ctx=ctx, droptol=droptol)
    M = F.tocsr()
    F.close()
    num = X.shape[0]
    s = np.zeros((num, 9), dtype=np.float64, order="F")
    s[:, 0:4] = np.asarray(m)[:, 0:4]
    s[:, 8] = np.asarray(m)[:, 8]
    s[:, 4:8] = X
    inf["rho"] = rho
    return (M, s, inf) if info else (M, s)


def NonReflectionTrace(c, m, ext, dt, *, ctx=None, info=False, nd=0):
    """Obstacle.jl:275-329: m with columns 5..8 overwritten by the fixed-time end states.  `nd`: the fan
    width of m (rows = source * nd + direction) as a layout hint for the kernel, 0 = none."""
    X, inf = _trace(c, m, ext, dt, TRACE_OVERSHOOT, ctx, info, nd)
    out = np.array(m, dtype=np.float64, order="F", copy=True)
    out[:, 4:8] = X
    return (out, inf) if info else out


def classify_rays(m, ns, nd, *, ctx=None):
    """Utility.jl:1-39 + ObstacleDemo.jl:48-67 on the GPU.  Returns a dict with 0-based numpy index
    arrays `ortho_index` (:48-49), `unbroken` (:63-64), `unbroken_bound` (:65-66), the per-source 1-based
    `lo`, `hi` of derivativeCheck and the aligned exit angles `arg` (ns, nd)."""
    ctx = ctx or default_context()
    m = np.asfortranarray(m, dtype=np.float64)
    num = ns * nd
    if m.shape[0] != num or m.shape[1] < 8:
        raise ValueError("m must be (ns*nd, >= 8)")
    ortho = np.zeros(num, dtype=np.uint8)
    arg = np.zeros(num, dtype=np.float64)
    lo = np.zeros(ns, dtype=np.int32)
    hi = np.zeros(ns, dtype=np.int32)
    _lib.check(ctx._lib.ray_classify(ctx.handle, _ptr(m), ns, nd, max(num, 1), _ptr(ortho), _ptr(arg), _ptr(lo), _ptr(hi)))
    unbroken, bound = [], []
    for s in range(ns):                                    # ObstacleDemo.jl:63-66 (1-based -> 0-based rows)
        unbroken.append(np.arange(s * nd, s * nd + lo[s]))
        unbroken.append(np.arange(s * nd + hi[s] - 1, (s + 1) * nd))
        bound += [s * nd + lo[s] - 1, s * nd + hi[s] - 1]
    return {"ortho_index": np.flatnonzero(ortho), "lo": lo, "hi": hi, "arg": arg.reshape(ns, nd),
            "unbroken": np.concatenate(unbroken) if unbroken else np.zeros(0, dtype=np.int64),
            "unbroken_bound": np.array(bound, dtype=np.int64)}


def misfit(m, obs, perc_thresh=0.2, *, ctx=None):
    """NonObstacle.jl:336,372; ObstacleDemo.jl:219-226.  Returns dict with `mismatch` (4*num,
    ray-major), `sum_mismatch2`, `sum_target2`, `objective`, `mask` (perc < perc_thresh)."""
    ctx = ctx or default_context()
    m = np.asfortranarray(m, dtype=np.float64)
    obs = np.asfortranarray(obs, dtype=np.float64)
    if m.shape != obs.shape or m.shape[1] != 9:
        raise ValueError("m and obs must both be (num, 9)")
    num = m.shape[0]
    out = np.zeros(2)
    mask = np.zeros(num, dtype=np.uint8)
    mism = np.zeros(4 * num)
    _lib.check(ctx._lib.ray_misfit(ctx.handle, _ptr(m), _ptr(obs), num, max(num, 1), _ptr(out),
                                   _ptr(mask), float(perc_thresh), _ptr(mism)))
    return {"mismatch": mism, "sum_mismatch2": out[0], "sum_target2": out[1],
            "objective": math.sqrt(out[0] / out[1]) if out[1] > 0 else float("nan"),
            "mask": mask}
