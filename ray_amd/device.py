"""Device-resident calls (the *_dev entry points of include/ray_amd.h) for callers that keep
m / status / observation in HBM -- bench.py and the multi-GPU sweep.  Pointers are plain ints
(e.g. torch.Tensor.data_ptr()); this module does not import torch."""
import ctypes as C
import math

from . import _lib
from . import Context, GridMedium, SinProd, Circle, TIME_SECONDS  # noqa: F401


def _vp(p):
    return C.c_void_p(int(p)) if p else None


def shoot_to_exit_dev(ctx, med, ns, nd, dt, m_ptr, ld, *, ob=None, dir=(0.0, math.pi), src_begin=0,
                      src_end=None, time_unit=TIME_SECONDS, max_steps=0, status_ptr=0, nsteps_ptr=0,
                      nreflect_ptr=0):
    """ray_shoot_to_exit_dev: asynchronous on the context's stream."""
    src_end = ns if src_end is None else src_end
    prm = _lib.ShootParams(ns, nd, src_begin, src_end, float(dir[0]), float(dir[1]), float(dt),
                           time_unit, max_steps)
    obs = ob._struct() if ob is not None else None
    _lib.check(ctx._lib.ray_shoot_to_exit_dev(ctx.handle, med._medium(ctx).handle,
                                              C.byref(obs) if obs else None, C.byref(prm),
                                              _vp(m_ptr), ld, _vp(status_ptr), _vp(nsteps_ptr),
                                              _vp(nreflect_ptr)))


def shoot_states_dev(ctx, med, dt, m_ptr, num, ld, *, ob=None, time_unit=TIME_SECONDS, max_steps=0,
                     status_ptr=0, nsteps_ptr=0, nreflect_ptr=0):
    obs = ob._struct() if ob is not None else None
    _lib.check(ctx._lib.ray_shoot_states_dev(ctx.handle, med._medium(ctx).handle,
                                             C.byref(obs) if obs else None, float(dt), time_unit,
                                             max_steps, _vp(m_ptr), num, ld, _vp(status_ptr),
                                             _vp(nsteps_ptr), _vp(nreflect_ptr)))


def trace_fixed_time_dev(ctx, med, m_ptr, num, ld, dt, mode, X_ptr, ldx, *, status_ptr=0, nsteps_ptr=0):
    _lib.check(ctx._lib.ray_trace_fixed_time_dev(ctx.handle, med.handle, _vp(m_ptr), num, ld,
                                                 float(dt), mode, _vp(X_ptr), ldx, _vp(status_ptr),
                                                 _vp(nsteps_ptr)))


def misfit_dev(ctx, m_ptr, obs_ptr, num, ld, out_ptr, *, mask_ptr=0, perc_thresh=0.2, mismatch_ptr=0):
    """out_ptr -> 2 doubles in HBM: {sum mismatch^2, sum target^2} (what the all-reduce carries)."""
    _lib.check(ctx._lib.ray_misfit_dev(ctx.handle, _vp(m_ptr), _vp(obs_ptr), num, ld, _vp(out_ptr),
                                       _vp(mask_ptr), float(perc_thresh), _vp(mismatch_ptr)))
