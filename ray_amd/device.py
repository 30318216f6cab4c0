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


# ---- the loop passes over a device-resident Frechet matrix, device vectors (NonObstacle.jl:340-367) ----------
def frechet_dofs_dev(F, fidelity_ptr, dofs_ptr):
    _lib.check(F.ctx._lib.ray_frechet_dofs_dev(F.handle, _vp(fidelity_ptr), _vp(dofs_ptr)))


def frechet_matvec_dev(F, x_ptr, y_ptr):
    _lib.check(F.ctx._lib.ray_frechet_matvec_dev(F.handle, _vp(x_ptr), _vp(y_ptr)))


def frechet_matvec_t_dev(F, v_ptr, mask_ptr, b_ptr):
    _lib.check(F.ctx._lib.ray_frechet_matvec_t_dev(F.handle, _vp(v_ptr), _vp(mask_ptr), _vp(b_ptr)))


def frechet_normal_matvec_dev(F, x_ptr, mask_ptr, w_ptr):
    """w = M[order,:]' (M[order,:] x) -- one launch per Krylov iteration."""
    _lib.check(F.ctx._lib.ray_frechet_normal_matvec_dev(F.handle, _vp(x_ptr), _vp(mask_ptr), _vp(w_ptr)))


def frechet_normal_diag_dev(F, mask_ptr, d_ptr):
    _lib.check(F.ctx._lib.ray_frechet_normal_diag_dev(F.handle, _vp(mask_ptr), _vp(d_ptr)))


def frechet_fidelity_update_dev(F, x_ptr, mismatch_ptr, mask_ptr, rejection, decay, fidelity_ptr, residual_ptr=0):
    _lib.check(F.ctx._lib.ray_frechet_fidelity_update_dev(F.handle, _vp(x_ptr), _vp(mismatch_ptr), _vp(mask_ptr),
                                                          float(rejection), float(decay), _vp(fidelity_ptr), _vp(residual_ptr)))
