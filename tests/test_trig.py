"""The reproducible sin/cos of the analytic media (ray_amd/csrc/ray_trig.h, host + device code) and its restatement in
the oracle (oracle/ray_oracle.c, flavour trig = 1): accuracy against long double, the two against each other bit for bit,
both against an exact-arithmetic reading of the same sequence of roundings, and the flavour against the libm one on the
shipped sweep.  CPU only: the header is compiled for the host with g++ (-ffp-contract=off, as the library is built)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include "ray_trig.h"
extern "C" void t_sincos(const double *a, long n, double *s, double *c)
{ for (long i = 0; i < n; ++i) ray::trig::sincos_bounded(a[i], s[i], c[i]); }
extern "C" void t_shift(const double *s0, const double *c0, const double *d, long n, double *s, double *c)
{ for (long i = 0; i < n; ++i) ray::trig::shift(s0[i], c0[i], d[i], s[i], c[i]); }
extern "C" void t_shift_fused(const double *s0, const double *c0, const double *d, long n, double *s, double *c)
{ for (long i = 0; i < n; ++i) ray::trig::shift_fused(s0[i], c0[i], d[i], s[i], c[i]); }
extern "C" void t_shift_tiny(const double *s0, const double *c0, const double *d, long n, double *s, double *c)
{ for (long i = 0; i < n; ++i) ray::trig::shift_tiny(s0[i], c0[i], d[i], s[i], c[i]); }
extern "C" int t_anchor_period(double w, double dt, double cmax) { return ray::trig::anchor_period(w, dt, cmax); }
"""


@pytest.fixture(scope="module")
def hdr(tmp_path_factory):
    d = tmp_path_factory.mktemp("trig")
    src, so = d / "h.cpp", d / "h.so"
    src.write_text(HARNESS)
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                           "-I", os.path.join(ROOT, "ray_amd", "csrc"), "-o", str(so), str(src)])
    return C.CDLL(str(so))


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


def hdr_sincos(hdr, a):
    s, c = np.empty_like(a), np.empty_like(a)
    hdr.t_sincos(_dp(a), C.c_long(a.size), _dp(s), _dp(c))
    return s, c


def hdr_shift(hdr, s0, c0, d):
    s, c = np.empty_like(d), np.empty_like(d)
    hdr.t_shift(_dp(s0), _dp(c0), _dp(d), C.c_long(d.size), _dp(s), _dp(c))
    return s, c


def hdr_shift_fused(hdr, s0, c0, d):
    s, c = np.empty_like(d), np.empty_like(d)
    hdr.t_shift_fused(_dp(s0), _dp(c0), _dp(d), C.c_long(d.size), _dp(s), _dp(c))
    return s, c


def ulps(got, ref):
    """|got - ref| in units of the last place of ref (ref in long double)."""
    return np.abs((got.astype(np.longdouble) - ref) / np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble))


@pytest.mark.parametrize("scale,bound", [(0.8, 1.0), (12.0, 1.0), (100.0, 1.0), (2.6e5, 4.0)])
def test_sincos_within_one_ulp(hdr, scale, bound):
    """arguments of the unit disk's media are below 2 pi * 1.1; the reduction keeps 1 ulp up to 100 and a few ulp
    (results next to a zero of sin/cos) up to its limit 2^18."""
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("no extended precision on this host")
    rng = np.random.default_rng(11)
    a = rng.uniform(-scale, scale, 400000)
    a = a[np.abs(a) < 262144.0]
    s, c = hdr_sincos(hdr, a)
    al = a.astype(np.longdouble)
    assert float(ulps(s, np.sin(al)).max()) <= bound
    assert float(ulps(c, np.cos(al)).max()) <= bound


def test_shift_accuracy(hdr):
    """stages 2-4 of a step: absolute error against the exact sin/cos of the stage's own argument."""
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("no extended precision on this host")
    rng = np.random.default_rng(12)
    a0 = rng.uniform(-8, 8, 400000)
    a1 = a0 + rng.uniform(-0.25, 0.25, a0.size)
    d = a1 - a0
    keep = np.abs(d) <= 0.25
    a0, a1, d = a0[keep], a1[keep], d[keep]
    s0, c0 = hdr_sincos(hdr, a0)
    s, c = hdr_shift(hdr, s0, c0, d)
    al = a1.astype(np.longdouble)
    assert float(np.abs(s - np.sin(al)).max()) <= 2.3e-16           # measured 1.6e-16; a correctly rounded value: 1.1e-16
    assert float(np.abs(c - np.cos(al)).max()) <= 2.3e-16


def test_fused_shift_accuracy_and_restatements(hdr, orc):
    """the contracted advance of the fused analytic flavour: as accurate as the plain one (one advance from a full
    evaluation), the C oracle's and the Python restatement's (exact-rational fma) readings equal the header's bit for
    bit; the anchor period is the same integer in all three."""
    import ray_oracle as P
    rng = np.random.default_rng(21)
    a0 = rng.uniform(-8, 8, 200000)
    d = rng.uniform(-0.25, 0.25, a0.size)
    s0, c0 = hdr_sincos(hdr, a0)
    s, c = hdr_shift_fused(hdr, s0, c0, d)
    if np.finfo(np.longdouble).nmant >= 63:
        al = a0.astype(np.longdouble) + d.astype(np.longdouble)
        assert float(np.abs(s - np.sin(al)).max()) <= 2.3e-16
        assert float(np.abs(c - np.cos(al)).max()) <= 2.3e-16
    for i in range(0, 20000):
        assert orc.rep_shift_fused(float(s0[i]), float(c0[i]), float(d[i])) == (s[i], c[i])
    for i in range(0, 400):
        assert P.rep_shift_fused(float(s0[i]), float(c0[i]), float(d[i])) == (s[i], c[i])
    hdr.t_anchor_period.argtypes = [C.c_double] * 3
    for w, dt, cmax in [(2 * math.pi, 1e-2, 1.2), (math.pi, 1e-2, 1.2), (1.5 * math.pi, 1e-2, 1.3), (math.pi, 5e-3, 2.1),
                        (0.0, 1e-2, 1.0), (1e-9, 1e-2, 1.0), (50.0, 1e-1, 2.0), (3.0, 2e-2, 1.25), (float("nan"), 1e-2, 1.0)]:
        k = hdr.t_anchor_period(w, dt, cmax)
        assert 1 <= k <= 64 and k == orc.anchor_period(w, dt, cmax)
        if w == w:
            assert k == P.anchor_period(w, dt, cmax) if w != 0.0 else k == 64


def test_tiny_shift_accuracy_and_restatements(hdr, orc):
    """the fused analytic flavour's advance from the stage before (|d| <= 2^-10): two advances away from a full evaluation
    (full -> shift_fused -> shift_tiny) the result is within 3.5e-16 of the exact sin/cos of the final argument; the C
    oracle's and the Python restatement's readings equal the header's bit for bit."""
    import ray_oracle as P
    rng = np.random.default_rng(22)
    a0 = rng.uniform(-8, 8, 200000)
    d1 = rng.uniform(-0.25, 0.25, a0.size)
    d2 = rng.uniform(-1, 1, a0.size) * 0.0009765625
    s0, c0 = hdr_sincos(hdr, a0)
    s1, c1 = hdr_shift_fused(hdr, s0, c0, d1)
    s, c = np.empty_like(d2), np.empty_like(d2)
    hdr.t_shift_tiny(_dp(s1), _dp(c1), _dp(d2), C.c_long(d2.size), _dp(s), _dp(c))
    if np.finfo(np.longdouble).nmant >= 63:
        al = a0.astype(np.longdouble) + d1.astype(np.longdouble) + d2.astype(np.longdouble)
        assert float(np.abs(s - np.sin(al)).max()) <= 3.5e-16
        assert float(np.abs(c - np.cos(al)).max()) <= 3.5e-16
    for i in range(0, 20000):
        assert orc.rep_shift_tiny(float(s1[i]), float(c1[i]), float(d2[i])) == (s[i], c[i])
    for i in range(0, 400):
        assert P.rep_shift_tiny(float(s1[i]), float(c1[i]), float(d2[i])) == (s[i], c[i])
    assert P.REP_TINY_MAX == 2.0 ** -10


def test_oracle_restatement_equals_header(hdr, orc):
    rng = np.random.default_rng(13)
    a = np.concatenate([rng.uniform(-12, 12, 20000), rng.uniform(-2.6e5, 2.6e5, 2000),
                        np.array([0.0, -0.0, math.pi / 4, math.pi / 2, math.pi, 1e-300, 262143.99])])
    s, c = hdr_sincos(hdr, a)
    for i in range(a.size):
        assert orc.rep_sincos(float(a[i])) == (s[i], c[i]), a[i]
    d = rng.uniform(-0.25, 0.25, 20000)
    s2, c2 = hdr_shift(hdr, s[:d.size].copy(), c[:d.size].copy(), d)
    for i in range(d.size):
        assert orc.rep_shift(float(s[i]), float(c[i]), float(d[i])) == (s2[i], c2[i])
    # outside the reproducible range the oracle is the libm
    assert orc.rep_sincos(1e7) == (math.sin(1e7), math.cos(1e7))
    assert all(math.isnan(v) for v in orc.rep_sincos(float("nan")))


# ---- the Python restatement's reading of the same sequence: exact rational arithmetic, each operation rounded once
def test_exact_arithmetic_reading(orc):
    """the decimal coefficients of both sources denote the doubles this reading builds from small-integer quotients
    (the Taylor ones) or parses from the same digits (fdlibm's), and every step rounds once."""
    import ray_oracle as P
    rng = np.random.default_rng(14)
    for a in list(rng.uniform(-12, 12, 400)) + [0.0, 3.0, -7.75, 100.5, -2.5e5, 1e7]:
        assert P.rep_sincos(float(a)) == orc.rep_sincos(float(a)), a
    for _ in range(400):
        a0, d = float(rng.uniform(-8, 8)), float(rng.uniform(-0.25, 0.25))
        s0, c0 = orc.rep_sincos(a0)
        assert P.rep_shift(s0, c0, d) == orc.rep_shift(s0, c0, d)


def test_flavour_stays_on_the_libm_one(orc):
    """the reproducible flavour against the as-written one (host libm) on the shipped sweeps: same events, values
    within 1e-10 (measured ~1e-13) -- it is the same computation up to the last bit of sin/cos."""
    for prm, ob, tu in ((orc.SMOOTH_DEMO, None, 0), (orc.OBSTACLE_DEMO, 0.3, 1)):
        m0, st0, n0, r0 = orc.scatter_relation(prm, 50, 120, 1e-2, obstacle_r0=ob, time_unit=tu, nthreads=4)
        m1, st1, n1, r1 = orc.scatter_relation(prm, 50, 120, 1e-2, obstacle_r0=ob, time_unit=tu, nthreads=4, trig=1)
        assert np.array_equal(st0, st1) and np.array_equal(n0, n1) and np.array_equal(r0, r1)
        assert np.max(np.abs(m0 - m1) / np.maximum(1.0, np.abs(m0))) <= 1e-10
        m2, st2, n2, r2 = orc.scatter_relation(prm, 50, 120, 1e-2, obstacle_r0=ob, time_unit=tu, nthreads=4, trig=1, arith=1)
        assert np.array_equal(st0, st2) and np.array_equal(n0, n2) and np.array_equal(r0, r2)     # the fused flavour too
        assert np.max(np.abs(m0 - m2) / np.maximum(1.0, np.abs(m0))) <= 1e-10
    m0, st0, n0, _ = orc.scatter_relation(orc.SMOOTH_ALT, 20, 60, 1e-2, nthreads=4)
    for arith in (0, 1):
        m1, st1, n1, _ = orc.scatter_relation(orc.SMOOTH_ALT, 20, 60, 1e-2, nthreads=4, trig=1, arith=arith)
        assert np.array_equal(st0, st1) and np.array_equal(n0, n1)
        assert np.max(np.abs(m0 - m1) / np.maximum(1.0, np.abs(m0))) <= 1e-10
