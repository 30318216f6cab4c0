"""CPU tests of the boundary (no GPU): libray_amd.so loads, exports every symbol include/ray_amd.h
declares, the ctypes table covers them all, host-only entry points work, and every GPU entry point
fails LOUDLY (error code + message, no crash, no CPU fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ray_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ray_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ray_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ray_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ray_amd/_lib.py SIGNATURES out of sync with the header"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ray_[a-z0-9_]+)", out))
    assert exported == set(syms), exported ^ set(syms)


def test_library_links_rccl():
    """the node-wide group entry points issue their one collective through RCCL: libray_amd.so must carry it
    as a load-time dependency (no dlopen fallbacks)."""
    from ray_amd import _lib
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl.so" in out and "libamdhip64.so" in out, out


def test_product_never_touches_the_oracle():
    """ray_amd/ and include/ must not import, link or mention the oracle."""
    for base in ("ray_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "libray_oracle" not in txt and "import orc" not in txt and "ray_oracle.h" not in txt, f


def test_linspace_matches_oracle(orc):
    from ray_amd import _lib
    lib = _lib.load()
    for a, b, n in [(-1.5, 1.5, 75), (-1.5, 1.5, 512), (-1.5, 1.5, 1024), (0.0, 2 * np.pi, 8193),
                    (0.0, np.pi, 2050), (0.3, 0.3, 1), (-2.0, 7.0, 2)]:
        out = np.zeros(n)
        assert lib.ray_linspace(a, b, n, out.ctypes.data) == 0
        assert np.array_equal(out, orc.linspace(a, b, n)), (a, b, n)
    assert lib.ray_linspace(0.0, 1.0, 0, None) != 0
    assert b"linspace" in lib.ray_last_error()


def test_version_and_struct_layout():
    from ray_amd import _lib
    lib = _lib.load()
    assert lib.ray_version() >= 101              # 101: ray_ctx_set_stream(ctx, mode, stream)
    assert C.sizeof(_lib.Obstacle) == 32 and C.sizeof(_lib.ShootParams) == 48 and C.sizeof(_lib.Counters) == 32


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present: the no-device behaviour cannot be exercised")
def test_fails_loudly_without_gpu():
    import ray_amd
    from ray_amd import _lib
    lib = _lib.load()
    n = C.c_int(-1)
    lib.ray_device_count(C.byref(n))
    assert n.value == 0
    h = C.c_void_p()
    rc = lib.ray_ctx_create(0, C.byref(h))
    assert rc != 0 and not h.value and len(lib.ray_last_error()) > 0
    with pytest.raises(ray_amd.RayError):
        ray_amd.Context(0)
    # every compute entry point rejects a null context instead of computing anything
    z = C.c_void_p()
    assert lib.ray_shoot_to_exit(None, None, None, None, None, 0, None, None, None) != 0
    assert lib.ray_trace_fixed_time(None, None, None, 0, 0, 0.1, 0, None, 0, None, None) != 0
    assert lib.ray_misfit(None, None, None, 0, 0, None, None, 0.2, None) != 0
    assert lib.ray_medium_create_grid(None, None, 8, 1.5, C.byref(z)) != 0 and not z.value


def test_group_create_failure_returns_an_error():
    """ray_group_create whose member contexts cannot be built (a device index that does not exist; here or on a box
    with one GPU) must come back with an error code and *out == NULL -- its clean-up used to dereference the
    members that had no context yet."""
    from ray_amd import _lib
    lib = _lib.load()
    for devs in ([63], [0, 63], [62, 63, 61]):
        h = C.c_void_p(1)
        arr = (C.c_int * len(devs))(*devs)
        rc = lib.ray_group_create(arr, len(devs), C.byref(h))
        assert rc != 0 and not h.value and len(lib.ray_last_error()) > 0
    h = C.c_void_p(1)
    assert lib.ray_group_create(None, 0, C.byref(h)) != 0 and not h.value


def test_runtime_info_names_the_bound_libraries():
    import ray_amd
    info = ray_amd.runtime_info()
    assert info.startswith("hip ") and "; rccl " in info and "librccl" in info and "libamdhip64" in info


def test_markstein_quotients_equal_ieee_division(tmp_path):
    """ray_device.h computes x/6.0 and (x-p1)/h as q0 = a*r; q = fma(fma(-b,q0,a), r, q0) with
    r = RN(1/b).  The kernels' bit-exactness rests on q == a/b for every operand: checked here on
    4e7 operands per divisor, including values at and next to multiples of b."""
    src = tmp_path / "mk.c"
    src.write_text(r'''
#include <math.h>
#include <stdint.h>
#include <stdio.h>
static inline double mk(double a, double b, double rb) { double q0 = a * rb; double r = fma(-b, q0, a); return fma(r, rb, q0); }
static uint64_t s = 88172645463325252ULL;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(void) {
    double hs[] = {6.0, 3.0 / 511, 3.0 / 74, 3.0 / 44, 3.0 / 1023, 3.0 / 128};
    long bad = 0;
    for (int k = 0; k < 6; ++k) {
        double b = hs[k], rb = 1.0 / b;
        for (long i = 0; i < 40000000L; ++i) {
            uint64_t u = rnd();
            double a;
            if (k == 0) { int e = (int)(rnd() % 60) - 50; a = ldexp((double)(u >> 11) / 9007199254740992.0 + 1.0, e); if (u & 1) a = -a; }
            else {
                a = (double)(u >> 11) / 9007199254740992.0 * 3.0;
                if ((u & 7) == 0) a = floor(a / b) * b;
                if ((u & 7) == 1) a = nextafter(floor(a / b) * b, 10);
                if ((u & 7) == 2) a = nextafter(floor(a / b) * b, -10);
            }
            if (mk(a, b, rb) != a / b) ++bad;
        }
    }
    printf("%ld\n", bad);
    return 0;
}
''')
    exe = tmp_path / "mk"
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"])
    assert subprocess.check_output([str(exe)], text=True).strip() == "0"


def test_library_reads_no_environment_variable():
    """kernel shapes are options of the context (RAY_OPT_TUNE_*), never environment variables that would change a
    caller's launches behind its back (robustness 14)."""
    for f in os.listdir(os.path.join(ROOT, "ray_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "ray_amd", "csrc", f)).read(), f


def test_stencil_segments_cover_the_twelve_nodes():
    """ray_frechet.h sorts the FOUR SEGMENTS of a record's stencil instead of its twelve nodes: in node order (i + j N) the
    stencil (NonObstacle.jl:107-139: the cell's nodes, its x-neighbours, its y-neighbours) is the rows j-1 (i, i+1), j and
    j+1 (i-1 .. i+2), j+2 (i, i+1).  The header's packed constants (stencil_di / stencil_dj, and the nibble table that maps
    (segment, x-offset) back to the stencil index) must say exactly that."""
    import re
    src = open(os.path.join(ROOT, "ray_amd", "csrc", "ray_frechet.h")).read()
    di_c = int(re.search(r"stencil_di\(int k\) \{ return \(int\)\(\((0x[0-9a-f]+)u >>", src).group(1), 16)
    dj_c = int(re.search(r"stencil_dj\(int k\) \{ return \(int\)\(\((0x[0-9a-f]+)u >>", src).group(1), 16)
    tab = int(re.search(r"\(unsigned\)\((0x[0-9a-f]+)ull >> \(4 \* \(sg \* 4 \+ di \+ 1\)\)\)", src).group(1), 16)
    di = [((di_c >> (2 * k)) & 3) - 1 for k in range(12)]
    dj = [((dj_c >> (2 * k)) & 3) - 1 for k in range(12)]
    assert di == [0, 1, 0, 1, 2, -1, 2, -1, 0, 1, 0, 1] and dj == [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, -1, -1]
    seen = set()
    for sg, (lo, hi) in enumerate([(0, 1), (-1, 2), (-1, 2), (0, 1)]):          # rows j - 1 + sg
        for d in range(lo, hi + 1):
            k = (tab >> (4 * (sg * 4 + d + 1))) & 15
            assert (di[k], dj[k]) == (d, sg - 1), (sg, d, k)
            seen.add(k)
    assert seen == set(range(12))
