"""ray_misfit_static_dev -- the allocation-free entry point (include/ray_amd.h) -- through the C ABI against
ray_misfit_dev, bit for bit, on two calls in a row, the second with a smaller input (the static scratch keeps
the first call's contents).  Collected last (test_zz_*) so that a failure here hides no other test."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(lib, ctx, fn, m, obs, num, ld, thresh):
    import hiprt
    out = hiprt.DevBuf.from_numpy(np.full(2, -1.0))
    mask = hiprt.DevBuf.from_numpy(np.full(max(num, 1), 7, dtype=np.uint8))
    mism = hiprt.DevBuf.from_numpy(np.full(4 * max(num, 1), -2.0))
    rc = fn(ctx.handle, m.ptr, obs.ptr, num, ld, out.ptr, mask.ptr, thresh, mism.ptr)
    assert rc == 0, lib.ray_last_error()
    ctx.synchronize()
    return out.to_numpy(np.float64, 2), mask.to_numpy(np.uint8, max(num, 1))[:num], mism.to_numpy(np.float64, 4 * max(num, 1))[:4 * num]


def test_static_misfit_matches_misfit_dev_twice(ctx, orc):
    import hiprt
    from ray_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(99)
    ld = 300001
    host_m = np.asfortranarray(rng.standard_normal((ld, 9)))
    host_o = np.asfortranarray(host_m + 0.15 * rng.standard_normal((ld, 9)) * rng.random((ld, 1)))
    m, obs = hiprt.DevBuf.from_numpy(host_m), hiprt.DevBuf.from_numpy(host_o)     # column-major (ld, 9)
    scratch_out = hiprt.DevBuf.from_numpy(np.zeros(2))
    for num in (300001, 4097, 0):                                       # big, then smaller, then empty
        want = _run(lib, ctx, lib.ray_misfit_dev, m, obs, num, ld, 0.2)
        got = _run(lib, ctx, lib.ray_misfit_static_dev, m, obs, num, ld, 0.2)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        # the property the entry point exists for: a warm call allocates nothing on the device
        free0 = hiprt.mem_free_bytes()
        rc = lib.ray_misfit_static_dev(ctx.handle, m.ptr, obs.ptr, num, ld, scratch_out.ptr, None, 0.2, None)
        assert rc == 0, lib.ray_last_error()
        ctx.synchronize()
        assert hiprt.mem_free_bytes() == free0
        assert np.array_equal(scratch_out.to_numpy(np.float64, 2), got[0])
        # and against the oracle (the existing entry point's own bar)
        mism, out, mask = orc.misfit(np.asfortranarray(host_m[:num]), np.asfortranarray(host_o[:num]), 0.2)
        assert np.array_equal(got[2], mism) and np.array_equal(got[1], mask)
        assert np.allclose(got[0], out, rtol=1e-12, atol=0)


def test_static_misfit_rejects_oversize(ctx):
    from ray_amd import _lib
    lib = _lib.load()
    big = (1 << 28) + 1
    rc = lib.ray_misfit_static_dev(ctx.handle, C.c_void_p(8), C.c_void_p(8), big, big, C.c_void_p(8), None, 0.2, None)
    assert rc == -4 and b"RAY_MISFIT_STATIC_MAX_RAYS" in lib.ray_last_error()


def test_static_misfit_under_graph_capture(orc):
    """The caller the entry point is for: the call recorded into a HIP graph on a borrowed stream (a capture admits no
    allocation, no synchronisation), the graph replayed, the sums equal to the eager call's bit for bit."""
    import hiprt
    import ray_amd
    from ray_amd import _lib
    lib = _lib.load()
    hip = hiprt.hip()
    ctx = ray_amd.Context(0)
    rng = np.random.default_rng(7)
    num = 50001
    host_m = np.asfortranarray(rng.standard_normal((num, 9)))
    host_o = np.asfortranarray(host_m + 0.1 * rng.standard_normal((num, 9)))
    m, obs = hiprt.DevBuf.from_numpy(host_m), hiprt.DevBuf.from_numpy(host_o)
    out = hiprt.DevBuf.from_numpy(np.full(2, -1.0))
    mism = hiprt.DevBuf.from_numpy(np.full(4 * num, -2.0))
    s = hiprt.stream_create()
    ctx.set_stream(s)
    # warm call outside capture (loads the code object), then the same call captured into a graph
    assert lib.ray_misfit_static_dev(ctx.handle, m.ptr, obs.ptr, num, num, out.ptr, None, 0.2, mism.ptr) == 0
    hiprt.stream_sync(s)
    want = out.to_numpy(np.float64, 2)
    free0 = hiprt.mem_free_bytes()
    assert hip.hipStreamBeginCapture(C.c_void_p(s), C.c_int(2)) == 0          # hipStreamCaptureModeRelaxed
    rc = lib.ray_misfit_static_dev(ctx.handle, m.ptr, obs.ptr, num, num, out.ptr, None, 0.2, mism.ptr)
    graph = C.c_void_p()
    assert hip.hipStreamEndCapture(C.c_void_p(s), C.byref(graph)) == 0
    assert rc == 0, lib.ray_last_error()
    exe = C.c_void_p()
    assert hip.hipGraphInstantiate(C.byref(exe), graph, None, None, C.c_size_t(0)) == 0
    out.fill_bytes_async(0, s)
    assert hip.hipGraphLaunch(exe, C.c_void_p(s)) == 0
    hiprt.stream_sync(s)
    got = out.to_numpy(np.float64, 2)
    assert np.array_equal(got, want)
    assert hip.hipGraphExecDestroy(exe) == 0 and hip.hipGraphDestroy(graph) == 0
    assert hiprt.mem_free_bytes() >= free0 - (1 << 21)                        # the graph's own bookkeeping aside: nothing allocated
    ctx.use_own_stream()
    hiprt.stream_destroy(s)
    ctx.close()
