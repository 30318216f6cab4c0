"""(work in progress: run by naming this file) ray_misfit_static_dev under HIP graph capture."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_static_misfit_under_graph_capture(orc):
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
