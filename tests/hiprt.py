"""Minimal HIP runtime access for the GPU tests (ctypes on the libamdhip64 the product library already
loaded): device buffers, streams, async copies -- without importing torch, whose wheel brings a second HIP
runtime of its own that cannot initialise once another one owns the device in the same process."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        for name in ("hipMalloc", "hipFree", "hipMemcpy", "hipMemcpyAsync", "hipMemsetAsync", "hipStreamCreateWithFlags",
                     "hipStreamDestroy", "hipStreamSynchronize", "hipDeviceSynchronize"):
            getattr(_hip, name).restype = C.c_int
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


H2D, D2H = 1, 2


class DevBuf:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        _ok(hip().hipMalloc(C.byref(self.ptr), C.c_size_t(max(self.nbytes, 8))), "hipMalloc")

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a) if not a.flags.f_contiguous else a
        b = cls(a.nbytes)
        if a.nbytes:
            _ok(hip().hipMemcpy(b.ptr, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), H2D), "hipMemcpy H2D")
        return b

    def to_numpy(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            _ok(hip().hipMemcpy(C.c_void_p(out.ctypes.data), self.ptr, C.c_size_t(out.nbytes), D2H), "hipMemcpy D2H")
        return out

    def fill_bytes_async(self, value, stream):
        _ok(hip().hipMemsetAsync(self.ptr, C.c_int(value), C.c_size_t(self.nbytes), C.c_void_p(stream)), "hipMemsetAsync")

    def upload_async(self, a, stream):
        _ok(hip().hipMemcpyAsync(self.ptr, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), H2D, C.c_void_p(stream)), "hipMemcpyAsync H2D")

    def download_async(self, out, stream):
        _ok(hip().hipMemcpyAsync(C.c_void_p(out.ctypes.data), self.ptr, C.c_size_t(out.nbytes), D2H, C.c_void_p(stream)), "hipMemcpyAsync D2H")

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def download(ptr, count, dtype):
    """count elements of dtype from the device address `ptr` (blocking copy: waits for the device)."""
    out = np.empty(count, dtype=dtype)
    if out.nbytes:
        _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
        _ok(hip().hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), D2H), "hipMemcpy D2H")
    return out


def mem_free_bytes():
    free, total = C.c_size_t(), C.c_size_t()
    _ok(hip().hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    return free.value


def stream_create(nonblocking=True):
    s = C.c_void_p()
    _ok(hip().hipStreamCreateWithFlags(C.byref(s), C.c_uint(1 if nonblocking else 0)), "hipStreamCreateWithFlags")
    return s.value


def stream_sync(stream):
    _ok(hip().hipStreamSynchronize(C.c_void_p(stream)), "hipStreamSynchronize")


def stream_destroy(stream):
    hip().hipStreamDestroy(C.c_void_p(stream))
