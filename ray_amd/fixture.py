"""RAYFIX1 -- a language-neutral container for golden vectors and ray data (SURVEY.md section 8f NEXT-4).

The reference has no on-disk format; the parity fixtures under tests/golden are NumPy .npz files,
which only Python reads.  RAYFIX1 holds the same named arrays so that the Julia shim
(julia/RayAMD.jl: read_rayfix) and C/C++ callers can check themselves against the same vectors:

    line 1   b"RAYFIX1\\n"
    line 2   one JSON object, UTF-8, terminated by "\\n":
             {"arrays": [{"name": str, "dtype": "f8"|"i4"|"i8"|"u1", "shape": [..], "offset": int}, ...]}
             offset = byte offset of the array's data from the end of line 2, 16-byte aligned
    rest     raw little-endian data, every array in COLUMN-MAJOR (Fortran / Julia) order
"""
import json

import numpy as np

MAGIC = b"RAYFIX1\n"
_DT = {"f8": np.dtype("<f8"), "i4": np.dtype("<i4"), "i8": np.dtype("<i8"), "u1": np.dtype("u1")}
_CODE = {v: k for k, v in _DT.items()}


def write_rayfix(path, arrays):
    """arrays: dict name -> array-like (float64, int32, int64 or uint8; scalars become shape ())."""
    entries, blobs, off = [], [], 0
    for name, a in arrays.items():
        a = np.asarray(a)
        if a.dtype.kind == "f":
            a = a.astype("<f8")
        elif a.dtype.kind == "b":
            a = a.astype("u1")
        elif a.dtype.kind in "iu" and a.dtype.newbyteorder("<") not in _CODE:
            a = a.astype("<i8")
        dt = a.dtype.newbyteorder("<") if a.dtype.itemsize > 1 else a.dtype
        if dt not in _CODE:
            raise TypeError(f"{name}: unsupported dtype {a.dtype}")
        data = np.asfortranarray(a).tobytes(order="F")
        entries.append({"name": name, "dtype": _CODE[dt], "shape": list(a.shape), "offset": off})
        pad = (-len(data)) % 16
        blobs.append(data + b"\0" * pad)
        off += len(data) + pad
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(json.dumps({"arrays": entries}).encode("utf-8") + b"\n")
        for b in blobs:
            f.write(b)


def read_rayfix(path):
    """-> dict name -> numpy array (Fortran-ordered; scalars as 0-d arrays)."""
    with open(path, "rb") as f:
        if f.readline() != MAGIC:
            raise ValueError(f"{path}: not a RAYFIX1 file")
        head = json.loads(f.readline().decode("utf-8"))
        body = f.read()
    out = {}
    for e in head["arrays"]:
        dt = _DT[e["dtype"]]
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        a = np.frombuffer(body, dtype=dt, count=n, offset=e["offset"])
        out[e["name"]] = a.reshape(e["shape"], order="F").copy(order="F")
    return out
