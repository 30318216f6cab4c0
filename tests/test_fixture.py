"""RAYFIX1 container (ray_amd/fixture.py): round trip of every golden fixture, layout facts a
non-Python reader relies on (column-major data, 16-byte aligned offsets, little endian)."""
import json
import os

import numpy as np

from conftest import GOLDEN


def test_rayfix_roundtrip_all_golden(tmp_path):
    from ray_amd.fixture import read_rayfix, write_rayfix
    for name in sorted(os.listdir(GOLDEN)):
        if not name.endswith(".npz"):
            continue
        z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        p = tmp_path / (name[:-4] + ".rayfix")
        write_rayfix(p, {k: z[k] for k in z.files})
        back = read_rayfix(p)
        assert set(back) == set(z.files)
        for k in z.files:
            assert back[k].shape == z[k].shape
            assert np.array_equal(back[k], z[k])


def test_rayfix_layout(tmp_path):
    from ray_amd.fixture import MAGIC, write_rayfix
    a = np.arange(6, dtype=np.float64).reshape(2, 3)           # [[0,1,2],[3,4,5]]
    p = tmp_path / "t.rayfix"
    write_rayfix(p, {"a": a, "n": np.int32(7), "mask": np.array([1, 0, 1], dtype=np.uint8)})
    raw = open(p, "rb").read()
    assert raw.startswith(MAGIC)
    nl = raw.index(b"\n", len(MAGIC))
    head = json.loads(raw[len(MAGIC):nl])
    body = raw[nl + 1:]
    ent = {e["name"]: e for e in head["arrays"]}
    assert ent["a"]["dtype"] == "f8" and ent["a"]["shape"] == [2, 3] and all(e["offset"] % 16 == 0 for e in ent.values())
    col_major = np.frombuffer(body, dtype="<f8", count=6, offset=ent["a"]["offset"])
    assert col_major.tolist() == [0, 3, 1, 4, 2, 5]              # first index fastest, as Julia stores it
    assert np.frombuffer(body, dtype="<i4", count=1, offset=ent["n"]["offset"])[0] == 7
