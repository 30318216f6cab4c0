#!/usr/bin/env python3
"""tests/golden/*.npz -> *.rayfix (RAYFIX1, ray_amd/fixture.py) for the Julia shim and C callers.
usage: scripts/npz_to_rayfix.py [out_dir]   (default: gpurun_out/rayfix; the .npz stay the tracked fixtures)"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd.fixture import write_rayfix  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "rayfix")
os.makedirs(out, exist_ok=True)
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    z = np.load(f, allow_pickle=False)
    dst = os.path.join(out, os.path.basename(f)[:-4] + ".rayfix")
    write_rayfix(dst, {k: z[k] for k in z.files})
    print(dst)
