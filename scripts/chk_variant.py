import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import ray_amd
ctx = ray_amd.Context(0)
c = np.asfortranarray(np.ones((64, 64)))
med = ray_amd.GridMedium(c, 1.5, ctx)
m, info = ray_amd.ScatterRelation(med, None, 4, 64, 1e-2, ctx=ctx, info=True)
print(os.environ.get("RAY_AMD_LIB"), info["status"][:8], info["nsteps"][:70:5], m[:3, 4:9])
