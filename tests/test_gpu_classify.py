"""SURVEY.md section 8f NEXT-3: ray classification (Utility.jl:1-39; ObstacleDemo.jl:48-67) on the GPU
against the pure-Python restatement in oracle/orc.py.  Index work: exact (lo/hi, ortho, unbroken
sets); aligned angles: 1e-12 (device atan2 vs libm)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_classify_obstacle_demo(ctx, orc):
    import ray_amd
    ns, nd = 20, 300                                            # ObstacleDemo.jl:39-40 has 50 x 300
    m, *_ = orc.scatter_relation(orc.OBSTACLE_DEMO, ns, nd, 1e-2, obstacle_r0=0.3, time_unit=1)
    m = np.asfortranarray(m)
    ortho, lo, hi, arg = orc.classify(m, ns, nd)
    got = ray_amd.classify_rays(m, ns, nd, ctx=ctx)
    assert np.array_equal(got["lo"], lo) and np.array_equal(got["hi"], hi)
    assert np.array_equal(got["ortho_index"], ortho)
    assert np.max(np.abs(got["arg"] - arg)) <= 1e-12
    # the obstacle breaks the exit-angle signal of every source somewhere: real apertures were found
    assert (hi - lo < nd - 1).any() and ((lo > 1) | (hi < nd)).any()
    unb = np.concatenate([np.r_[s * nd:s * nd + lo[s], s * nd + hi[s] - 1:(s + 1) * nd] for s in range(ns)])
    assert np.array_equal(got["unbroken"], unb)
    assert got["unbroken_bound"].tolist() == [v for s in range(ns) for v in (s * nd + lo[s] - 1, s * nd + hi[s] - 1)]


def test_classify_smooth_and_synthetic(ctx, orc):
    import ray_amd
    # smooth medium, coarse fan: the 2 pi wrap of atan2 must be removed by alignment before the check
    ns, nd = 16, 64
    m, *_ = orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, 1e-2)
    m = np.asfortranarray(m)
    ortho, lo, hi, arg = orc.classify(m, ns, nd)
    got = ray_amd.classify_rays(m, ns, nd, ctx=ctx)
    assert np.array_equal(got["lo"], lo) and np.array_equal(got["hi"], hi)
    assert np.max(np.abs(got["arg"] - arg)) <= 1e-12
    assert np.max(np.abs(np.diff(got["arg"], axis=1))) < 1.5 * np.pi      # aligned
    # synthetic: back-reflected rays (exit = entry point, reversed momentum) are "orthogonal hits"
    m2 = m.copy()
    m2[5::7, 4:6] = m2[5::7, 0:2]
    m2[5::7, 6:8] = -m2[5::7, 2:4]
    got2 = ray_amd.classify_rays(m2, ns, nd, ctx=ctx)
    o2 = orc.classify(m2, ns, nd)[0]
    assert np.array_equal(got2["ortho_index"], o2) and set(range(5, ns * nd, 7)) <= set(o2.tolist())
