"""ray_group: the node-wide fan-out behind the C ABI (include/ray_amd.h) -- what the reference's callers get from
addprocs / @parallel / remotecall_wait (NonObstacle.jl:59,236-240).  On the one-GPU box: a group of n contexts on
device 0 (members' sums folded on the device) and a one-member group (the RCCL all-reduce path, one rank) must
both reproduce the single-context results: m / status / step counts / mask / mismatch bit for bit, the two sums
to 1e-12.  Distinct-device groups run the same code with n ranks; that is unmeasured until an 8-GPU node exists."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0, 0]], ids=["rccl-1rank", "fold-2", "fold-5"])
def test_group_reproduces_single_context(ctx, orc, devices):
    import ray_amd
    ns, nd, dt, N = 7, 100, 1e-2, 75
    c = orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.5)
    med1 = ray_amd.GridMedium(c, 1.5, ctx)
    want, winfo = ray_amd.ScatterRelation(med1, None, ns, nd, dt, ctx=ctx, info=True)
    g = ray_amd.Group(devices)
    assert g.size == len(devices) and g.uses_rccl == (len(set(devices)) == len(devices))
    try:
        med = g.grid_medium(c, 1.5)
        m, info = g.ScatterRelation(med, ns, nd, dt, info=True)
        assert np.array_equal(m, want)
        for k in ("status", "nsteps", "nreflect"):
            assert np.array_equal(info[k], winfo[k])
        cnt = g.counters()
        assert cnt["rays"] == ns * nd and cnt["ray_steps"] == int(winfo["nsteps"].sum())
        # the observation: a perturbed copy; misfit of the RESIDENT slabs against it
        rng = np.random.default_rng(5)
        obs = np.asfortranarray(want * (1 + 0.3 * rng.standard_normal(want.shape) * (rng.random((want.shape[0], 1)) < 0.5)))
        ref = ray_amd.misfit(want, obs, 0.2, ctx=ctx)
        g.set_observation(obs)
        got = g.misfit(0.2)
        assert np.array_equal(got["mismatch"], ref["mismatch"]) and np.array_equal(got["mask"], ref["mask"])
        assert got["sum_mismatch2"] == pytest.approx(ref["sum_mismatch2"], rel=1e-12)
        assert got["sum_target2"] == pytest.approx(ref["sum_target2"], rel=1e-12)
        # a sub-range of sources, device-resident only, then the misfit again (theta-blocks of [2, 6))
        assert g.ScatterRelation(med, ns, nd, dt, src_begin=2, src_end=6, keep_on_device=True) is None
        g.set_observation(obs[2 * nd:6 * nd])
        got = g.misfit(0.2)
        ref = ray_amd.misfit(want[2 * nd:6 * nd], obs[2 * nd:6 * nd], 0.2, ctx=ctx)
        assert np.array_equal(got["mask"], ref["mask"]) and got["sum_mismatch2"] == pytest.approx(ref["sum_mismatch2"], rel=1e-12)
        # new nodal values on every replica
        c2 = np.asfortranarray(0.9 * c + 0.1)
        med.update(c2)
        med1.update(c2)
        assert np.array_equal(g.ScatterRelation(med, ns, nd, dt), ray_amd.ScatterRelation(med1, None, ns, nd, dt, ctx=ctx))
    finally:
        g.close()


def test_group_more_members_than_sources_and_obstacle(ctx, orc):
    """ns = 2 over three members (one shoots nothing); analytic medium + circle, t in steps (ObstacleDemo.jl:43)."""
    import ray_amd
    sp = ray_amd.SinProd.obstacle_demo()
    want, winfo = ray_amd.ScatterRelationObstacle(sp, None, ray_amd.Circle(0.3), None, 2, 150, 1e-2, ctx=ctx, info=True)
    g = ray_amd.Group([0, 0, 0])
    try:
        med = g.sinprod_medium(sp)
        m, info = g.ScatterRelation(med, 2, 150, 1e-2, ob=ray_amd.Circle(0.3), time_unit=ray_amd.TIME_STEPS, info=True)
        assert np.array_equal(m, want) and np.array_equal(info["nreflect"], winfo["nreflect"])
        g.set_observation(want)
        r = g.misfit()
        assert r["sum_mismatch2"] == 0.0 and r["sum_target2"] > 0 and r["mask"].all()
        with pytest.raises(ray_amd.RayError):
            ray_amd.Group([0, 0]).set_observation(want)          # no shoot yet: row count does not match
        with pytest.raises(ray_amd.RayError):
            g.ScatterRelation(ray_amd.Group([0]).sinprod_medium(sp), 2, 8, 1e-2)   # medium of another group
    finally:
        g.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]], ids=["rccl-1rank", "fold-3"])
def test_group_trace_reproduces_single_context(ctx, orc, devices):
    """ray_group_trace_fixed_time: the reconstruction iteration's traces over the members -- fans recognised in m, fans
    told, rows in no order (contiguous blocks), more members than fans -- equal the single-context traces bit for bit
    in both modes, and the misfit of the resident traced states against the data equals the single-context one."""
    import ray_amd
    ns, nd, N = 7, 60, 45
    data, _, _, _ = orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, 1e-2, nthreads=4)
    c0 = np.asfortranarray(0.85 * orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.5) + 0.1)
    med1 = ray_amd.GridMedium(c0, 1.5, ctx)
    perm = np.random.default_rng(3).permutation(ns * nd)
    shuffled = np.asfortranarray(data[perm])
    g = ray_amd.Group(devices)
    try:
        med = g.grid_medium(c0, 1.5)
        for mode in (ray_amd.TRACE_SHORTEN_CEASE, ray_amd.TRACE_OVERSHOOT):
            want, winfo = ray_amd._trace(med1, data, 1.5, 5e-2, mode, ctx, True)
            for rows, hint in ((data, 0), (data, nd), (data, ns * nd), (shuffled, 0), (data[:nd + 7], 0)):
                rows = np.asfortranarray(rows)
                X, info = g.trace(med, rows, 5e-2, mode, nd=hint, info=True)
                sel = perm if rows is shuffled else np.arange(rows.shape[0])
                assert np.array_equal(X, want[sel])
                assert np.array_equal(info["status"], winfo["status"][sel]) and np.array_equal(info["nsteps"], winfo["nsteps"][sel])
        # the loop's objective: traced states (resident) against the measured data
        X = g.trace(med, data, 5e-2, ray_amd.TRACE_SHORTEN_CEASE)
        s = np.array(data, order="F", copy=True)
        s[:, 4:8] = X
        ref = ray_amd.misfit(s, data, 0.2, ctx=ctx)
        g.set_observation(data)
        got = g.misfit(0.2)
        assert np.array_equal(got["mismatch"], ref["mismatch"]) and np.array_equal(got["mask"], ref["mask"])
        assert got["sum_mismatch2"] == pytest.approx(ref["sum_mismatch2"], rel=1e-12)
        assert got["sum_target2"] == pytest.approx(ref["sum_target2"], rel=1e-12)
        assert g.trace(med, data[:0], 5e-2).shape == (0, 4)
        with pytest.raises(ray_amd.RayError):
            g.trace(g.sinprod_medium(ray_amd.SinProd.smooth_demo()), data, 5e-2)        # gridded media only
    finally:
        g.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]], ids=["one-member", "three-members"])
def test_group_forward_operator_reproduces_single_context(ctx, orc, devices):
    """ray_group_forward_operator: (M, s) of ScatterForwardOperator built by the members concurrently equals the
    single-context one -- s bit for bit, M entry for entry (same kernels on the same rays) -- for fans, shuffled rows
    and ragged counts; the members' parts are ordinary handles (dofs on them = the single-context dofs of their rays)."""
    import ray_amd
    ns, nd, N = 5, 48, 33
    data, _, _, _ = orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, 1e-2, nthreads=4)
    c0 = np.asfortranarray(0.85 * orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.5) + 0.1)
    med1 = ray_amd.GridMedium(c0, 1.5, ctx)
    perm = np.random.default_rng(4).permutation(ns * nd)
    g = ray_amd.Group(devices)
    try:
        med = g.grid_medium(c0, 1.5)
        for rows in (data, np.asfortranarray(data[perm]), np.asfortranarray(data[:nd + 5])):
            M1, s1 = ray_amd.ScatterForwardOperator(med1, rows, 1.5, 5e-2, ctx=ctx)
            M, s, info = g.ScatterForwardOperator(med, rows, 5e-2, info=True)
            assert np.array_equal(s, s1)
            A, A1 = M.tocsr(), M1
            assert A.shape == A1.shape and A.nnz == A1.nnz
            assert np.array_equal(A.indptr, A1.indptr) and np.array_equal(A.indices, A1.indices)
            assert np.array_equal(A.data, A1.data)
            fid = np.linspace(0.0, 1.0, N * N)
            Md = ray_amd.forward_operator_dev(med1, rows, 1.5, 5e-2, ctx=ctx, want_rho=False)[0]
            want = Md.dofs(fid)
            for part, rays in M.members():
                assert np.allclose(part.dofs(fid), want[rays], rtol=0, atol=1e-12)
            # the gradient sums: one all-reduce of N^2 doubles over the members (fp64 atomics: order-dependent rounding)
            rng = np.random.default_rng(8)
            v, x = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[1])
            mask = (rng.random(A.shape[0] // 4) < 0.7).astype(np.uint8)
            sel = np.repeat(mask.astype(bool), 4)
            for mk, rows_sel in ((None, slice(None)), (mask, sel)):
                b, w = M.matvec_t(v, mk), M.normal_matvec(x, mk)
                b_ref = A1[rows_sel].T @ v[rows_sel]
                w_ref = A1[rows_sel].T @ (A1[rows_sel] @ x)
                assert np.max(np.abs(b - b_ref)) <= 1e-12 * max(1.0, np.max(np.abs(b_ref)))
                assert np.max(np.abs(w - w_ref)) <= 1e-12 * max(1.0, np.max(np.abs(w_ref)))
            M.close()
        M, s = g.ScatterForwardOperator(med, data[:0], 5e-2)
        assert M.shape == (0, N * N) and M.nnz == 0 and s.shape == (0, 9)
        with pytest.raises(ray_amd.RayError):
            g.ScatterForwardOperator(g.sinprod_medium(ray_amd.SinProd.smooth_demo()), data, 5e-2)
    finally:
        g.close()


def test_group_create_with_a_missing_device_fails_cleanly(ctx):
    """the failure path of ray_group_create (member k built, k+1.. not) returns an error, no crash."""
    import ray_amd
    for devs in ([63], [0, 63]):
        with pytest.raises(ray_amd.RayError):
            ray_amd.Group(devs)
    g = ray_amd.Group([0])                       # and the library still works afterwards
    assert g.size == 1
    g.close()


def test_group_frechet_outlives_its_group(ctx, orc):
    """a GroupFrechetMatrix whose group was closed reports an error on every pass (it used to read freed
    memory) and can still be destroyed."""
    import ray_amd
    c = orc.sinprod_grid(orc.SMOOTH_DEMO, 31, 1.5)
    m, *_ = orc.scatter_relation(orc.SMOOTH_DEMO, 3, 8, 1e-2)
    g = ray_amd.Group([0, 0])
    med = g.grid_medium(c, 1.5)
    M, s = g.ScatterForwardOperator(med, np.asfortranarray(m), 5e-2, nd=8)
    b = M.matvec_t(np.ones(M.shape[0]))
    assert np.isfinite(b).all() and np.abs(b).max() > 0
    g.close()
    with pytest.raises(ray_amd.RayError):
        M.matvec_t(np.ones(M.shape[0]))
    with pytest.raises(ray_amd.RayError):
        M.normal_matvec(np.ones(M.shape[1]))
    with pytest.raises(ray_amd.RayError):
        M.tocsr()
    M.close()
