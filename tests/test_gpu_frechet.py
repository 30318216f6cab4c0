"""GPU parity of the complete ScatterForwardOperator (SURVEY.md section 8f NEXT-1; NonObstacle.jl:189-243)
through the C ABI: end states bit-exact against the fixed-time trace, Jacobian rho and Frechet
matrix M against the oracle's restatement by tolerance (the reference itself goes through LAPACK
inv / BLAS gemm for these, so no bit pattern is defined; 1e-10 relative to the largest entry)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MTOL = 1e-10


@pytest.fixture(params=[0, 1], ids=["exact", "fused"])
def arith(ctx, request):
    import ray_amd
    ctx.set_arith(request.param)
    yield request.param
    ctx.set_arith(ray_amd.ARITH_FUSED)


def _data(orc, N, ns, nd, arith, bump=0.0):
    c = orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.5)
    tab = orc.GridTables(c, 1.5, arith)
    m, *_ = orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, 1e-2)
    if bump:
        m[:, 8] *= 1.0 + bump * np.arange(m.shape[0]) / m.shape[0]
    return c, tab, np.asfortranarray(m)


def test_forward_operator_long_rays(ctx, orc):
    """rays of ~500-1200 steps (dt = 2e-3): 16384-key sorts that loop over key groups, records read in place by the
    merge (more than 224 rows), mixed with short and zero-time rays in one call; and the step limit."""
    import ray_amd
    c, tab, m = _data(orc, 31, 3, 8, 1)
    m[5, 8] = 0.0                                                  # a ray that takes no step
    m[7, 8] *= 0.05                                                # and a short one
    dt = 2e-3
    Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, dt)
    assert 700 < nso.max() <= 1364 and nso.min() == 0
    M, s, info = ray_amd.ScatterForwardOperator(c, m, 1.5, dt, ctx=ctx, info=True, droptol=None)
    assert np.array_equal(info["nsteps"], nso) and np.array_equal(info["status"], sto)
    assert np.array_equal(s[:, 4:8], Xo)
    assert np.max(np.abs(info["rho"] - rho_o)) <= MTOL * np.max(np.abs(rho_o))
    assert np.max(np.abs(M.toarray() - Mo)) <= MTOL * np.max(np.abs(Mo))
    assert M[4 * 5:4 * 5 + 4].nnz == 0
    with pytest.raises(ray_amd.RayError):                          # more than 1364 steps: the per-ray merge is sized for 16384 entries
        ray_amd.ScatterForwardOperator(c, m, 1.5, 5e-4, ctx=ctx)


@pytest.mark.parametrize("N,ns,nd,dt", [(31, 4, 16, 5e-2), (61, 5, 13, 5e-2), (75, 3, 70, 2e-2)])
def test_forward_operator_matches_oracle(ctx, orc, arith, N, ns, nd, dt):
    import ray_amd
    c, tab, m = _data(orc, N, ns, nd, arith)
    Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, dt)
    med = ray_amd.GridMedium(c, 1.5, ctx)
    M, s, info = ray_amd.ScatterForwardOperator(med, m, 1.5, dt, ctx=ctx, info=True, droptol=None)
    assert np.array_equal(info["nsteps"], nso)
    assert np.array_equal(info["status"], sto)
    assert np.array_equal(s[:, 4:8], Xo)                       # same expressions as the trace kernel
    Xt, _ = ray_amd._trace(med, m, 1.5, dt, ray_amd.TRACE_SHORTEN_CEASE, ctx, True)
    assert np.array_equal(s[:, 4:8], Xt)
    assert np.max(np.abs(info["rho"] - rho_o)) <= MTOL * np.max(np.abs(rho_o))
    Md = M.toarray()
    assert Md.shape == Mo.shape
    assert np.max(np.abs(Md - Mo)) <= MTOL * np.max(np.abs(Mo))
    # symplectic flow: det(rho) = 1 up to the integrator's O(dt^4) defect
    assert np.max(np.abs(np.linalg.det(info["rho"]) - 1.0)) < 1e-3


def test_forward_operator_droptol_and_shape(ctx, orc):
    import ray_amd
    c, tab, m = _data(orc, 41, 4, 9, 1)
    M, s = ray_amd.ScatterForwardOperator(c, m, 1.5, 5e-2, ctx=ctx)
    assert M.shape == (4 * m.shape[0], 41 * 41)
    assert M.nnz > 0 and np.all(np.abs(M.data) > 1e-12)
    assert M.has_canonical_format                             # sorted columns, no duplicates
    Mall, _ = ray_amd.ScatterForwardOperator(c, m, 1.5, 5e-2, ctx=ctx, droptol=None)
    keep = np.abs(Mall.data) > 1e-12                          # droptol! (:242) is a per-entry filter
    assert M.nnz == int(keep.sum()) and np.array_equal(M.data, Mall.data[keep])
    # the four rows of a ray share one stencil pattern before the drop
    ip = Mall.indptr
    assert np.array_equal(Mall.indices[ip[0]:ip[1]], Mall.indices[ip[1]:ip[2]])
    assert np.array_equal(s[:, 0:4], m[:, 0:4]) and np.array_equal(s[:, 8], m[:, 8])


def test_forward_operator_cease_and_zero_time(ctx, orc, arith):
    """travel times stretched by up to 60%: late rays leave the 1+2h disc and take the cease branch
    (NonObstacle.jl:177-181); rows with T = 0 contribute nothing and keep rho = I."""
    import ray_amd
    c, tab, m = _data(orc, 51, 4, 20, arith, bump=0.6)
    m[::7, 8] = 0.0
    Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, 5e-2)
    assert (sto & 16).any()
    M, s, info = ray_amd.ScatterForwardOperator(c, m, 1.5, 5e-2, ctx=ctx, info=True, droptol=None)
    assert np.array_equal(info["status"], sto) and np.array_equal(info["nsteps"], nso)
    assert np.array_equal(s[:, 4:8], Xo)
    assert np.max(np.abs(M.toarray() - Mo)) <= MTOL * np.max(np.abs(Mo))
    assert np.array_equal(info["rho"][::7], np.broadcast_to(np.eye(4), info["rho"][::7].shape))
    assert M[0:4].nnz == 0


def test_forward_operator_empty_and_errors(ctx, orc):
    import ray_amd
    c = orc.sinprod_grid(orc.SMOOTH_DEMO, 31, 1.5)
    M, s = ray_amd.ScatterForwardOperator(c, np.zeros((0, 9), order="F"), 1.5, 5e-2, ctx=ctx)
    assert M.shape == (0, 31 * 31) and s.shape == (0, 9)
    with pytest.raises(ray_amd.RayError):
        ray_amd.ScatterForwardOperator(ray_amd.SinProd.smooth_demo(), np.zeros((1, 9), order="F"), 1.5, 5e-2, ctx=ctx)
    with pytest.raises(ray_amd.RayError):
        ray_amd.ScatterForwardOperator(c, np.ones((2, 9), order="F"), 1.5, 0.0, ctx=ctx)


def test_forward_operator_after_update(ctx, orc):
    """the hess records follow ray_medium_update_grid (one medium object across iterates,
    NonObstacle.jl:333 inside the reconstruction loop)."""
    import ray_amd
    c, tab, m = _data(orc, 41, 3, 11, 1)
    med = ray_amd.GridMedium(c, 1.5, ctx)
    ray_amd.ScatterForwardOperator(med, m, 1.5, 5e-2, ctx=ctx)
    c2 = np.asfortranarray(1.0 + 0.5 * (c - 1.0))
    med.update(c2)
    M, s = ray_amd.ScatterForwardOperator(med, m, 1.5, 5e-2, ctx=ctx, droptol=None)
    Xo, _, Mo, _, _ = orc.forward_operator(orc.GridTables(c2, 1.5, 1), m, 5e-2)
    assert np.array_equal(s[:, 4:8], Xo)
    assert np.max(np.abs(M.toarray() - Mo)) <= MTOL * np.max(np.abs(Mo))


def test_forward_operator_golden(ctx, arith):
    """against tests/golden/frechet_n31_ns4_nd8.npz (made by oracle/gen_golden.py from the
    independent Python restatement; includes the cease branch)."""
    import ray_amd
    from conftest import golden
    g = golden("frechet_n31_ns4_nd8.npz")
    N = int(g["N"])
    m = np.zeros((g["rows"].shape[0], 9), order="F")
    m[:, 0:4] = g["rows"]
    m[:, 8] = g["T"]
    M, s, info = ray_amd.ScatterForwardOperator(np.asfortranarray(g["c"]), m, float(g["ext"]),
                                                float(g["trace_dt"]), ctx=ctx, info=True, droptol=None)
    Mg = np.zeros((4 * m.shape[0], N * N))
    Mg[g["M_rows"], g["M_cols"]] = g["M_vals"]
    if arith == 0:
        assert np.array_equal(s[:, 4:8], g["X"])
    else:
        assert np.max(np.abs(s[:, 4:8] - g["X"])) <= 1e-10
    assert np.max(np.abs(info["rho"] - g["rho"])) <= MTOL * np.max(np.abs(g["rho"]))
    assert np.max(np.abs(M.toarray() - Mg)) <= MTOL * np.max(np.abs(Mg))


def test_reconstruction_loop_passes(ctx, orc):
    """SURVEY.md section 8f NEXT-2 (NonObstacle.jl:340-367): dofs, b = M' mismatch over the selected rays,
    residual + fidelity update, on the device block rows against the numpy restatement in oracle/orc.py."""
    import ray_amd
    N = 61
    c, tab, m = _data(orc, N, 6, 24, 1)
    c0 = np.asfortranarray(1.0 + 0.6 * (c - 1.0))                  # the iterate differs from the truth
    F, X, rho, info = ray_amd.forward_operator_dev(c0, m, 1.5, 5e-2, ctx=ctx)
    M = F.tocsr()
    num = m.shape[0]
    mismatch = (m[:, 4:8] - X).reshape(-1)                        # (:336) ray-major
    rng = np.random.default_rng(5)
    fidelity = rng.uniform(0, 1, N * N) * (rng.uniform(0, 1, N * N) < 0.5)
    # dofs (:340-342)
    d_ref = orc.loop_dofs(M, fidelity)
    d = F.dofs(fidelity)
    assert np.max(np.abs(d - d_ref)) <= 1e-12 * np.max(np.abs(d_ref))
    # order / gradient (:345-354)
    mask, rows = orc.loop_order(d_ref, np.median(d_ref))
    assert 0 < mask.sum() < num
    b_ref = orc.loop_gradient(M, mismatch, rows)
    b = F.rmatvec(mismatch, mask)
    assert np.max(np.abs(b - b_ref)) <= 1e-12 * np.max(np.abs(b_ref))
    b_all = F.rmatvec(mismatch)
    assert np.max(np.abs(b_all - M.T @ mismatch)) <= 1e-12 * np.max(np.abs(b_all))
    # M x
    x = rng.normal(size=N * N) * 1e-2
    assert np.max(np.abs(F.matvec(x) - M @ x)) <= 1e-12 * np.max(np.abs(M @ x))
    # residual + fidelity (:359-367): thresholds chosen so that some rows pass and some do not
    res_all = np.abs(M @ x - mismatch)
    rejection = float(np.median(res_all[rows]))
    fid_ref, res_ref = orc.loop_fidelity(M, x, mismatch, rows, fidelity, rejection, 0.5)
    fid, res = F.fidelity_update(x, mismatch, fidelity, rejection, 0.5, mask)
    assert np.max(np.abs(res[rows] - res_ref)) <= 1e-12 * np.max(res_ref)
    unsel = np.setdiff1d(np.arange(4 * num), rows)
    assert not res[unsel].any()
    # rows within rounding of the threshold may flip; none here by construction of the margin check
    margin = np.min(np.abs(res_ref - rejection))
    assert margin > 1e-12
    assert np.max(np.abs(fid - fid_ref)) <= 1e-12
    # and exactly the max over the accepted rows of 1 - decay * (the device's own residual): max is order-independent
    fid_own = fidelity.copy()
    for k in rows[res[rows] < rejection]:
        cols = M.indices[M.indptr[k]:M.indptr[k + 1]]
        fid_own[cols] = np.maximum(fid_own[cols], 1 - 0.5 * res[k])
    assert np.array_equal(fid, fid_own)
    assert (fid != fidelity).any()
    F.close()


class tuned:
    """RAY_OPT_TUNE_* options of the context for the length of a `with` block."""

    def __init__(self, ctx, **kv):
        from ray_amd import _lib
        self.ctx, self.kv, self.lib = ctx, kv, _lib
        self.defaults = {"FRECHET_SLAB_MB": 0, "FRECHET_LDS_ROWS": -1, "MERGE_GRID": 0, "SORT_KEY_BITS": 0, "MERGE_ENTRIES": 0}

    def __enter__(self):
        for k, v in self.kv.items():
            self.ctx.set_option(getattr(self.lib, "OPT_TUNE_" + k), v)

    def __exit__(self, *exc):
        for k in self.kv:
            self.ctx.set_option(getattr(self.lib, "OPT_TUNE_" + k), self.defaults[k])


def test_forward_operator_chunked_equals_single(ctx, orc):
    """a slab budget of 1 MB forces several ray chunks: the concatenated CSR is identical."""
    import ray_amd
    c, tab, m = _data(orc, 61, 12, 40, 1)                         # 480 rays, 8 waves
    med = ray_amd.GridMedium(c, 1.5, ctx)
    M1, s1, i1 = ray_amd.ScatterForwardOperator(med, m, 1.5, 2e-2, ctx=ctx, info=True)
    with tuned(ctx, FRECHET_SLAB_MB=1):
        M2, s2, i2 = ray_amd.ScatterForwardOperator(med, m, 1.5, 2e-2, ctx=ctx, info=True)
    assert np.array_equal(M1.indptr, M2.indptr) and np.array_equal(M1.indices, M2.indices)
    assert np.array_equal(M1.data, M2.data)
    assert np.array_equal(s1, s2) and np.array_equal(i1["rho"], i2["rho"])
    assert np.array_equal(i1["nsteps"], i2["nsteps"]) and np.array_equal(i1["status"], i2["status"])


@pytest.mark.parametrize("lds_rows", [0, 20])
def test_forward_operator_merge_paths_agree(ctx, orc, lds_rows):
    """k_merge_emit keeps a ray's records and sorted entries in LDS when they fit and reads them in place
    otherwise (rays of more than 224 steps): identical matrix either way (RAY_OPT_TUNE_FRECHET_LDS_ROWS caps the LDS form;
    at 20 the rays of this case split between the two forms); block rows against the exported CSR."""
    import ray_amd
    import hiprt
    c, tab, m = _data(orc, 61, 6, 40, 1, bump=0.3)
    med = ray_amd.GridMedium(c, 1.5, ctx)
    M1, s1, i1 = ray_amd.ScatterForwardOperator(med, m, 1.5, 2e-2, ctx=ctx, info=True)
    assert i1["nsteps"].min() < 19 < i1["nsteps"].max()
    with tuned(ctx, FRECHET_LDS_ROWS=lds_rows, MERGE_GRID=7 if lds_rows else 0):   # (7: seven workgroups looping over the 240 rays)
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, 1.5, 2e-2, ctx=ctx, droptol=1e-12)
    M2 = F.tocsr()
    assert np.array_equal(M1.indptr, M2.indptr) and np.array_equal(M1.indices, M2.indices)
    assert np.array_equal(M1.data, M2.data)
    # the device-resident block rows hold the same matrix: per ray the union of its four rows' columns
    n, uoff_p, ucols_p, uvals_p = F.block_rows_dev()
    num = m.shape[0]
    uoff = hiprt.download(uoff_p, num + 1, np.int64)
    ucols = hiprt.download(ucols_p, n, np.int32)
    uvals = hiprt.download(uvals_p, 4 * n, np.float64).reshape(n, 4)
    assert uoff[0] == 0 and uoff[-1] == n and np.all(np.diff(uoff) >= 0)
    D = M2.toarray()
    for j in (0, 7, num // 2, num - 1):
        cols = ucols[uoff[j]:uoff[j + 1]]
        assert np.all(np.diff(cols) > 0)
        vals = uvals[uoff[j]:uoff[j + 1]]
        kept = np.where(np.abs(vals) > 1e-12, vals, 0.0)
        blk = np.zeros((4, D.shape[1]))
        blk[:, cols] = kept.T
        assert np.array_equal(blk, D[4 * j:4 * j + 4])
    F.close()


@pytest.mark.parametrize("N,ns,nd,dt", [(61, 6, 40, 2e-2), (75, 3, 70, 2e-2)])
def test_forward_operator_sort_key_widths_agree(ctx, orc, N, ns, nd, dt):
    """k_merge_sort runs on 32-bit keys when node and emission index fit together and on 64-bit keys
    otherwise: identical CSR either way (RAY_OPT_TUNE_SORT_KEY_BITS = 64 forces the wide keys)."""
    import ray_amd
    c, tab, m = _data(orc, N, ns, nd, 1, bump=0.3)
    med = ray_amd.GridMedium(c, 1.5, ctx)
    M1, s1, i1 = ray_amd.ScatterForwardOperator(med, m, 1.5, dt, ctx=ctx, info=True, droptol=None)
    with tuned(ctx, SORT_KEY_BITS=64):
        M2, s2, i2 = ray_amd.ScatterForwardOperator(med, m, 1.5, dt, ctx=ctx, info=True, droptol=None)
    assert np.array_equal(M1.indptr, M2.indptr) and np.array_equal(M1.indices, M2.indices)
    assert np.array_equal(M1.data, M2.data) and np.array_equal(s1, s2)
    Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, dt)
    assert np.max(np.abs(M1.toarray() - Mo)) <= MTOL * np.max(np.abs(Mo))


def test_forward_operator_is_the_derivative_at_scale(ctx, orc):
    """Size-independent property at grids the dense oracle cannot hold (N = 257: 32-bit sort keys; N = 1024,
    BASELINE config 5's grid: 64-bit keys): (s(c + eps d) - s(c - eps d)) / 2 eps -> M d, with the O(h)
    defect of the reference's operator (tests/test_oracle.py::test_forward_operator_is_the_derivative)
    shrinking under refinement."""
    import ray_amd
    defect = []
    for N in (257, 1024):
        c = orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.5)
        m, *_ = orc.scatter_relation(orc.SMOOTH_DEMO, 16, 48, 1e-2, nthreads=8)
        m = np.asfortranarray(m)
        m[:, 8] *= 0.8                                   # stay inside the disc: no cease branch
        x = np.linspace(-1.5, 1.5, N)
        d = np.asfortranarray(np.exp(-((x[:, None] - 0.1) ** 2 + (x[None, :] + 0.2) ** 2) / 0.18))
        med = ray_amd.GridMedium(c, 1.5, ctx)
        F, X, rho, info = ray_amd.forward_operator_dev(med, m, 1.5, 2e-2, ctx=ctx)
        assert not info["status"].any()
        lin = F.matvec(d.reshape(-1, order="F"))
        nnz = F.tocsr().nnz
        F.close()
        eps = 1e-6
        med.update(np.asfortranarray(c + eps * d))
        Xp, _ = ray_amd._trace(med, m, 1.5, 2e-2, ray_amd.TRACE_SHORTEN_CEASE, ctx, True)
        med.update(np.asfortranarray(c - eps * d))
        Xm, _ = ray_amd._trace(med, m, 1.5, 2e-2, ray_amd.TRACE_SHORTEN_CEASE, ctx, True)
        fd = ((Xp - Xm) / (2 * eps)).reshape(-1)
        defect.append(np.linalg.norm(lin - fd) / np.linalg.norm(fd))
        assert nnz > 4 * m.shape[0] * 50
    assert defect[0] < 0.02 and defect[1] < 0.6 * defect[0], defect


def test_workspaces_are_reused_and_released(ctx, orc):
    """a second identical call reuses the context's workspaces and the closed handle's CSR arrays (same
    device addresses); release_workspaces gives them back and the next call still works."""
    import ray_amd
    c, tab, m = _data(orc, 41, 4, 16, 1)
    med = ray_amd.GridMedium(c, 1.5, ctx)
    F1, X1, *_ = ray_amd.forward_operator_dev(med, m, 1.5, 5e-2, ctx=ctx)
    a1 = F1.device_arrays()
    M1 = F1.tocsr()
    F1.close()
    F2, X2, *_ = ray_amd.forward_operator_dev(med, m, 1.5, 5e-2, ctx=ctx)
    a2 = F2.device_arrays()
    assert tuple(a1) == tuple(a2)
    M2 = F2.tocsr()
    F2.close()
    assert np.array_equal(M1.data, M2.data) and np.array_equal(M1.indices, M2.indices) and np.array_equal(X1, X2)
    ctx.release_workspaces()
    F3, X3, *_ = ray_amd.forward_operator_dev(med, m, 1.5, 5e-2, ctx=ctx)
    assert np.array_equal(F3.tocsr().data, M1.data) and np.array_equal(X3, X1)
    F3.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_forward_operator_random_media(ctx, orc, arith, seed):
    """Randomised parity of the complete operator: random smooth medium, grid size, extent, step, rays started
    inside the disk with random directions and random travel times (some long enough to cease)."""
    import ray_amd
    rng = np.random.default_rng(seed)
    N = int(rng.integers(14, 44))
    ext = float(rng.uniform(1.3, 1.8))
    dt = float(rng.choice([2e-2, 5e-2]))
    p = np.linspace(-ext, ext, N)
    c = np.ones((N, N))
    for _ in range(3):
        a, kx, ky, ph = rng.uniform(0.03, 0.1), rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5), rng.uniform(0, 6.28)
        c += a * np.sin(kx * p[:, None] + ph) * np.cos(ky * p[None, :] - ph)
    c = np.asfortranarray(c)
    num = int(rng.integers(20, 90))
    r, th, ph = 0.8 * np.sqrt(rng.uniform(0, 1, num)), rng.uniform(0, 2 * np.pi, num), rng.uniform(0, 2 * np.pi, num)
    m = np.zeros((num, 9), order="F")
    m[:, 0], m[:, 1] = r * np.cos(th), r * np.sin(th)
    m[:, 2], m[:, 3] = np.cos(ph), np.sin(ph)
    m[:, 8] = rng.uniform(0.0, 2.2, num)
    tab = orc.GridTables(c, ext, arith)
    Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, dt)
    med = ray_amd.GridMedium(c, ext, ctx)
    M, s, info = ray_amd.ScatterForwardOperator(med, m, ext, dt, ctx=ctx, info=True, droptol=None)
    assert np.array_equal(info["status"], sto) and np.array_equal(info["nsteps"], nso)
    assert np.array_equal(s[:, 4:8], Xo)
    assert np.max(np.abs(info["rho"] - rho_o)) <= MTOL * np.max(np.abs(rho_o))
    assert np.max(np.abs(M.toarray() - Mo)) <= MTOL * np.max(np.abs(Mo))


def test_forward_operator_n513_entry_for_entry(ctx, orc, arith):
    """The Frechet matrix at the grid size the build is timed on (N = 513, dt = 1e-2: the 32-bit sort keys, ~130 steps
    for the longest rays), entry for entry against the C oracle's dense rows: 64 rays of 2 sources x 32 directions, from
    grazing chords of ~20 steps to diameters of ~200 (8.4 MB of dense oracle rows per ray)."""
    import ray_amd
    import scipy.sparse as sps
    N, dt = 513, 1e-2
    c, tab, m = _data(orc, N, 2, 32, arith)
    Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, dt, nthreads=16)
    assert nso.min() <= 25 and nso.max() >= 120
    med = ray_amd.GridMedium(c, 1.5, ctx)
    M, s, info = ray_amd.ScatterForwardOperator(med, m, 1.5, dt, ctx=ctx, info=True, droptol=None)
    assert np.array_equal(info["nsteps"], nso) and np.array_equal(info["status"], sto)
    assert np.array_equal(s[:, 4:8], Xo)
    assert np.max(np.abs(info["rho"] - rho_o)) <= MTOL * np.max(np.abs(rho_o))
    scale = float(np.max(np.abs(Mo)))
    Mo = sps.csr_matrix(Mo)                                        # the dense rows are 540 MB: compare sparse
    D = (M - Mo).tocoo()
    assert (np.max(np.abs(D.data)) if D.nnz else 0.0) <= MTOL * scale
    # every stencil node the oracle touched is a stored entry of the device matrix (droptol = None keeps exact zeros too)
    Mo.eliminate_zeros()
    pat_o = sps.csr_matrix((np.ones(Mo.nnz), Mo.indices, Mo.indptr), shape=Mo.shape)
    pat_m = sps.csr_matrix((np.ones(M.nnz), M.indices, M.indptr), shape=M.shape)
    assert M.nnz >= Mo.nnz and pat_o.multiply(pat_m).nnz == Mo.nnz
    # with the reference's droptol (1e-12, NonObstacle.jl:242) the stored pattern is the oracle's filtered one
    Md, _ = ray_amd.ScatterForwardOperator(med, m, 1.5, dt, ctx=ctx)
    keep = Mo.copy()
    keep.data[np.abs(keep.data) <= 1e-12] = 0.0
    keep.eliminate_zeros()
    pat_d = sps.csr_matrix((np.ones(Md.nnz), Md.indices, Md.indptr), shape=Md.shape)
    pat_k = sps.csr_matrix((np.ones(keep.nnz), keep.indices, keep.indptr), shape=keep.shape)
    assert abs(pat_d - pat_k).sum() <= 1e-5 * keep.nnz              # entries within rounding of the threshold may differ


def test_config5_grid_operator_and_normal_passes_vs_oracle(ctx, orc):
    """BASELINE config 5's grid (N = 1024, dt = 1e-2 as scripts/gpu_c5.sh runs it), against the ORACLE rather than as a
    property: 16 rays (2 sources x 8 directions, grazing to diameter; 537 MB of dense oracle rows) -- end states bit for
    bit, rho and M entry for entry, and the matrix-free loop's device passes y = M x, b = M'v, w = M'(M x), diag(M'M)
    against the same products formed on the host from the oracle's matrix (1e-12: order-free sums)."""
    import ray_amd
    import scipy.sparse as sps
    N, dt = 1024, 1e-2
    c, tab, m = _data(orc, N, 2, 8, 1)
    Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, dt, nthreads=16)
    assert nso.max() >= 120
    scale = float(np.max(np.abs(Mo)))
    Mo = sps.csr_matrix(Mo)
    med = ray_amd.GridMedium(c, 1.5, ctx)
    F, X, rho, info = ray_amd.forward_operator_dev(med, m, 1.5, dt, ctx=ctx, droptol=None)
    try:
        assert np.array_equal(info["nsteps"], nso) and np.array_equal(info["status"], sto)
        assert np.array_equal(X, Xo)
        assert np.max(np.abs(rho - rho_o)) <= MTOL * np.max(np.abs(rho_o))
        D = (F.tocsr() - Mo).tocoo()
        assert (np.max(np.abs(D.data)) if D.nnz else 0.0) <= MTOL * scale
        rng = np.random.default_rng(5)
        x = rng.standard_normal(N * N) * 1e-2
        v = rng.standard_normal(4 * m.shape[0])
        y = Mo @ x
        ys = float(np.max(np.abs(y)))
        assert np.max(np.abs(F.matvec(x) - y)) <= 1e-12 * ys
        b = Mo.T @ v
        assert np.max(np.abs(F.rmatvec(v) - b)) <= 1e-12 * float(np.max(np.abs(b)))
        w = Mo.T @ y
        assert np.max(np.abs(F.normal_matvec(x) - w)) <= 1e-12 * float(np.max(np.abs(w)))
        d = np.asarray(Mo.multiply(Mo).sum(axis=0)).ravel()
        assert np.max(np.abs(F.normal_diag() - d)) <= 1e-12 * float(np.max(d))
    finally:
        F.close()


def test_merge_by_segments_equals_merge_by_entries(ctx, orc):
    """The merge sorts the four stencil SEGMENTS of every record (k_seg_sort / k_seg_emit); the round-2 form that sorts the
    twelve entries (RAY_OPT_TUNE_MERGE_ENTRIES = 1: k_merge_sort / k_merge_emit) is kept as a cross-check: same pattern,
    same values up to the order of the sums -- short rays, long rays read in place, wide keys, and the oracle beside both."""
    import ray_amd
    for N, ns, nd, dt, bump, lds in ((61, 6, 40, 2e-2, 0.3, -1), (31, 3, 8, 2e-3, 0.0, -1), (75, 3, 70, 2e-2, 0.3, 20)):
        c, tab, m = _data(orc, N, ns, nd, 1, bump=bump)
        med = ray_amd.GridMedium(c, 1.5, ctx)
        with tuned(ctx, FRECHET_LDS_ROWS=lds):
            M1, s1, i1 = ray_amd.ScatterForwardOperator(med, m, 1.5, dt, ctx=ctx, info=True, droptol=None)
            with tuned(ctx, MERGE_ENTRIES=1):
                M2, s2, i2 = ray_amd.ScatterForwardOperator(med, m, 1.5, dt, ctx=ctx, info=True, droptol=None)
            with tuned(ctx, SORT_KEY_BITS=64):
                M3, s3, i3 = ray_amd.ScatterForwardOperator(med, m, 1.5, dt, ctx=ctx, info=True, droptol=None)
        assert np.array_equal(M1.indptr, M2.indptr) and np.array_equal(M1.indices, M2.indices) and np.array_equal(s1, s2)
        assert np.max(np.abs(M1.data - M2.data)) <= 1e-13 * np.max(np.abs(M2.data))
        assert np.array_equal(M1.indptr, M3.indptr) and np.array_equal(M1.indices, M3.indices) and np.array_equal(M1.data, M3.data)
        Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, dt)
        assert np.max(np.abs(M1.toarray() - Mo)) <= MTOL * np.max(np.abs(Mo))


@pytest.mark.parametrize("entries", [0, 1])
def test_forward_operator_on_the_grid_rim(ctx, orc, arith, entries):
    """ext = 1: the unit circle touches the grid's rim, rays start in the cells i = 0 / j = 0 whose stencils reach one node
    outside the grid (I - 1 < 1: the reference's sparse() would throw there).  Oracle and kernels drop those entries and flag
    the ray: same matrix, same flags -- the merge's clipped and dead segments."""
    import ray_amd
    N, ns, nd, dt = 21, 16, 6, 2e-2
    c = orc.sinprod_grid(orc.SMOOTH_DEMO, N, 1.0)
    tab = orc.GridTables(c, 1.0, arith)
    m, *_ = orc.scatter_relation(orc.SMOOTH_DEMO, ns, nd, 1e-2)
    m = np.asfortranarray(m)
    Xo, rho_o, Mo, sto, nso = orc.forward_operator(tab, m, dt)
    assert (sto != 0).any()
    med = ray_amd.GridMedium(c, 1.0, ctx)
    ctx.set_arith(arith)
    try:
        with tuned(ctx, MERGE_ENTRIES=entries):
            M, s, info = ray_amd.ScatterForwardOperator(med, m, 1.0, dt, ctx=ctx, info=True, droptol=None)
    finally:
        ctx.set_arith(ray_amd.ARITH_FUSED)
    assert np.array_equal(info["nsteps"], nso) and np.array_equal(info["status"], sto)
    assert np.array_equal(s[:, 4:8], Xo)
    assert np.max(np.abs(M.toarray() - Mo)) <= MTOL * np.max(np.abs(Mo))
