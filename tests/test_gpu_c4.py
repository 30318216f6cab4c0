"""BASELINE config 4 on one GPU: theta-blocks of the 8192 x 8192 ray sweep on the 512^2 grid -- one rank's share
(1024 sources x 8192 directions = 8 388 608 rays, 604 MB of m) of the 8-GPU job.  No 8-GPU node is needed to run
a shard: checked here are closed-form chords (c == 1), event status, counters, an oracle sub-sweep bit for bit in both
arithmetic flavours, the whole block against the oracle from (ns, nd) bit for bit, and that a block's rows do not
depend on which block call produced them (shards abut exactly)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NS = ND = 8192
N, EXT, DT = 512, 1.5, 1e-2
BLOCK = NS // 8


def test_c4_theta_block_constant_medium(ctx, orc):
    """K1 on the first rank's block: straight chords, exit angle s + 2 phi, p unchanged, t = 2 sin(phi)."""
    import ray_amd
    med1 = ray_amd.GridMedium(np.ones((N, N), order="F"), EXT, ctx)
    ctx.reset_counters()
    m, info = ray_amd.ScatterRelation(med1, None, NS, ND, DT, ctx=ctx, src_begin=0, src_end=BLOCK, info=True)
    cnt = ctx.counters()
    assert m.shape == (BLOCK * ND, 9)
    assert cnt["rays"] == BLOCK * ND and cnt["ray_steps"] == int(info["nsteps"].astype(np.int64).sum())
    assert np.all(info["status"] == ray_amd.ST_EXITED)
    src = np.repeat(orc.linspace(0, 2 * math.pi, NS + 1)[:BLOCK], ND)
    phi = np.tile(orc.linspace(0, math.pi, ND + 2)[1:ND + 1], BLOCK)
    assert np.max(np.abs(m[:, 8] - 2 * np.sin(phi))) < 1e-12
    assert np.max(np.abs(m[:, 4] - np.cos(src + 2 * phi))) < 1e-12
    assert np.max(np.abs(m[:, 5] - np.sin(src + 2 * phi))) < 1e-12
    assert np.max(np.abs(m[:, 6:8] - m[:, 2:4])) < 1e-13


def test_c4_theta_blocks_smooth_medium(ctx, orc):
    import ray_amd
    c = orc.sinprod_grid(orc.SMOOTH_DEMO, N, EXT)
    med = ray_amd.GridMedium(c, EXT, ctx)
    # rank 3's block [3072, 4096) and a call that straddles it, [3584, 4608): common rows must be identical
    a, ainfo = ray_amd.ScatterRelation(med, None, NS, ND, DT, ctx=ctx, src_begin=3 * BLOCK, src_end=4 * BLOCK, info=True)
    assert np.all(ainfo["status"] == ray_amd.ST_EXITED)
    b = ray_amd.ScatterRelation(med, None, NS, ND, DT, ctx=ctx, src_begin=3 * BLOCK + BLOCK // 2, src_end=4 * BLOCK + BLOCK // 2)
    half = (BLOCK // 2) * ND
    assert np.array_equal(a[half:], b[:half])
    del b
    # oracle sub-sweep, bit for bit from the same states: every 61st ray of the block (137 519 rays)
    for arith in (ray_amd.ARITH_EXACT, ray_amd.ARITH_FUSED):
        ctx.set_arith(arith)
        try:
            idx = np.arange(0, a.shape[0], 61)
            sub = np.asfortranarray(a[idx].copy())
            sub[:, 4:] = 0
            ginfo = ray_amd.shoot_states(med, sub, DT, ctx=ctx)
            ref = np.asfortranarray(a[idx].copy())
            ref[:, 4:] = 0
            st, nst, _ = orc.shoot(orc.GridTables(c, EXT, arith), ref, DT, nthreads=16)
            assert np.array_equal(ginfo["status"], st) and np.array_equal(ginfo["nsteps"], nst)
            assert np.array_equal(sub, ref)
            if arith == ray_amd.ARITH_FUSED:       # the block call itself ran fused: its rows are these rows
                assert np.array_equal(a[idx], ref) and np.array_equal(ainfo["nsteps"][idx], nst)
        finally:
            ctx.set_arith(ray_amd.ARITH_FUSED)
    # the library's own initial states are the host-libm oracle's to the last-bit difference of two math libraries ...
    init = orc.init_states(orc.GridTables(c, EXT), NS, ND, src_begin=3 * BLOCK, src_end=3 * BLOCK + 2)
    assert np.max(np.abs(a[:2 * ND, 0:4] - init[:, 0:4])) < 1e-14
    # ... and under the oracle's reproducible sin/cos flavour the WHOLE block call is the oracle's, from (ns, nd):
    # all 8 388 608 rays (1.1e9 ray-steps, ~8 s of CPU on the box's 16 threads), every bit, status and step count
    ref, st, nst, _ = orc.scatter_relation(orc.GridTables(c, EXT, 1), NS, ND, DT, src_begin=3 * BLOCK, src_end=4 * BLOCK,
                                           nthreads=16, trig=1)
    assert np.array_equal(ainfo["status"], st) and np.array_equal(ainfo["nsteps"], nst)
    assert np.array_equal(a, ref)
