"""Multi-GPU form of the sweep: one process per GPU (torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests), rays sharded by source-angle (theta) block.

The reference spreads rays over `addprocs(3)` workers through a SharedArray
(NonObstacleDemo.jl:1; NonObstacle.jl:58-67,223-240) and then sums the misfit serially on the
master (NonObstacle.jl:336,372).  Here each rank shoots a contiguous block of sources -- rows
[src_begin*nd, src_end*nd) of m, a contiguous slab of every column -- and the ONLY exchange per
reconstruction iteration is one all-reduce of {sum mismatch^2, sum target^2} (plus ray / ray-step
counters).  No ray data moves between GPUs.  With the Frechet path on the device the same single
collective also carries the gradient b = M' mismatch (length N^2, NonObstacle.jl:354) and, when the
caller forms them per rank, the normal-matrix contributions: allreduce_loop_sums packs everything
into one buffer so that it stays ONE all-reduce per iteration.
"""
import math


def theta_block(ns, rank, world):
    """Sources [begin, end) owned by `rank`: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(ns, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def theta_comb(ns, rank, world):
    """Comb sharding: `rank` owns the sources rank, rank + world, ... < ns -- (src_begin, src_end, stride, count) for
    ray_amd's RAY_OPT_SOURCE_STRIDE.  Every rank sweeps the whole source circle at 1/world of its density, so the
    ranks' ray-step counts agree closely whatever the medium's asymmetry (contiguous theta-blocks of the shipped
    medium differ by 16 %, which a max-over-ranks step time pays in full)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank, ns, world, (ns - rank + world - 1) // world


def allreduce_misfit(sums, group=None):
    """In-place SUM all-reduce of a 2-element float64 tensor {sum mismatch^2, sum target^2}
    (device tensor under nccl, CPU tensor under gloo).  Returns the global objective
    sqrt(sum mismatch^2 / sum target^2) (NonObstacle.jl:372)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    s = sums.tolist()
    return math.sqrt(s[0] / s[1]) if s[1] > 0 else float("nan")


def sharded_scatter_relation(shoot_block, ns, nd, rank, world):
    """Shoot this rank's theta-block.  `shoot_block(src_begin, src_end)` returns the (rows, 9)
    array for those sources (ray_amd.ScatterRelation(..., src_begin=, src_end=) on a GPU).
    Returns (m_local, row_begin, row_end): the rank's slab of the global m."""
    b, e = theta_block(ns, rank, world)
    m = shoot_block(b, e)
    if m.shape[0] != (e - b) * nd:
        raise ValueError("block has the wrong number of rows")
    return m, b * nd, e * nd


def allreduce_loop_sums(sums, *more, group=None):
    """ONE SUM all-reduce for everything a reconstruction iteration adds up over rays: `sums`
    ({sum mismatch^2, sum target^2}) and any further float64 tensors of rank-local partial sums on the
    same device (the gradient b = M[order,:]' mismatch[order], the normal matrix M'M over the rank's
    rays, ...).  The tensors are updated in place; returns the global objective (NonObstacle.jl:372).
    (The fidelity update is a MAX over rays, not a sum: ranks that update fidelity locally combine it
    with allreduce_fidelity.)"""
    import torch
    import torch.distributed as dist
    parts = [sums, *more]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if len(parts) == 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        else:
            flat = torch.cat([p.reshape(-1) for p in parts])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            off = 0
            for p_ in parts:
                n = p_.numel()
                p_.copy_(flat[off:off + n].reshape(p_.shape))
                off += n
    s = sums.tolist()
    return math.sqrt(s[0] / s[1]) if s[1] > 0 else float("nan")


def allreduce_fidelity(fidelity, group=None):
    """In-place MAX all-reduce of the fidelity vector (NonObstacle.jl:365 is a running max over rays,
    so per-rank results combine exactly)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(fidelity, op=dist.ReduceOp.MAX, group=group)
    return fidelity
