"""Multi-GPU sharding of a render_path batch: one process per GPU, NeRF-H weights replicated,
frames block-partitioned over ranks, ONE gather of rgb+disp at the end (RCCL over xGMI when the
backend is "nccl"; "gloo" on CPU for tests).

Replaces the serial `for i, c2w in enumerate(render_poses)` loop of
/root/reference/script/models/rendering.py:420-452 — the reference has no collective at all.
The path has no exchange step, so there is no data-path collective: only the result gather.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun) if WORLD_SIZE > 1.
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def rank_world():
    """(rank, world) of the initialised process group, (0, 1) without one."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def frame_block(n_frames, rank, world):
    """Contiguous block [lo, hi) of frame indices owned by `rank` (sizes differ by at most 1;
    keeps PNG numbering trivial: global index = lo + local index)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_frames(local, n_frames, dst=0):
    """Gather per-rank frame stacks (tensor [n_local, ...]) to rank `dst` in global frame order.

    One collective per call: ranks pad their block to the common maximum length, rank dst receives
    world buffers (a direct gather — every peer writes straight to the root, the right shape for
    point-to-point xGMI links; a ring would be bound by one link) and drops the padding.
    Returns the [n_frames, ...] tensor on rank dst, None elsewhere.  world == 1: identity."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [frame_block(n_frames, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    pad = local.new_zeros((longest,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)


def allreduce_gradients(params):
    """Data-parallel DFNet_dm: average the pose regressor's gradients over ranks with ONE
    all-reduce of a flat bucket (15.4 M parameters = 61.6 MB for DFNet; RCCL over xGMI with the nccl backend).
    Parameters without a gradient are skipped (the same set on every rank).  world == 1: no-op."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def max_over_ranks(value, device):
    """MAX all-reduce of a python float (timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
