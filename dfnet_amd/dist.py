"""Multi-GPU sharding of a render_path batch: one process per GPU, NeRF-H weights replicated,
frames block-partitioned over ranks, ONE gather of rgb+disp(+per-frame errors) at the end (gather_frames_direct: a grouped
send / receive batch into rank 0's final tensors, RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests).

Replaces the serial `for i, c2w in enumerate(render_poses)` loop of
/root/reference/script/models/rendering.py:420-452 — the reference has no collective at all.
The path has no exchange step, so there is no data-path collective: only the result gather.
"""
import os

import torch
import torch.distributed as dist


def force_collectives():
    """DFN_FORCE_COLLECTIVES=1: run every collective of this module even in a process group of ONE rank (the early-outs at
    world size 1 are skipped and the group is initialised under a launcher that sets WORLD_SIZE=1) — how the one-GPU test box
    brings RCCL up and pushes real frames / gradients through gather and all-reduce (tests/test_gpu_dist.py)."""
    return os.environ.get("DFN_FORCE_COLLECTIVES", "0") == "1"


def active():
    """True when the collectives of this module actually run: an initialised group of more than one rank, or a forced one."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives())


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun) if WORLD_SIZE > 1.
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or (force_collectives() and "WORLD_SIZE" in os.environ)) and not dist.is_initialized():
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


def band_mode(n_frames, world):
    """SURVEY 8(e)'s small-batch fallback: fewer frames than ranks (the single validation frames of run_nerf.py:200,228; configs[0]'s
    four frames on eight GPUs) -> frames are split into row bands so that no rank idles."""
    return 0 < n_frames < world


def band_unit(n_frames, H, rank, world):
    """Row-band work unit of `rank` when band_mode(): the RANKS are block-partitioned over the frames (frame f is rendered by the
    ranks of frame_block(world, f, n_frames): world // n_frames of them, or one more), and the frame's H rows are block-partitioned
    over those ranks.  Returns (frame, row_lo, row_hi); a band of a [H, W, ...] frame is contiguous memory, so the end gather
    receives bands in place exactly as it receives frame blocks."""
    assert band_mode(n_frames, world)
    for f in range(n_frames):
        lo, hi = frame_block(world, f, n_frames)
        if lo <= rank < hi:
            r0, r1 = frame_block(H, rank - lo, hi - lo)
            return f, r0, r1
    raise AssertionError("every rank belongs to a frame")


def gather_bands_direct(tensors, n_frames, H, dst=0):
    """End gather of the row-band fallback: `tensors` = this rank's band of each output, shaped [rows, ...]; rank `dst` allocates
    the final [n_frames, H, ...] tensors and receives every peer's band straight into its view — the same single grouped
    send / receive batch as gather_frames_direct (one ncclGroup over xGMI on the nccl backend).  Returns the list of full tensors
    on rank dst, a list of None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    units = [band_unit(n_frames, H, r, world) for r in range(world)]
    ops, outs = [], [None] * len(tensors)
    if rank == dst:
        outs = [t.new_empty((n_frames, H) + tuple(t.shape[1:])) for t in tensors]
        for r, (f, r0, r1) in enumerate(units):
            if r == dst:
                for t, o in zip(tensors, outs):
                    o[f, r0:r1].copy_(t)
            elif r1 > r0:
                ops += [dist.P2POp(dist.irecv, o[f, r0:r1], r) for o in outs]
    elif tensors[0].shape[0] > 0:
        ops += [dist.P2POp(dist.isend, t.contiguous(), dst) for t in tensors]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return outs


def gather_frames(local, n_frames, dst=0):
    """Gather per-rank frame stacks (tensor [n_local, ...]) to rank `dst` in global frame order.

    One collective per call: ranks pad their block to the common maximum length, rank dst receives
    world buffers (a direct gather — every peer writes straight to the root, the right shape for
    point-to-point xGMI links; a ring would be bound by one link) and drops the padding.
    Returns the [n_frames, ...] tensor on rank dst, None elsewhere.  world == 1: identity."""
    if not active():
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


def root_buffers(tails, n_frames, device, dst=0):
    """Where a rank renders its frames so that the end gather needs no staging copy: rank `dst` allocates the FINAL
    [n_frames, ...] tensors and renders into its own block's views; every other rank gets plain local tensors.
    tails: per-frame shapes, e.g. [(H, W, 3), (H, W), ()].  Returns (outs | None, local tensors)."""
    rank, world = rank_world()
    lo, hi = frame_block(n_frames, rank, world)
    if rank == dst and active() and world > 1:
        outs = [torch.empty((n_frames,) + tuple(t), device=device) for t in tails]
        return outs, [o[lo:hi] for o in outs]
    return None, [torch.empty((hi - lo,) + tuple(t), device=device) for t in tails]


def gather_frames_direct(tensors, n_frames, dst=0, outs=None, extra=None):
    """The end gather as ONE grouped point-to-point exchange: rank `dst` posts a receive for every peer's block straight into
    views of the final [n_frames, ...] tensors (`outs`, root_buffers(); allocated here when None) and every peer sends its
    local tensors as they are — no padding to a common length, no packing, no concatenation: the root holds each frame once.
    On the nccl backend the batch is one ncclGroup of sends / receives (every peer writes over its own xGMI link to the root);
    gloo runs the same operations one by one.  `extra`: an optional small per-rank float tensor (range flags, timings) gathered
    in the same batch -> returned as [world, len] on rank dst.  Returns (list of full tensors | list of None, extras | None).
    A forced one-rank group (DFN_FORCE_COLLECTIVES) has no peer to send to: it runs the padded dist.gather of gather_frames so
    that the communicator is still exercised."""
    tensors = list(tensors)
    if not active():
        return tensors, (None if extra is None else extra[None])
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        full = [gather_frames(t, n_frames, dst) for t in tensors]
        return full, (None if extra is None else gather_frames(extra[None], 1, dst))
    sizes = [frame_block(n_frames, r, world) for r in range(world)]
    ops, ext = [], None
    if rank == dst:
        if outs is None:
            outs = [t.new_empty((n_frames,) + tuple(t.shape[1:])) for t in tensors]
        lo, hi = sizes[dst]
        if extra is not None:
            ext = extra.new_empty((world,) + tuple(extra.shape))
            ext[dst].copy_(extra)
        for t, o in zip(tensors, outs):
            if hi > lo and o[lo:hi].data_ptr() != t.data_ptr():
                o[lo:hi].copy_(t)
        for r, (a, b) in enumerate(sizes):   # per peer: its tensors in order, then its extras — the order the peer sends them in
            if r == dst:
                continue
            if b > a:
                ops += [dist.P2POp(dist.irecv, o[a:b], r) for o in outs]
            if extra is not None:
                ops.append(dist.P2POp(dist.irecv, ext[r], r))
    else:
        if tensors[0].shape[0] > 0:
            ops += [dist.P2POp(dist.isend, t.contiguous(), dst) for t in tensors]
        if extra is not None:
            ops.append(dist.P2POp(dist.isend, extra.contiguous(), dst))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return [None] * len(tensors), None
    return outs, ext


def gathered_bytes(tensors, n_frames, dst=0):
    """Bytes that cross links in gather_frames_direct(tensors): every frame not owned by rank dst."""
    rank, world = rank_world()
    lo, hi = frame_block(n_frames, dst, world)
    per_frame = sum(int(torch.Size(t.shape[1:]).numel()) * t.element_size() for t in tensors)
    return (n_frames - (hi - lo)) * per_frame


def all_gather_flags(value, device):
    """Every rank's integer flag word (the NeRF-H range-guard bits) on EVERY rank: list of `world` ints.  An int32 all-gather, not a
    float MAX: bits 1 and 2 from different ranks must OR, and every rank must see the same words so that all of them raise the same
    error after the collective instead of some returning into the caller's next collective.  world 1: [value]."""
    if not active():
        return [int(value)]
    v = torch.tensor([int(value)], dtype=torch.int32, device=device)
    out = [torch.empty_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(out, v)
    return [int(t.item()) for t in out]


def all_gather_floats(values, device):
    """[world, len(values)] float64 tensor of every rank's small record (timings, clocks); world 1: the record itself."""
    v = torch.tensor([list(values)], dtype=torch.float64, device=device)
    if not active():
        return v.cpu()
    out = [torch.empty_like(v[0]) for _ in range(dist.get_world_size())]
    dist.all_gather(out, v[0])
    return torch.stack(out).cpu()


def data_parallel_rounds(n_items, rank, world):
    """Round-robin schedule of n_items work items (training images) over `world` ranks that keeps EVERY rank on the same number
    of optimisation steps: items are dealt in rounds of `world`; yields, per item index, (mine, end_of_round, contributors).
    `mine`: this rank computes the item's gradients; `end_of_round`: every rank now joins ONE gradient all-reduce and steps;
    `contributors`: items in this round (< world in the last, partial round: ranks without an item contribute zero gradients
    and the sum is divided by `contributors`, allreduce_gradients(contributors=...)).  Without this, n_items % world != 0 left
    some ranks one all-reduce short per epoch: their next epoch's first all-reduce paired with the others' last."""
    for i in range(n_items):
        start = i - i % world
        contributors = min(world, n_items - start)
        yield i % world == rank, (i % world == world - 1) or (i == n_items - 1), contributors


def allreduce_gradients(params, contributors=None):
    """Data-parallel steps (the NeRF-H training loop): average the gradients over ranks with ONE
    all-reduce of a flat bucket (15.4 M parameters = 61.6 MB for DFNet; RCCL over xGMI with the nccl backend).
    Parameters without a gradient are skipped (the same set on every rank).  contributors: divide the sum by this many ranks
    instead of the world size (a partial last round: the idle ranks pass zeros).  world == 1: no-op."""
    if not active():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= (contributors if contributors else dist.get_world_size())
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def max_over_ranks(value, device):
    """MAX all-reduce of a python float (timing)."""
    if not active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if active():
        dist.barrier()


def sync_buffers(module, src=0):
    """Data-parallel training keeps parameters identical through the gradient all-reduce, but BatchNorm's running statistics
    are updated from each rank's OWN mini-batch: average them (and broadcast the integer step counters) before rank `src`
    writes a checkpoint, so that the saved model is the one every rank would evaluate.  world == 1: no-op."""
    if not active():
        return
    world = dist.get_world_size()
    for name, buf in module.named_buffers():
        if buf.is_floating_point():
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            buf /= world
        else:
            dist.broadcast(buf, src=src)
