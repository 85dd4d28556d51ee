"""
Multi-GPU layout of the hot path: streams are independent (each owns its leftover PCM and its
feature window, network weights are read-only), so the batch of streams is split across ranks --
one process per GPU -- with NO collective on the data path.  The only exchange is the gather of
the scalar probabilities (``torch.distributed`` all-gather: RCCL over xGMI on GPUs, gloo in the
CPU tests).  The reference has no counterpart (one stream per process,
/root/reference/precise/scripts/engine.py:53-63).
"""
import os


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process default)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def shard_bounds(n_streams: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of the global stream index space owned by ``rank``; the first
    ``n_streams % world`` ranks take one extra stream."""
    if not 0 <= rank < world:
        raise ValueError('rank %d outside world of %d' % (rank, world))
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_probabilities(local, n_streams: int, group=None, dst=None, force_collective: bool = False):
    """
    Gather per-rank probability blocks into global stream order.

    local: tensor [..., n_local] (e.g. [steps, n_local]) on this rank's device, where n_local is
    this rank's ``shard_bounds`` width.  Shards may be uneven; blocks are padded to the widest shard
    for the collective and trimmed afterwards.

    dst=None: all-gather, every rank returns [..., n_streams].
    dst=r:    gather to rank r only (RCCL send/recv group: the 7 peers of an 8-GPU node write to r over
              7 different xGMI links at once, ~3x cheaper than the ring all-gather for this payload);
              rank r returns [..., n_streams], the others None.
    force_collective: issue the collective even in a world of one rank (a 1-GPU box can then run the RCCL calls themselves:
              tests/rccl_one_rank_check.py); by default a single rank returns its block untouched.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        if local.shape[-1] != n_streams:
            raise ValueError('single-rank gather expects all %d streams, got %d' % (n_streams, local.shape[-1]))
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_streams, rank, world)
    if local.shape[-1] != hi - lo:
        raise ValueError('rank %d holds %d streams, expected %d' % (rank, local.shape[-1], hi - lo))
    widest = shard_bounds(n_streams, 0, world)[1]
    lead = tuple(local.shape[:-1])
    block = local.new_zeros(lead + (widest,))
    block[..., :hi - lo] = local
    if dst is None:
        flat = local.new_empty((world * block.numel(),))
        dist.all_gather_into_tensor(flat, block.contiguous().view(-1), group=group)
        gathered = flat.view((world,) + lead + (widest,))
    else:
        flat_in = block.contiguous().view(-1)
        if rank == dst:
            pieces = [local.new_empty((flat_in.numel(),)) for _ in range(world)]
            dist.gather(flat_in, pieces, dst=dst, group=group)
            gathered = torch.stack(pieces).view((world,) + lead + (widest,))
        else:
            dist.gather(flat_in, None, dst=dst, group=group)
            return None
    parts = []
    for r in range(world):
        rlo, rhi = shard_bounds(n_streams, r, world)
        parts.append(gathered[r][..., :rhi - rlo])
    return torch.cat(parts, dim=-1)
