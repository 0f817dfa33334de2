"""Channel sharding across the GPUs of a node.

Channels of every unit are independent, so rank r of W simply owns a contiguous channel range and runs its own
banks: no data-path collective.  The one exchange step of the hot path is the cross-channel per-bin reduction of
the spectral path (the MultiSpectralProcessor-style callback of BASELINE config 5): every rank reduces its own
channels on the device (mi_analyzer_bank_reduce_bins) and the partial sums are all-reduced -- RCCL over xGMI when
the tensors live on GPUs (torch.distributed backend "nccl"), gloo in the CPU tests.  The message is tiny
(2^(rank-1)+1 floats per frame), so frames are batched into one collective."""


def shard_range(total, rank, world):
    """Contiguous, balanced channel range [lo, hi) of `rank`; the first (total % world) ranks hold one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_bins(partial, group=None):
    """Sum per-bin partial sums over all ranks, in place. `partial`: torch tensor [frames, bins] (any device)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial
