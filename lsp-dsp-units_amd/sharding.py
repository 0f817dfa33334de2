"""Channel sharding across the GPUs of a node.

Channels of every unit are independent, so rank r of W simply owns a contiguous channel range and runs its own
banks: no data-path collective.  The one exchange step of the hot path is the cross-channel per-bin reduction of
the spectral path (the MultiSpectralProcessor-style callback of BASELINE config 5): every rank reduces its own
channels on the device (mi_analyzer_bank_reduce_bins, a shard-composable summation order) and the partial sums are
all-reduced by the LIBRARY: mi_analyzer_bank_allreduce_bins issues one ncclAllReduce (RCCL over xGMI) from its C++
host side on the caller's stream, through a communicator made with mi_dspu_comm_create.  `library_comm` below only
carries the 128-byte communicator id from rank 0 to the others over whatever process group the launcher set up.
The message is tiny (2^(rank-1)+1 floats per frame), so frames are batched into one collective.
`allreduce_bins` is the same sum through torch.distributed, for process groups without GPUs (the gloo tests)."""


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


def library_comm(mi, group=None):
    """The library's own RCCL communicator over the ranks of the (initialised) torch.distributed group: rank 0 draws the
    id (mi_dspu_comm_unique_id), the group broadcasts its 128 bytes, every rank joins (mi_dspu_comm_create).
    Returns None for a single rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return None
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # broadcast_object_list takes the GLOBAL rank of the source: the group's rank 0 draws the id and sends it
    src = dist.get_global_rank(group, 0) if group is not None else 0
    # Every rank must come out of this with the same answer (a communicator on all of them or on none): a rank that went on
    # with torch's collective while the others wait inside the library's would hang the job.
    box = [None]
    if rank == 0:
        try:
            box[0] = mi.Comm.unique_id()
        except Exception as e:                               # the others are told: nobody joins
            box[0] = None
            print("library_comm: no unique id (%s)" % e)
    dist.broadcast_object_list(box, src=src, group=group)
    if box[0] is None:
        return None
    import torch

    def flag_device():
        # (a group made with the default, multi-backend setting reports "cuda:nccl,cpu:gloo": anything with nccl in it carries
        # device tensors)
        backend = str(dist.get_backend(group)).lower()
        return torch.device("cuda", torch.cuda.current_device()) if ("nccl" in backend and torch.cuda.is_available()) else torch.device("cpu")

    def all_agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=flag_device())
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) != 0

    # 1. Everything a rank can find out WITHOUT joining -- the id arrived whole, the library has the entry points, a device is
    #    there -- is agreed on first: joining is a rendezvous (ncclCommInitRank), and a rank that raised before it would leave
    #    the others blocked inside it, never reaching the vote below.
    ready = True
    try:
        ready = isinstance(box[0], (bytes, bytearray)) and len(box[0]) == mi.Comm.ID_BYTES and mi.device_count() > 0 \
                and hasattr(mi.lib, "mi_dspu_comm_create")
    except Exception as e:
        ready = False
        print("library_comm: rank %d is not ready to join (%s)" % (rank, e))
    if not all_agree(ready):
        return None
    # 2. The rendezvous itself, with a deadline: a peer that died on the way leaves ncclCommInitRank waiting for ever, and a job
    #    that hangs is worse than one that ends -- the rank gives up with a non-zero status (MI_COMM_INIT_TIMEOUT seconds, 180).
    import os
    import threading
    result = {}

    # (HIP's current device is per thread and a new thread starts on device 0: the device the main thread works on is taken
    # here and selected again inside the thread -- ncclCommInitRank binds to the calling thread's device)
    dev = torch.cuda.current_device() if torch.cuda.is_available() else 0

    def join():
        try:
            if torch.cuda.is_available():
                torch.cuda.set_device(dev)
            mi.check(mi.lib.mi_dspu_set_device(dev))
            result["comm"] = mi.Comm(box[0], world, rank)
            result["device"] = dev
        except Exception as e:
            result["error"] = e
    th = threading.Thread(target=join, daemon=True)
    th.start()
    th.join(float(os.environ.get("MI_COMM_INIT_TIMEOUT", "180")))
    if th.is_alive():
        print("library_comm: rank %d: the communicator's rendezvous did not complete in time; giving up" % rank, flush=True)
        os._exit(3)
    comm, ok = result.get("comm"), 1
    if comm is None:
        ok = 0
        print("library_comm: rank %d could not join (%s)" % (rank, result.get("error")))
    # 3. A communicator on every rank or on none
    if not all_agree(ok):
        if comm is not None:
            comm.close()
        return None
    return comm
