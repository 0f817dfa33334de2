"""N > 1 path on CPU: two gloo ranks shard the channels of the spectral workload, reduce their own channels
per bin (oracle Analyzer in place of the GPU bank) and all-reduce the partial sums; the result must equal the
unsharded reduction."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, channels, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import spectral as sp
    sharding = importlib.import_module("lsp-dsp-units_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(channels, rank, world)
    x = _signal(channels)[lo:hi]
    a = sp.Analyzer(hi - lo, 8, 48000, 1.0, 0)
    a.configure(sample_rate=48000, rate=100.0, rank=8, window_name="hann", reactivity=0.1, shift=1.0)
    frames = []
    for k in range(0, x.shape[1], 480):
        a.process(x[:, k:k + 480])
        frames.append(a.amp[:, :a.csize].sum(axis=0, dtype=np.float32))
    part = torch.from_numpy(np.stack(frames))
    sharding.allreduce_bins(part)                      # one collective for all frames
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), part.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _signal(channels):
    rng = np.random.default_rng(7)
    return rng.standard_normal((channels, 480 * 6)).astype(np.float32)


def test_shard_ranges_cover_everything():
    sharding = importlib.import_module("lsp-dsp-units_amd.sharding")
    for total in (1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_bin_reduction_matches_unsharded(tmp_path):
    import torch.multiprocessing as mp
    from oracle import spectral as sp
    channels, world = 10, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, channels, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "reduced.npy"))
    # unsharded reference: the staggered schedule depends on the channel count, so reduce per shard here too
    x = _signal(channels)
    ref = np.zeros_like(got)
    sharding = importlib.import_module("lsp-dsp-units_amd.sharding")
    for r in range(world):
        lo, hi = sharding.shard_range(channels, r, world)
        a = sp.Analyzer(hi - lo, 8, 48000, 1.0, 0)
        a.configure(sample_rate=48000, rate=100.0, rank=8, window_name="hann", reactivity=0.1, shift=1.0)
        for i, k in enumerate(range(0, x.shape[1], 480)):
            a.process(x[lo:hi, k:k + 480])
            ref[i] += a.amp[:, :a.csize].sum(axis=0, dtype=np.float32)
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)
