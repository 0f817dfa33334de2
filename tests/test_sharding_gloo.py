"""N > 1 path on CPU (gloo, world_size 2): the channels of the spectral workload are sharded with
sharding.shard_range, every rank sums the spectra of ITS channels per bin and the partial sums are all-reduced.

The reference is the UNSHARDED reduction: every channel's spectrum is computed on its own (one analyzer object per
channel, so nothing in it depends on how many channels share an object or a rank -- the GPU bank likewise analyses
every channel at the strobe) and all channels are summed in float64.  The sharded result must equal it to float32
summation round-off whatever the shard boundaries are, including an uneven split."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAME, FRAMES = 480, 6


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _signal(channels):
    rng = np.random.default_rng(7)
    return rng.standard_normal((channels, FRAME * FRAMES)).astype(np.float32)


def _channel_spectra(x):
    """[frames][bins] smoothed magnitudes of ONE channel (oracle Analyzer with a single channel: analysed at the strobe)."""
    from oracle import spectral as sp
    a = sp.Analyzer(1, 8, 48000, 1.0, 0)
    a.configure(sample_rate=48000, rate=100.0, rank=8, window_name="hann", reactivity=0.1, shift=1.0)
    out = []
    for k in range(0, x.size, FRAME):
        a.process(x[None, k:k + FRAME])
        out.append(a.amp[0, :a.csize].copy())
    return np.stack(out)


def _worker(rank, world, port, channels, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    sharding = importlib.import_module("lsp-dsp-units_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(channels, rank, world)
    x = _signal(channels)
    part = np.zeros((FRAMES, 129), np.float32)
    for c in range(lo, hi):                                 # this rank's channels only
        part += _channel_spectra(x[c])
    t = torch.from_numpy(part)
    sharding.allreduce_bins(t)                              # one collective for all frames
    np.save(os.path.join(out_dir, "reduced_rank%d.npy" % rank), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    sharding = importlib.import_module("lsp-dsp-units_amd.sharding")
    for total in (1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("channels", [10, 7])               # an even and an uneven split over two ranks
def test_two_rank_bin_reduction_matches_unsharded(tmp_path, channels):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), channels, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "reduced_rank%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(got[0], got[1])           # every rank holds the same sum after the all-reduce
    x = _signal(channels)
    ref = np.zeros((FRAMES, 129), np.float64)
    for c in range(channels):                               # no shard boundaries anywhere in the reference
        ref += _channel_spectra(x[c])
    assert np.abs(ref).max() > 1.0
    np.testing.assert_allclose(got[0], ref, rtol=2e-6, atol=2e-6 * np.abs(ref).max())


def test_library_comm_is_none_for_one_rank():
    """sharding.library_comm only builds a communicator when a process group with more than one rank is up."""
    sharding = importlib.import_module("lsp-dsp-units_amd.sharding")
    mi = importlib.import_module("lsp-dsp-units_amd")
    assert sharding.library_comm(mi) is None


def _comm_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    sharding = importlib.import_module("lsp-dsp-units_amd.sharding")
    mi = importlib.import_module("lsp-dsp-units_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = sharding.library_comm(mi)                        # no device here: nobody can join -- and everybody must say so
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("none" if comm is None else "comm")
    dist.barrier()
    dist.destroy_process_group()


def test_library_comm_agrees_across_ranks_when_it_cannot_be_built(tmp_path):
    """Two ranks on a box without a device: the library's communicator cannot be created -- every rank must come out of
    sharding.library_comm with None (and out of it at all: a rank that went on alone would hang the job at its first collective)."""
    mi = importlib.import_module("lsp-dsp-units_amd")
    if mi.device_count() > 0:
        pytest.skip("a device is present: the communicator would be built")
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_comm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() for r in range(world)] == ["none", "none"]
