"""The batch loop of bench.py's C5 row with the collective beside the next batch (mi_analyzer_bank_allreduce_bins_begin /
mi_dspu_comm_wait), on ONE GPU through a single-rank RCCL communicator: used by tests/test_spectral_gpu.py (the double-buffered
sums equal the serial ones bit for bit) and by tests/prof_comm_overlap.sh (a rocprofv3 trace of what runs beside what).

usage: python tests/comm_overlap_demo.py [batches] [channels]   -> prints the sums' checksum per batch"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(mi, batches=6, C=1024, overlapped=True, rank_fft=12, hop=2048, batch=16, seed=70):
    """Returns the list of per-batch sums [batch][bins] after the collective."""
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    sr = 48000
    bins = (1 << (rank_fft - 1)) + 1
    an = mi.AnalyzerBank(C, rank_fft, sr, 1.0, 0)
    for what, v in ((an.SAMPLE_RATE, sr), (an.RATE, sr / float(hop)), (an.RANK, rank_fft), (an.WINDOW, 0), (an.REACTIVITY, 0.2), (an.SHIFT, 1.0)):
        an.configure(what, v)
    rng = np.random.default_rng(seed)
    ring = 8
    xin = [mi.DeviceBuffer.from_host((rng.standard_normal((C, hop)) * 0.25).astype(np.float32)) for _ in range(ring)]
    an.process(xin[0], hop)
    comm = mi.Comm(mi.Comm.unique_id(), 1, 0)
    sums = [mi.DeviceBuffer((batch, bins)) for _ in range(2)]
    totals = [mi.DeviceBuffer((batch, bins)) for _ in range(batches)]      # (one per batch: nothing is read back inside the loop)
    for k in range(batches):
        b = k & 1
        blocks = [xin[(k * batch + j) % ring] for j in range(batch)]
        if overlapped:
            comm.wait(b)                                    # the collective that read sums[b] two batches ago
            an.process_reduce_frames(blocks, hop, sums[b])
            an.allreduce_bins_begin(sums[b], totals[k], batch, comm, b)
        else:
            an.process_reduce_frames(blocks, hop, sums[b])
            an.allreduce_bins(sums[b], batch, comm)
            totals[k].upload(sums[b].download())
    if overlapped:
        comm.wait(0)
        comm.wait(1)
    mi.check(mi.lib.mi_dspu_stream_synchronize(None))
    out = [t.download() for t in totals]
    mi.check(mi.lib.mi_dspu_stream_synchronize(None))
    comm.close()
    an.close()
    return out


if __name__ == "__main__":
    mi = importlib.import_module("lsp-dsp-units_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    for k, s in enumerate(run(mi, n, C)):
        print("batch %d: sum of sums %.6e" % (k, float(s.astype(np.float64).sum())))
