"""Two ranks, ONE GPU, the PRODUCT: every rank runs the analyzer bank of its channel shard on the device (the strobes of a
batch as one launch, the shard-composable per-bin sums), the shards' sums are all-reduced over gloo
(lsp-dsp-units_amd.sharding.allreduce_bins), and rank 0 compares with the unsharded bank on the same device -- bit for bit:
the top level of the unsharded reduction's binary tree IS the sum of the two halves.  Started by
tests/test_spectral_gpu.py::test_product_shards_and_collective_equal_the_unsharded_bank through torch.distributed.run
(--nproc-per-node 2, gloo).  Prints one JSON line on rank 0."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

mi = importlib.import_module("lsp-dsp-units_amd")
sharding = importlib.import_module("lsp-dsp-units_amd.sharding")


def bank(C, rank_fft, hop, sr=48000):
    an = mi.AnalyzerBank(C, rank_fft, sr, 1.0, 0)
    for what, v in ((an.SAMPLE_RATE, sr), (an.RATE, sr / float(hop)), (an.RANK, rank_fft), (an.WINDOW, 0), (an.REACTIVITY, 0.2), (an.SHIFT, 1.0)):
        an.configure(what, v)
    return an


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    total, rank_fft, hop, batch, batches = int(sys.argv[1]), int(sys.argv[2]), 1 << (int(sys.argv[2]) - 1), 16, 3
    lo, hi = sharding.shard_range(total, rank, world)
    gen = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn((batches * batch + 1, total, hop), generator=gen, dtype=torch.float32) * 0.25
    bins = (1 << (rank_fft - 1)) + 1
    stream = torch.cuda.current_stream()

    def run(an, rows):
        xs = x[:, rows[0]:rows[1]].contiguous().to(dev)
        an.process(xs[0], hop, stream=stream)                                   # (fills the first half of the first frame)
        out = []
        for b in range(batches):
            sums = torch.zeros((batch, bins), dtype=torch.float32, device=dev)
            an.process_reduce_frames([xs[1 + b * batch + j] for j in range(batch)], hop, sums, stream=stream)
            out.append(sums)
        torch.cuda.synchronize()
        return out

    mine = run(bank(hi - lo, rank_fft, hop), (lo, hi))
    launch = mi.last_launch()
    reduced = []
    for s in mine:
        t = s.cpu()
        sharding.allreduce_bins(t)                                              # gloo: the shards' sums, summed
        reduced.append(t)
    ok, worst, peak = True, 0.0, 0.0
    if rank == 0:
        whole = run(bank(total, rank_fft, hop), (0, total))
        for a, b in zip(reduced, whole):
            b = b.cpu()
            ok = ok and bool(torch.equal(a, b))
            worst = max(worst, float((a - b).abs().max()))
            peak = max(peak, float(b.abs().max()))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "channels": total, "rank": rank_fft, "frames": batches * batch, "bit_equal": ok,
                          "worst_abs_diff": worst, "peak": peak, "last_launch": launch}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
