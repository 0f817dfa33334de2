"""Rate of whole-frame and sub-frame calls at the small ranks: ranks 9 and 10, 1024 channels, an impulse
response of 8 partitions; the same input is pushed as whole frames, half frames, quarter frames and as 31-sample pieces
(the reference utests' chunking, utest/util/convolver.cpp:88-136).  python tests/experiments/conv_small_rank_rate.py"""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
dev = torch.device("cuda:0")
C = 1024
res = {}
for rank in (9, 10, 11):
    B = 1 << (rank - 1)
    taps = 8 * B - 17
    g = torch.Generator().manual_seed(rank)
    irs = (torch.randn((C, taps), generator=g) / taps ** 0.5)
    total = 64 * B
    x = (torch.randn((C, total), generator=g) * 0.25).to(dev)
    out = torch.empty_like(x)
    for piece in (B, B // 2, B // 4, 31):
        cb = mi.ConvolverBank(irs, rank)
        stream = torch.cuda.current_stream()
        def run():
            o = 0
            while o < total:
                n = min(piece, total - o)
                cb.process(out[:, o:], x[:, o:], n, out_stride=total, in_stride=total, stream=stream)
                o += n
        run(); torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter(); run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
        res["rank%d_piece%d" % (rank, piece)] = {"us_per_call": round(best / ((total + piece - 1) // piece) * 1e6, 2),
                                                 "msamples_s": round(C * total / best / 1e6, 1), "last_launch": mi.last_launch()}
        print("rank", rank, "piece", piece, res["rank%d_piece%d" % (rank, piece)], flush=True)
        cb.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/conv_small_rank_rate.json", "w"), indent=1)
