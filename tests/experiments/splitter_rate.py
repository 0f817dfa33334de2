# SpectralSplitter, rank 12, 4 bands of gains, 4096-sample calls on resident buffers, at several channel counts: one launch per
# hop against the hops of the call in one launch with one or two handlers per workgroup.  (One launch per hop runs one
# workgroup per (channel, handler) up to 512 channels and one per channel above.)
#   python3 tests/experiments/splitter_rate.py        (timed with events around a hipGraph-free loop of 200 calls)
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
gpu = importlib.import_module("lsp-dsp-units_amd")
rank, bands, n = 12, 4, 4096
edges = [(None, (300.0, -32.0)), ((300.0, -32.0), (2000.0, -32.0)), ((2000.0, -32.0), (8000.0, -32.0)), ((8000.0, -32.0), None)]
for C in (64, 256, 512, 1024, 2048):
    x = (torch.randn((C, n)) * 0.25).cuda()
    outs = [torch.empty((C, n), device="cuda") for _ in range(bands)]
    row = []
    for label, env in (("one launch per hop", {"MI_SPLITTER_HOP_LAUNCHES": "1"}), ("hops in one launch", {}),
                       ("one launch, two handlers per workgroup", {"MI_SPLITTER_BANDS_PER_WG": "2"})):
        for k in ("MI_SPLITTER_HOP_LAUNCHES", "MI_SPLITTER_BANDS_PER_WG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        sp = gpu.SplitterBank(C, rank, bands)
        for b, (hp, lp) in enumerate(edges):
            sp.bind_mask(b, gpu.crossover_fft_mask(hp, lp, 1.0, 1.0, 48000, rank))
        st = torch.cuda.current_stream()
        for _ in range(10):
            sp.process(outs, x, n, stream=st)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                sp.process(outs, x, n, stream=st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        row.append("%s %.1f us" % (label, best))
        sp.close()
    print("%5d channels: " % C + " | ".join(row))
