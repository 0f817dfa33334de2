# C5 (4096-point spectrum of 1024 channels every 2048 samples + the per-bin sum over the channels): the analysis and the
# reduction on ONE stream against the reduction of frame i on a second stream underneath the analysis of frame i + 1.
# The analysis double-buffers its spectra (vAmp old / new), so frame i + 1 does not write what the reduction of frame i reads;
# it does write what the reduction of frame i - 1 read, hence the wait for that one.  Same sums either way (checked).
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
C, rank, hop, sr, batch = 1024, 12, 2048, 48000, 8
def bank():
    an = mi.AnalyzerBank(C, rank, sr, 20.0, 0)       # ring of 8960 samples: the positions repeat every 70 frames
    for what, v in ((an.SAMPLE_RATE, sr), (an.RATE, sr / float(hop)), (an.RANK, rank), (an.WINDOW, 0), (an.REACTIVITY, 0.2), (an.SHIFT, 1.0)):
        an.configure(what, v)
    return an
bins = (1 << (rank - 1)) + 1
x = torch.randn((8, C, hop)).cuda()
torch.cuda.synchronize()
s0 = torch.cuda.Stream(); s1 = torch.cuda.Stream()
def serial(an, sums, n):
    for i in range(n):
        an.process(x[i % 8], hop, stream=s0)
        an.reduce_bins(sums[i % batch], stream=s0)
def piped(an, sums, n):
    done = [None, None]
    for i in range(n):
        if done[i & 1] is not None:
            s0.wait_event(done[i & 1])                      # the reduction of frame i - 2 read the buffer frame i writes
        an.process(x[i % 8], hop, stream=s0)
        ev = torch.cuda.Event(); ev.record(s0)
        s1.wait_event(ev)
        an.reduce_bins(sums[i % batch], stream=s1)
        done[i & 1] = torch.cuda.Event(); done[i & 1].record(s1)
    s0.wait_stream(s1)
res = {}
for name, fn in (("one stream", serial), ("reduction on a second stream", piped)):
    an = bank(); sums = torch.zeros((batch, bins), device="cuda")
    fn(an, sums, 16); torch.cuda.synchronize()
    res[name] = sums.clone()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0); fn(an, sums, 200); e1.record(s0); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    print("%-30s %.2f us per frame" % (name, best))
    an.close()
# the same two forms as hipGraphs of 280 frames (no host work per frame: what the bench's regions do)
import ctypes
for name, fn in (("one stream, hipGraph", serial), ("second stream, hipGraph", piped)):
    an = bank(); sums = torch.zeros((batch, bins), device="cuda")
    fn(an, sums, 16); torch.cuda.synchronize()
    mi.check(mi.lib.mi_dspu_graph_begin_capture(ctypes.c_void_p(s0.cuda_stream)))
    fn(an, sums, 280)
    h = ctypes.c_void_p()
    mi.check(mi.lib.mi_dspu_graph_end_capture(ctypes.c_void_p(s0.cuda_stream), ctypes.byref(h)))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0)
        for _ in range(4):
            mi.check(mi.lib.mi_dspu_graph_launch(h, ctypes.c_void_p(s0.cuda_stream)))
        e1.record(s0); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 1120 * 1e3)
    print("%-30s %.2f us per frame" % (name, best))
    res[name] = sums.clone()
    mi.lib.mi_dspu_graph_destroy(h)
    an.close()
print("same sums (graphs):", bool(torch.equal(res["one stream, hipGraph"], res["second stream, hipGraph"])))
print("same sums:", bool(torch.equal(res["one stream"], res["reduction on a second stream"])))
