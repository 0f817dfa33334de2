# LoudnessMeter + ILUFSMeter banks (512 stereo meters each, 4096-sample calls): both on one stream against one stream each.
# The banks are independent objects (as in the reference); a host that runs both on the same rows may overlap them.
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
M, K, n = 512, 2, 4096
lm = mi.LoudnessBank(M, K, 400.0); lm.set_sample_rate(48000)
im = mi.ILUFSBank(M, K, 10.0, 400.0); im.set_sample_rate(48000)
x = (torch.randn((4, M * K, n)) * 0.25).cuda()
o1 = torch.empty((M, n), device="cuda"); o2 = torch.empty((M, n), device="cuda")
s0 = torch.cuda.current_stream(); s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
def serial(i):
    lm.process(o1, None, x[i % 4], n, stream=s0); im.process(o2, x[i % 4], n, stream=s0)
def split(i):
    lm.process(o1, None, x[i % 4], n, stream=s1); im.process(o2, x[i % 4], n, stream=s2)
for name, fn in (("one stream", serial), ("one stream each", split), ("one stream", serial), ("one stream each", split)):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s1.wait_stream(s0); s2.wait_stream(s0)
        e0.record(s0)
        s1.wait_event(e0); s2.wait_event(e0)
        for i in range(200):
            fn(i)
        s0.wait_stream(s1); s0.wait_stream(s2)
        e1.record(s0); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    print("%-16s %.1f us per step" % (name, best))
