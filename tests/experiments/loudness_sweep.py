"""Per-launch time of LoudnessBank.process against the block length (512 stereo meters).  Run on the GPU box."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mi = importlib.import_module("lsp-dsp-units_amd")
dev = torch.device("cuda", 0)
M, K = 512, 2
period = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
for n in ([int(a) for a in sys.argv[2:]] or [256, 1024, 2048, 4096]):
    lm = mi.LoudnessBank(M, K, period)
    lm.set_sample_rate(48000)
    x = (torch.randn((M * K, n), dtype=torch.float32) * 0.25).to(dev)
    o = torch.empty((M, n), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream()
    for _ in range(20):
        lm.process(o, None, x, n, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 300
    e0.record()
    for _ in range(reps):
        lm.process(o, None, x, n, stream=st)
    e1.record()
    torch.cuda.synchronize()
    print("period %.0f ms  n=%5d  %.2f us per process()" % (period, n, e0.elapsed_time(e1) * 1e3 / reps), flush=True)
    lm.close()
