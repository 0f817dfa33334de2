"""The three kernels built on csrc/fft_wave.h against the calls one by one over many run lengths (experiment, not a test; on the GPU box):
   python tests/experiments/wave_kernels_soak.py [first K] [last K]
Every K: the equalizer's, the SpectralProcessor's and the splitter's run-of-blocks test at rank 12 (segments, warm-up blocks, the state
a run leaves behind for the call after it)."""
import importlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
ts = importlib.import_module("test_spectral_gpu")
tp = importlib.import_module("test_splitter_gpu")
te = importlib.import_module("test_equalizer_gpu")
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 80)
bad = []
rng = np.random.default_rng(5)
for K in range(a, b):
    bands = int(rng.integers(2, 7))
    listen = sorted(rng.choice(bands, size=int(rng.integers(1, bands + 1)), replace=False).tolist())
    for name, fn in (("stft", lambda: ts.test_spectral_process_blocks_equal_block_by_block(gpu, 12, True, 2, K)),
                     ("splitter", lambda: tp.test_process_blocks_equal_block_by_block(gpu, 12, bands, 2, K, listen)),
                     ("splitter/oracle", (lambda: tp.test_runs_of_blocks_match_the_oracle(gpu, 12, min(bands, 4), 2, K)) if K % 7 == 0 else None),
                     ("equalizer ring", (lambda: te.test_runs_of_blocks_into_a_ring_of_buffers(gpu, 8 if K % 2 else 16)) if K % 9 == 0 else None)):
        if fn is None:
            continue
        try:
            fn()
        except AssertionError as e:
            bad.append((name, K, bands, listen)); print("FAILED", name, K, bands, listen, str(e)[:200].replace("\n", " "))
print("wave kernels soak, K in [%d, %d): failed" % (a, b), bad)
