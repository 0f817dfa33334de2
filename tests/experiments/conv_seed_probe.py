# runs test_subframe_calls_on_large_frames for the given seeds and prints every failure (experiment helper)
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
mod = importlib.import_module("test_convolver_gpu")
for seed in [int(v) for v in sys.argv[1:]]:
    try:
        mod.test_subframe_calls_on_large_frames(gpu, seed)
        print("seed", seed, "ok")
    except AssertionError as e:
        print("seed", seed, "FAILED", str(e)[:200].replace("\n", " "))
