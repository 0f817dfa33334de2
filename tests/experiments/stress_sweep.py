"""Runs one of the differential stress tests over many seeds (experiment, not a test):
   python tests/experiments/stress_sweep.py <test module> <test function> <first seed> <last seed> [further arguments of the test, integers]"""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
mod = importlib.import_module(sys.argv[1])
fn = getattr(mod, sys.argv[2])
bad = []
for seed in range(int(sys.argv[3]), int(sys.argv[4])):
    try:
        fn(gpu, seed, *[int(v) for v in sys.argv[5:]])
    except AssertionError as e:
        bad.append(seed); print("seed", seed, "FAILED", str(e)[:300].replace("\n", " "))
print(sys.argv[2], "failed seeds:", bad)
