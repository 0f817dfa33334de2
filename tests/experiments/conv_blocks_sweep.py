"""Sweep of tests/test_convolver_gpu.py::test_process_blocks_random_scripts over many seeds: conv_blocks_sweep.py <first> <count>."""
import importlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gpu = importlib.import_module("lsp-dsp-units_amd")
t = importlib.import_module("test_convolver_gpu")
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    try:
        t.test_process_blocks_random_scripts(gpu, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, str(e)[:300], flush=True)
print("seeds %d .. %d: %d failures" % (first, first + count - 1, bad))
