"""One-off sweep of tests/test_biquad_gpu.py::test_process_blocks_random_geometries over many seeds (round 4: 1000 .. 1599)."""
import importlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gpu = importlib.import_module("lsp-dsp-units_amd")
t = importlib.import_module("test_biquad_gpu")
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, first + count):
    try:
        t.test_process_blocks_random_geometries(gpu, seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
        print("FAIL", seed, str(e)[:300], flush=True)
print("seeds %d .. %d: %d failures" % (first, first + count - 1, len(bad)))
