"""Sweep of test_process_blocks_random_geometries over many seeds: blocks_stress_sweep.py <first> <count> [biquad | crossover]
(round 4: biquad 1000 .. 1599 and 2000 .. 2299 on the last tree, crossover 1000 .. 1599)."""
import importlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gpu = importlib.import_module("lsp-dsp-units_amd")
first, count = int(sys.argv[1]), int(sys.argv[2])
t = importlib.import_module("test_%s_gpu" % (sys.argv[3] if len(sys.argv) > 3 else "biquad"))
bad = []
for seed in range(first, first + count):
    try:
        t.test_process_blocks_random_geometries(gpu, seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
        print("FAIL", seed, str(e)[:300], flush=True)
print("seeds %d .. %d: %d failures" % (first, first + count - 1, len(bad)))
