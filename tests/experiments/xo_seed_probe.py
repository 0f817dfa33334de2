import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
gpu = importlib.import_module("lsp-dsp-units_amd")
import test_crossover_gpu as t
for seed in (10624, 10996):
    try:
        t.test_random_retune_scripts(gpu, seed); print(seed, "ok")
    except AssertionError as e:
        m = str(e); print(seed, "FAILED", m[:200], "...", m[-300:])
