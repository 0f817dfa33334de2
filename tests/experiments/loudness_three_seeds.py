import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
gpu = importlib.import_module("lsp-dsp-units_amd")
import test_loudness_gpu as t
for seed in (29232, 29481, 4902):
    try:
        t.test_random_operation_sequences(gpu, seed); print(seed, "ok")
    except AssertionError as e:
        print(seed, "FAILED", str(e)[:900])
