"""Runs the equalizer differential stress of tests/test_equalizer_gpu.py over many seeds (experiment, not a test)."""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
import conftest  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
import test_equalizer_gpu as t
bad = []
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        t.test_random_operation_sequences_match_oracle.__wrapped__(gpu, seed) if hasattr(t.test_random_operation_sequences_match_oracle, "__wrapped__") else t.test_random_operation_sequences_match_oracle(gpu, seed)
    except AssertionError as e:
        bad.append(seed); print("seed", seed, "FAILED", str(e)[:300])
print("failed seeds:", bad)
