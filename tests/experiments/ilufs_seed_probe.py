"""Experiment: replays one seed of tests/test_ilufs_gpu.py::test_random_operation_sequences and prints, at the failing call, the
meters' gating histories (oracle and GPU) relative to the absolute gate.   python tests/experiments/ilufs_seed_probe.py <seed>"""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
import test_ilufs_gpu as t
from oracle import ilufs as oi

seed = int(sys.argv[1])
orig_run = t._run
state = {}

def spy(gpu_, bank, refs, x, calls, K, gain):
    got, want = orig_run(gpu_, bank, refs, x, calls, K, gain=gain)
    state.update(bank=bank, refs=refs, got=got, want=want, n=x.shape[1])
    hist, head, count = bank.history()
    print("call n=%d blk=%d level=%.1e:" % (x.shape[1], refs[0].block_size, float(np.abs(x).max())),
          " ".join("m%d oracle count %d part %d off %d full %d | gpu count %d head %d" % (m, r.ms_count, r.block_part, r.block_offset, int(r.blk_full), int(count[m]), int(head[m]))
                   for m, r in enumerate(refs)))
    return got, want
t._run = spy
try:
    t.test_random_operation_sequences(gpu, seed)
    print("seed", seed, "passes")
except AssertionError as e:
    print("FAILED", str(e)[:300])
    bank, refs, got, want = state["bank"], state["refs"], state["got"], state["want"]
    GATE = float(oi.GATING_ABS_THRESH)
    hist, head, count = bank.history()
    for m, r in enumerate(refs):
        h = np.asarray(r.hist, np.float64).ravel()
        live = [(r.ms_head + r.ms_size - 1 - k) % r.ms_size for k in range(r.ms_count)]
        print("meter", m, "max_int", float(r.max_int_time), "ms_count", r.ms_count, "gpu count", int(count[m]), "loud oracle", float(r.loud), "gpu", float(bank.loudness()[m]))
        print("   oracle blocks / gate:", [round(float(h[i]) / GATE, 4) for i in live][:24])
        print("   gpu    blocks / gate:", [round(float(hist[m][i]) / GATE, 4) for i in live][:24])
        print("   gpu history, every slot up to its count:", [round(float(v) / GATE, 4) for v in hist[m][:int(count[m]) + 1]], "oracle gate margin", r.gate_margin)
        d = np.abs(got[m] - want[m]); i = int(d.argmax())
        print("   worst sample", i, "of", state["n"], "gpu", float(got[m][i]), "oracle", float(want[m][i]))
