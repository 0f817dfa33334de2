"""Helper of tests/test_crossover_gpu.py::test_fused_chain_equals_one_launch_per_filter: runs a 4-band crossover over
three blocks (4096 samples unless given) and saves the bands.  With MI_DSPU_TEST_PATH=crossover_unfused the bank runs one launch per
filter, with MI_DSPU_TEST_PATH=blocks_loop long calls stay with the super-block loop of biquad_chain_kernel."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(out_path, handlers, block=4096):
    mi = importlib.import_module("lsp-dsp-units_amd")
    C, bands, blocks = 6, 4, 3
    x = (np.random.default_rng(11).standard_normal((C, blocks * block)) * 0.25).astype(np.float32)
    bank = mi.CrossoverBank(C, bands)
    bank.set_sample_rate(48000)
    for i, (f, slope) in enumerate(((150.0, 2), (1200.0, 3), (6000.0, 1))):
        bank.set_slope(i, slope)
        bank.set_frequency(i, f)
    bank.set_gain(1, 1.5)
    got = np.zeros((bands, C, blocks * block), np.float32)
    for k in range(blocks):
        seg = slice(k * block, (k + 1) * block)
        din = mi.DeviceBuffer.from_host(x[:, seg])
        outs = [mi.DeviceBuffer.from_host(np.full((C, block), 7.0, np.float32)) if b in handlers else None for b in range(bands)]
        bank.process(outs, din, block)
        for b in handlers:
            got[b][:, seg] = outs[b].download()
    bank.close()
    np.save(out_path, got)


if __name__ == "__main__":
    run(sys.argv[1], [int(v) for v in sys.argv[2].split(",")], int(sys.argv[3]) if len(sys.argv) > 3 else 4096)
