"""GPU parity of mi_biquad_bank_* (FilterBank::process) against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

import oracle
from oracle import filter_design as fd
from conftest import assert_iir_parity, note, parity_report

import workloads as wl

pytestmark = pytest.mark.gpu


def run_bank(gpu, x, coef_list, blocks=1, in_place=False, pad=0, clear=False):
    """x: [blocks][C][n]; coef_list: per-channel (ns,5) arrays. Returns (y, state)."""
    nb, C, n = x.shape
    max_sec = max(1, max(len(c) for c in coef_list))
    bank = gpu.BiquadBank(C, max_sec)
    for c in range(C):
        bank.set_chains(c, coef_list[c], clear)
    stride = n + pad
    y = np.empty_like(x)
    dbuf_in = gpu.DeviceBuffer((C, stride))
    dbuf_out = dbuf_in if in_place else gpu.DeviceBuffer((C, stride))
    for b in range(nb):
        host = np.zeros((C, stride), np.float32)
        host[:, :n] = x[b]
        dbuf_in.upload(host)
        bank.process(dbuf_out, dbuf_in, n, stride, stride)
        y[b] = dbuf_out.download()[:, :n]
    st = bank.get_state()
    bank.close()
    return y, st


def oracle_bank(x, coef_list):
    nb, C, n = x.shape
    y32 = np.empty_like(x)
    y64 = np.empty(x.shape, np.float64)
    states = []
    for c in range(C):
        st = None
        for b in range(nb):
            y32[b, c], st = oracle.biquad_cascade(x[b, c], coef_list[c], st)
        states.append(st)
        y64[:, c, :] = oracle.biquad_cascade_f64(x[:, c, :].reshape(-1), coef_list[c]).reshape(nb, n)
    return y32, y64, states


def check_all(gpu_y, y32, y64, what):
    worst = None
    for c in range(gpu_y.shape[1]):
        r = assert_iir_parity(gpu_y[:, c], y32[:, c], y64[:, c], "%s ch%d" % (what, c))
        if worst is None or r["gpu_vs_ref32"] > worst["gpu_vs_ref32"]:
            worst = r
    return worst


def test_c1_readme_hishelf(gpu):
    """BASELINE config 0: 1 ch x 48000, FLT_BT_BWC_HISHELF slope 2 @1 kHz +6 dB (README.md:176-191), in place."""
    gain = float(np.float32(np.exp(np.float32(6.0) * np.float32(np.log(10.0)) * np.float32(0.05))))
    bq = wl.design(fd.FLT_BT_BWC_HISHELF, 2, 1000.0, 1000.0, gain, 0.0)
    assert bq.shape == (2, 5)
    x = (np.random.default_rng(1).standard_normal((1, 1, 48000)) * 0.25).astype(np.float32)
    y, _ = run_bank(gpu, x, [bq], in_place=True)
    y32, y64, _ = oracle_bank(x, [bq])
    r = check_all(y, y32, y64, "C1")
    assert r["gpu_vs_ref32"] <= 1e-5            # well-conditioned: the strict north-star tolerance holds
    imp = np.zeros((1, 1, 4096), np.float32)
    imp[0, 0, 0] = 1.0
    yi, _ = run_bank(gpu, imp, [bq])
    np.testing.assert_allclose(yi[0, 0, :8], [1.93714225, -0.114121534, -0.109540939, -0.10430833,
                                              -0.0985007137, -0.0922033042, -0.0855074227, -0.0785082579],
                               rtol=0, atol=1e-6)


def test_c2_shape_state_carry(gpu):
    """G2: 8-section LRX lowpass, 4 channels x 4096 x 3 consecutive blocks (state carried between calls)."""
    coef, _ = wl.c2_coefficients(4)
    x = wl.c2_input(4, 4096, blocks=3)
    y, st = run_bank(gpu, x, list(coef))
    y32, y64, states = oracle_bank(x, list(coef))
    check_all(y, y32, y64, "C2x4")
    # delay memory after the last block agrees with the oracle's (same scale as the output)
    for c in range(4):
        scale = max(1e-30, np.abs(states[c]).max())
        assert np.abs(st[c, :8] - states[c]).max() / scale < 5e-3


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 31, 64, 100, 511, 512, 513, 1000, 4095, 4096, 4097, 9001])
def test_ragged_lengths(gpu, n):
    rng = np.random.default_rng(100 + n)
    coefs = [wl.design(fd.FLT_BT_RLC_BELL, 2, 1000.0 * (c + 1), 0, 2.0, 1.0) for c in range(3)]
    x = (rng.standard_normal((2, 3, n))).astype(np.float32)
    y, _ = run_bank(gpu, x, coefs, pad=(n % 3))          # odd strides exercise the unaligned path
    y32, y64, _ = oracle_bank(x, coefs)
    check_all(y, y32, y64, "n=%d" % n)


@pytest.mark.parametrize("ns", [0, 1, 2, 3, 4, 7, 8, 9, 15, 16, 32])
def test_section_counts(gpu, ns):
    """x8/x4/x2/x1 packing of FilterBank::end() (FilterBank.cpp:106-236) == ns sections in series."""
    rng = np.random.default_rng(7 + ns)
    if ns == 0:
        coefs = [np.zeros((0, 5), np.float32)] * 2
    else:
        one = wl.design(fd.FLT_BT_RLC_BELL, ns, 2500.0, 0, 1.5, 0.7)
        assert one.shape[0] == ns
        coefs = [one, one[::-1].copy()]
    x = rng.standard_normal((2, 2, 2048)).astype(np.float32)
    y, _ = run_bank(gpu, x, coefs)
    y32, y64, _ = oracle_bank(x, coefs)
    check_all(y, y32, y64, "ns=%d" % ns)
    if ns == 0:
        np.testing.assert_array_equal(y, x)          # FilterBank.cpp:261-265: empty bank copies


def test_mixed_channels_and_in_place(gpu):
    rng = np.random.default_rng(11)
    coefs = [wl.design(fd.FLT_BT_LRX_HIPASS, 2, 100.0, 0, 1.0, 0.0),
             np.zeros((0, 5), np.float32),
             wl.design(fd.FLT_K_WEIGHTED),
             wl.design(fd.FLT_BT_BWC_LOPASS, 3, 8000.0, 0, 1.0, 0.2),
             wl.design(fd.FLT_DR_APO_PEAKING, 1, 3000.0, 0, 0.25, 4.0)]
    x = rng.standard_normal((3, 5, 1536)).astype(np.float32)
    y, _ = run_bank(gpu, x, coefs, in_place=True)
    y32, y64, _ = oracle_bank(x, coefs)
    check_all(y, y32, y64, "mixed")


def test_low_frequency_filters_noise_floor(gpu):
    """20-200 Hz sections: the float32 recursion is only reproducible to its own round-off noise
    (reference float32 vs float64 differ by >1e-5 here); the GPU must not be worse than that."""
    rng = np.random.default_rng(5)
    coefs = [wl.design(fd.FLT_BT_RLC_BELL, 4, 20.0, 0, 4.0, 2.0),
             wl.design(fd.FLT_BT_LRX_LOPASS, 4, 200.0, 0, 1.0, 0.75),
             wl.design(fd.FLT_BT_LRX_HIPASS, 4, 30.0, 0, 1.0, 0.0)]
    x = (rng.standard_normal((2, 3, 4096)) * 0.25).astype(np.float32)
    y, _ = run_bank(gpu, x, coefs)
    y32, y64, _ = oracle_bank(x, coefs)
    for c in range(3):
        r = assert_iir_parity(y[:, c], y32[:, c], y64[:, c], "lowfreq ch%d" % c)
        assert r["gpu_vs_exact"] <= 4.0 * max(r["noise"], 1e-5)


def test_clear_and_reset_semantics(gpu):
    """FilterBank::end(clear) clears delays when asked or when the section count changed (FilterBank.cpp:233-235)."""
    bq2 = wl.design(fd.FLT_BT_RLC_BELL, 2, 500.0, 0, 2.0, 1.0)
    bq3 = wl.design(fd.FLT_BT_RLC_BELL, 3, 500.0, 0, 2.0, 1.0)
    x = np.random.default_rng(3).standard_normal(1024).astype(np.float32)
    bank = gpu.BiquadBank(1, 8)
    din = gpu.DeviceBuffer.from_host(x.reshape(1, -1))
    dout = gpu.DeviceBuffer((1, 1024))
    bank.set_chains(0, bq2)
    bank.process(dout, din, 1024)
    st1 = bank.get_state()
    assert np.abs(st1[0, :2]).max() > 0
    bank.set_chains(0, bq2, clear=False)             # same count, no clear: memory kept
    bank.commit()
    np.testing.assert_array_equal(bank.get_state(), st1)
    bank.set_chains(0, bq3, clear=False)             # count changed: memory cleared
    assert bank.size(0) == 3
    np.testing.assert_array_equal(bank.get_state(), np.zeros_like(st1))
    bank.process(dout, din, 1024)
    bank.set_chains(0, bq3, clear=True)              # explicit clear
    np.testing.assert_array_equal(bank.get_state(), np.zeros_like(st1))
    bank.process(dout, din, 1024)
    y1 = dout.download()
    bank.reset()
    np.testing.assert_array_equal(bank.get_state(), np.zeros_like(st1))
    bank.process(dout, din, 1024)
    np.testing.assert_array_equal(dout.download(), y1)      # same start state -> bit-identical rerun
    bank.close()


@pytest.mark.parametrize("n", [4096, 1000, 37])
def test_rows_switched_off_keep_state_and_output(gpu, n):
    """mi_biquad_bank_set_row_enabled: what a caller of the reference gets by not calling FilterBank::process() for an
    object -- every launch variant (L = 16 / 8, the tail kernel) leaves the row's delay memory and its output alone."""
    C = 5
    rng = np.random.default_rng(n)
    x = rng.standard_normal((C, 2 * n)).astype(np.float32)
    bank = gpu.BiquadBank(C, 8)
    coef = wl.design(fd.FLT_BT_LRX_LOPASS, 4, 1500.0, 0, 1.0, 0.7)
    for c in range(C):
        bank.set_chains(c, coef)
    out = gpu.DeviceBuffer((C, n))
    bank.process(out, gpu.DeviceBuffer.from_host(x[:, :n]), n)
    st = bank.get_state()
    bank.set_row_enabled(1, False); bank.set_row_enabled(3, False)
    marked = np.full((C, n), -5.0, np.float32)
    out2 = gpu.DeviceBuffer.from_host(marked)
    bank.process(out2, gpu.DeviceBuffer.from_host(x[:, n:]), n)
    y2 = out2.download(); st2 = bank.get_state()
    for c in (1, 3):
        assert np.all(y2[c] == -5.0)
        np.testing.assert_array_equal(st2[c], st[c])
    for c in (0, 2, 4):
        r, _ = oracle.biquad_cascade(x[c], coef)
        assert_iir_parity(y2[c], r[n:], oracle.biquad_cascade_f64(x[c], coef)[n:])
    bank.set_row_enabled(1, True)                             # back on: continues from the memory it kept
    bank.process(out2, gpu.DeviceBuffer.from_host(x[:, n:]), n)
    r, _ = oracle.biquad_cascade(x[1], coef)
    assert_iir_parity(out2.download()[1], r[n:], oracle.biquad_cascade_f64(x[1], coef)[n:])
    bank.close()


def test_impulse_response_restores_state(gpu):
    """FilterBank::impulse_response (FilterBank.cpp:293-330)."""
    bq = wl.design(fd.FLT_BT_LRX_LOPASS, 2, 2000.0, 0, 1.0, 0.0)
    x = np.random.default_rng(4).standard_normal((2, 2048)).astype(np.float32)
    bank = gpu.BiquadBank(2, 4)
    bank.set_chains(0, bq)
    bank.set_chains(1, bq[:2])
    din = gpu.DeviceBuffer.from_host(x)
    dout = gpu.DeviceBuffer((2, 2048))
    bank.process(dout, din, 2048)
    st = bank.get_state()
    ir = gpu.DeviceBuffer((2, 600))
    bank.impulse_response(ir, 600)
    np.testing.assert_array_equal(bank.get_state(), st)
    h = ir.download()
    for c, q in enumerate([bq, bq[:2]]):
        ref = oracle.biquad_impulse_response(600, q, np.zeros((len(q), 2), np.float32))
        assert np.abs(h[c] - ref).max() <= 1e-5 * np.abs(ref).max()
    bank.close()


def run_bank_blocks(gpu, x, coef):
    """x: [blocks][C][n] through ONE mi_biquad_bank_process_blocks call (the launch bench.py's `value` comes from)."""
    nb, C, n = x.shape
    bank = gpu.BiquadBank(C, 8)
    bank.set_all_chains(coef)
    ins = [gpu.DeviceBuffer.from_host(x[b]) for b in range(nb)]
    outs = [gpu.DeviceBuffer((C, n)) for _ in range(nb)]
    bank.process_blocks(outs, ins, n)
    y = np.stack([o.download() for o in outs])
    st = bank.get_state()
    bank.close()
    return y, st


# (coefficient seed, input seed, how the 64 blocks are issued).  Seeds (3, 2) are BASELINE's C2; the others are the review's
# "hold the frozen factors 3.0 / 3.75 against >= 4 more input / cutoff draws" (VERDICT r04, next-round item 7).
C2_CASES = [(3, 2, "calls"), (3, 2, "blocks"), (13, 12, "blocks"), (23, 22, "blocks"), (33, 32, "blocks"), (43, 42, "blocks")]


@pytest.mark.parametrize("coef_seed,input_seed,how", C2_CASES)
def test_c2_full_size_all_channels(gpu, coef_seed, input_seed, how):
    """BASELINE config 1 at full size: 1024 ch x 4096, 8 sections, 64 consecutive blocks with carried state
    (BASELINE.md section 4), every channel checked against the oracle (OpenMP over channels) and against float64 --
    block by block (mi_biquad_bank_process: biquad_bank_kernel) and as ONE mi_biquad_bank_process_blocks call
    (biquad_stream_kernel: the launch the bench's headline is measured on), the latter over five draws of cutoffs and input.

    The per-channel figures behind the IIR parity rule are written to gpurun_out/c2_parity*.json (copied to
    profiles/) and summarised -- with the table per cutoff band -- in the pytest summary, so the
    distribution the rule's FROZEN factors (conftest.py: 3.0 / 3.75, set in round 3) are held against is on record: how many
    channels fall under the strict 1e-5, the worst distance from the oracle, and the worst distances in units of the float32
    recursion's own noise."""
    import json
    import os
    from conftest import IIR_EXACT_FACTOR, IIR_REF_FACTOR, IIR_RMS_MARGIN, NOISE_FLOOR, ROOT, TOL
    from concurrent.futures import ThreadPoolExecutor
    C, n, nb = 1024, 4096, 64
    coef, fc = wl.c2_coefficients(C, seed=coef_seed)
    x = wl.c2_input(C, n, blocks=nb, seed=input_seed)
    y, _ = run_bank(gpu, x, list(coef)) if how == "calls" else run_bank_blocks(gpu, x, coef)
    state = np.zeros((C, 8, 2), np.float32)
    nsec = np.full(C, 8, np.uint32)
    y32 = np.empty_like(x)
    for b in range(nb):
        y32[b] = oracle.biquad_bank(x[b], coef, nsec, state)
    rows = np.zeros((C, 7))

    def one(c):
        y64 = oracle.biquad_cascade_f64(x[:, c, :].reshape(-1), coef[c]).reshape(nb, n)
        r = parity_report(y[:, c], y32[:, c], y64)
        g, o = y[:, c].astype(np.float64), y32[:, c].astype(np.float64)
        rms = lambda v: float(np.sqrt(np.mean(v * v)))
        return (fc[c], r["noise"], r["gpu_vs_exact"], r["gpu_vs_ref32"], rms(o - y64), rms(g - y64), rms(g - o))
    with ThreadPoolExecutor(max_workers=8) as ex:
        for c, row in enumerate(ex.map(one, range(C))):
            rows[c] = row
    noise, exact, ref32 = rows[:, 1], rows[:, 2], rows[:, 3]
    strict = noise <= NOISE_FLOOR
    rx, rr = exact / noise, ref32 / noise
    loose = ~strict

    def worst(metric, mask):
        i = int(np.flatnonzero(mask)[np.argmax(metric[mask])])
        return {"value": float(metric[i]), "channel": i, "cutoff_hz": round(float(fc[i]), 1), "noise": float(noise[i])}
    summary = {
        "config": "C2: 1024 ch x 4096 x 64 blocks (state carried), FLT_BT_LRX_LOPASS slope 4, cutoffs 200 Hz .. 18 kHz",
        "rule": {"TOL": TOL, "NOISE_FLOOR": NOISE_FLOOR, "exact_factor": IIR_EXACT_FACTOR, "ref_factor": IIR_REF_FACTOR},
        "n_channels": C, "n_strict": int(strict.sum()),
        "worst_gpu_vs_ref32": worst(ref32, np.ones(C, bool)),
        "worst_gpu_vs_ref32_strict_channels": worst(ref32, strict),
        "worst_gpu_vs_exact_over_noise": worst(rx, loose) if loose.any() else None,
        "worst_gpu_vs_ref32_over_noise": worst(rr, loose) if loose.any() else None,
        "percentiles_noisy_channels": {
            "gpu_vs_exact_over_noise": {str(q): float(np.percentile(rx[loose], q)) for q in (50, 90, 99, 100)},
            "gpu_vs_ref32_over_noise": {str(q): float(np.percentile(rr[loose], q)) for q in (50, 90, 99, 100)},
        } if loose.any() else None,
        "by_cutoff": [{"band_hz": [lo, hi], "channels": int(m.sum()), "noise_max": float(noise[m].max()),
                       "gpu_vs_exact_max": float(exact[m].max()), "gpu_vs_ref32_max": float(ref32[m].max())}
                      for lo, hi in ((200, 400), (400, 800), (800, 1600), (1600, 3200), (3200, 6400), (6400, 18000))
                      for m in [(fc >= lo) & (fc < hi)] if m.any()],
    }
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    summary["case"] = {"coef_seed": coef_seed, "input_seed": input_seed, "call": "64 mi_biquad_bank_process calls" if how == "calls"
                       else "ONE mi_biquad_bank_process_blocks call of 64 blocks"}
    name = "c2_parity.json" if (coef_seed, input_seed, how) == (3, 2, "calls") else "c2_parity_%s_%d_%d.json" % (how, coef_seed, input_seed)
    with open(os.path.join(out, name), "w") as f:
        json.dump(summary, f, indent=1)
    brief = "C2 parity: %s" % json.dumps({k: summary[k] for k in ("n_strict", "worst_gpu_vs_ref32",
                                          "worst_gpu_vs_exact_over_noise", "worst_gpu_vs_ref32_over_noise")})
    note("C2 [cutoffs seed %d, input seed %d, %s] per band (Hz: channels, oracle noise max, |gpu - exact| max, |gpu - oracle| max): %s"
         % (coef_seed, input_seed, how, "; ".join("%d-%d: %d, %.1e, %.1e, %.1e" % (r["band_hz"][0], r["band_hz"][1], r["channels"], r["noise_max"],
                                                 r["gpu_vs_exact_max"], r["gpu_vs_ref32_max"]) for r in summary["by_cutoff"])))
    note("C2, all %d channels x 64 blocks: %d channels within 1e-5 of the oracle (worst %.2e); the others against the oracle's "
         "own float32 noise: |gpu - exact| / noise percentiles 50/90/99/100 = %s, |gpu - oracle| / noise = %s (bounds %g / %g)"
         % (C, summary["n_strict"], summary["worst_gpu_vs_ref32_strict_channels"]["value"] if isinstance(summary["worst_gpu_vs_ref32_strict_channels"], dict)
            else summary["worst_gpu_vs_ref32_strict_channels"],
            [round(v, 2) for v in summary["percentiles_noisy_channels"]["gpu_vs_exact_over_noise"].values()] if loose.any() else [],
            [round(v, 2) for v in summary["percentiles_noisy_channels"]["gpu_vs_ref32_over_noise"].values()] if loose.any() else [],
            IIR_EXACT_FACTOR, IIR_REF_FACTOR))
    assert np.all(np.isfinite(y)), brief
    assert np.all(ref32[strict] <= TOL), brief
    if loose.any():
        assert np.all(exact[loose] <= np.maximum(TOL, IIR_EXACT_FACTOR * noise[loose])), brief
        assert np.all(ref32[loose] <= np.maximum(TOL, IIR_REF_FACTOR * noise[loose])), brief
    assert strict.sum() > C // 4, brief
    # The derivation behind the two factors (conftest.py), on the quantity it is about: over the whole run of every channel the
    # GPU's distance from exact arithmetic is a round-off walk of at most twice the serial recursion's variance, its distance from
    # the oracle one of at most three times
    sig, e_rms, d_rms = rows[:, 4], rows[:, 5], rows[:, 6]
    ok = sig > 0
    r_exact, r_ref = e_rms[ok] / sig[ok], d_rms[ok] / sig[ok]
    note("C2 [cutoffs seed %d, input seed %d, %s] root-mean-square over the run, every channel: rms(gpu - exact) / rms(oracle - exact) "
         "percentiles 50/99/100 = %.2f %.2f %.2f (derived bound sqrt 2 = 1.41, allowed x %.2g); rms(gpu - oracle) / rms(oracle - exact) = "
         "%.2f %.2f %.2f (sqrt 3 = 1.73)" % (coef_seed, input_seed, how, *np.percentile(r_exact, (50, 99, 100)), IIR_RMS_MARGIN,
                                            *np.percentile(r_ref, (50, 99, 100))))
    from conftest import record_parity
    record_parity("iir rms over a run: rms(gpu - exact) <= %.2g sqrt 2 rms(oracle - exact)" % IIR_RMS_MARGIN, float(r_exact.max()), IIR_RMS_MARGIN * 2 ** 0.5)
    record_parity("iir rms over a run: rms(gpu - oracle) <= %.2g sqrt 3 rms(oracle - exact)" % IIR_RMS_MARGIN, float(r_ref.max()), IIR_RMS_MARGIN * 3 ** 0.5)
    assert float(r_exact.max()) <= IIR_RMS_MARGIN * 2 ** 0.5, (float(r_exact.max()), brief)
    assert float(r_ref.max()) <= IIR_RMS_MARGIN * 3 ** 0.5, (float(r_ref.max()), brief)


@pytest.mark.parametrize("n,nb", [(4096 + 48, 5), (4096, 8), (4096, 2), (4096, 3), (2064, 5), (6144, 3), (8192 + 16, 4),
                                  (4096, 131), (1024, 4), (4100, 3)])
def test_process_blocks_equals_separate_calls(gpu, n, nb):
    """mi_biquad_bank_process_blocks runs the blocks in ONE launch where they qualify (biquad_stream_kernel: more than 2048
    samples, whole chunks of 16; more than 128 blocks: several launches) and otherwise as process() calls issued from C --
    either way the same bits and the same carried state as `nb` separate process() calls.  Channels with different section
    counts, one without sections (a copy), one switched off."""
    C = 7
    rng = np.random.default_rng(321 + n + nb)
    coef = [wl.design(fd.FLT_BT_LRX_LOPASS, 4, 500.0 * (c + 1), 0, 1.0, 0.75)[:8 - c] for c in range(C - 1)]
    coef.append(np.zeros((0, 5), np.float32))
    x = (rng.standard_normal((nb, C, n)) * 0.25).astype(np.float32)

    def make_bank():
        bank = gpu.BiquadBank(C, 8)
        for c in range(C):
            bank.set_chains(c, coef[c], False)
        bank.set_row_enabled(2, False)
        return bank
    ref = make_bank()
    ins = [gpu.DeviceBuffer.from_host(x[b]) for b in range(nb)]
    sentinel = np.full((C, n), 7.0, np.float32)
    outs_ref = [gpu.DeviceBuffer.from_host(sentinel) for _ in range(nb)]
    for b in range(nb):
        ref.process(outs_ref[b], ins[b], n)
    y_ref = [o.download() for o in outs_ref]
    st_ref = ref.get_state()
    ref.close()
    assert np.all(y_ref[0][2] == 7.0) and np.array_equal(y_ref[0][6], x[0][6])        # switched off; no sections

    bank = make_bank()
    outs = [gpu.DeviceBuffer.from_host(sentinel) for _ in range(nb)]
    bank.process_blocks(outs, ins, n)
    for b in range(nb):
        np.testing.assert_array_equal(outs[b].download(), y_ref[b])
    np.testing.assert_array_equal(bank.get_state(), st_ref)
    bank.close()


@pytest.mark.parametrize("n", [8192, 8192 + 16, 65536, 48000, 3 * 4096 + 2048 + 32])
def test_long_calls_through_the_stream_kernel_same_bits(gpu, n, monkeypatch):
    """A process() call of four sub-blocks and more is walked by the stream kernel (four waves per channel) instead of the
    super-block loop of the one-block kernel: the same bits and the same carried state (MI_DSPU_TEST_PATH=blocks_loop selects the old
    path), over two consecutive calls, in place as well."""
    C = 9
    rng = np.random.default_rng(4000 + n)
    coef = [wl.design(fd.FLT_BT_LRX_LOPASS, 4, 300.0 * (c + 1), 0, 1.0, 0.75)[:8 - (c % 3)] for c in range(C)]
    x = (rng.standard_normal((2, C, n)) * 0.25).astype(np.float32)
    res = []
    for old in (True, False):
        if old:
            monkeypatch.setenv("MI_DSPU_TEST_PATH", "blocks_loop")
        else:
            monkeypatch.delenv("MI_DSPU_TEST_PATH", raising=False)
        bank = gpu.BiquadBank(C, 8)
        for c in range(C):
            bank.set_chains(c, coef[c], False)
        d0, o0 = gpu.DeviceBuffer.from_host(x[0]), gpu.DeviceBuffer((C, n))
        bank.process(o0, d0, n)
        d1 = gpu.DeviceBuffer.from_host(x[1])
        bank.process(d1, d1, n)                               # in place
        res.append((o0.download(), d1.download(), bank.get_state()))
        bank.close()
    for u, v in zip(res[0], res[1]):
        np.testing.assert_array_equal(u, v)


@pytest.mark.parametrize("seed", range(24))
def test_process_blocks_random_geometries(gpu, seed):
    """Differential stress of the one-launch path against separate process() calls, bit for bit: random channel counts, section
    counts per channel (empty and switched-off channels among them), block lengths (whole chunks of 16 above 2048 samples, and now
    and then one the stream kernel does not take), strides, numbers of blocks, blocks in place, output buffers that come round
    again, blocks that read an earlier block's output, and a second call on the state the first one left."""
    rng = np.random.default_rng(77000 + seed)
    C = int(rng.integers(1, 41))
    max_sec = int(rng.choice([1, 3, 8, 16, 32]))
    coef = []
    for c in range(C):
        k = int(rng.integers(0, max_sec + 1))
        q = [wl.design(fd.FLT_BT_RLC_BELL, 1, float(rng.uniform(60.0, 16000.0)), 0, float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.3, 3.0)))[:1]
             for _ in range(k)]
        coef.append(np.concatenate(q) if q else np.zeros((0, 5), np.float32))
    off = [c for c in range(C) if rng.integers(0, 9) == 0]
    n = int(rng.choice([2064, 4096, 4096 + 16 * int(rng.integers(1, 200)), 8192, 3 * 4096 + 32, 6000 - 6000 % 16, 1000, 4100]))
    stride = n + int(rng.choice([0, 4, 8, 64]))
    results = []
    for blocks_call in (False, True):
        r2 = np.random.default_rng(88000 + seed)            # the same plan for both runs
        bank = gpu.BiquadBank(C, max_sec)
        for c in range(C):
            bank.set_chains(c, coef[c], False)
        for c in off:
            bank.set_row_enabled(c, False)
        outputs = []
        for call in range(2):
            nb = int(r2.integers(2, 13))
            pool = [gpu.DeviceBuffer.from_host((r2.standard_normal((C, stride)) * 0.25).astype(np.float32)) for _ in range(nb + 2)]
            ins, outs = [], []
            for b in range(nb):
                kind = int(r2.integers(0, 6))
                i = pool[b]
                if kind == 0:
                    o = i                                    # in place
                elif kind == 1 and outs:
                    o = outs[int(r2.integers(0, len(outs)))] # an output buffer again
                else:
                    o = pool[nb + int(r2.integers(0, 2))] if kind == 2 else gpu.DeviceBuffer.from_host(np.full((C, stride), 3.0, np.float32))
                if kind == 3 and outs:
                    i = outs[int(r2.integers(0, len(outs)))] # reads what an earlier block wrote
                ins.append(i); outs.append(o)
            if blocks_call:
                bank.process_blocks(outs, ins, n, out_stride=stride, in_stride=stride)
            else:
                for o, i in zip(outs, ins):
                    bank.process(o, i, n, out_stride=stride, in_stride=stride)
            outputs.append([b.download() for b in pool] + [o.download() for o in outs])
        results.append((outputs, bank.get_state()))
        bank.close()
    for ca, cb in zip(results[0][0], results[1][0]):
        for u, v in zip(ca, cb):
            np.testing.assert_array_equal(u, v, err_msg=str((seed, C, max_sec, n, stride)))
    np.testing.assert_array_equal(results[0][1], results[1][1])


def test_c2_full_size_blocks_call_equals_block_by_block(gpu):
    """C2 at full size through the one-launch path: 1024 channels x 4096 x 20 blocks (the driver's bench shape) as ONE
    mi_biquad_bank_process_blocks call against 20 process() calls -- every bit of every block and of the filter memory.  (The
    block-by-block results are what test_c2_full_size_all_channels holds against the oracle.)"""
    C, n, nb = 1024, 4096, 20
    coef, _ = wl.c2_coefficients(C)
    x = wl.c2_input(C, n, blocks=nb)
    res = []
    for blocks_call in (False, True):
        bank = gpu.BiquadBank(C, 8)
        bank.set_all_chains(coef)
        ins = [gpu.DeviceBuffer.from_host(x[b]) for b in range(nb)]
        outs = [gpu.DeviceBuffer((C, n)) for _ in range(nb)]
        if blocks_call:
            bank.process_blocks(outs, ins, n)
        else:
            for o, i in zip(outs, ins):
                bank.process(o, i, n)
        res.append(([o.download() for o in outs], bank.get_state()))
        bank.close()
        del ins, outs
    for b in range(nb):
        np.testing.assert_array_equal(res[0][0][b], res[1][0][b], err_msg="block %d" % b)
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_process_blocks_falls_back_where_the_stream_kernel_does_not_apply(gpu):
    """More sections than the stream kernel has hand-over cells for (40 > 32), and rows that are not 16-byte aligned: the call
    runs the blocks as separate launches -- same bits either way."""
    C, n, nb = 3, 4096, 4
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((nb, C, n + 4)) * 0.25).astype(np.float32)
    for sections, shift in ((40, 0), (8, 1)):
        coef = [np.concatenate([wl.design(fd.FLT_BT_LRX_LOPASS, 4, 900.0 * (c + 1) + 37.0 * k, 0, 1.0, 0.75)[:8] for k in range(sections // 8)])
                for c in range(C)]
        res = []
        for blocks_call in (False, True):
            bank = gpu.BiquadBank(C, sections)
            for c in range(C):
                bank.set_chains(c, coef[c], False)
            ins = [gpu.DeviceBuffer.from_host(x[b]) for b in range(nb)]
            outs = [gpu.DeviceBuffer((C, n + 4)) for _ in range(nb)]
            vi = [b.ptr + 4 * shift for b in ins]             # raw device addresses: one float off the 16-byte grid
            vo = [b.ptr + 4 * shift for b in outs]
            if blocks_call:
                bank.process_blocks(vo, vi, n, out_stride=n + 4, in_stride=n + 4)
            else:
                for o, i in zip(vo, vi):
                    bank.process(o, i, n, out_stride=n + 4, in_stride=n + 4)
            res.append(([o.download() for o in outs], bank.get_state()))
            bank.close()
        for u, v in zip(res[0][0], res[1][0]):
            np.testing.assert_array_equal(u[:, shift:shift + n], v[:, shift:shift + n])
        np.testing.assert_array_equal(res[0][1], res[1][1])


def test_process_blocks_aliasing(gpu):
    """Blocks that depend on each other through memory keep the meaning of separate calls: a block processed in place, an
    output buffer that comes round again (a ring of buffers), a block that reads what an earlier block of the same call
    wrote (such a block starts a new launch)."""
    C, n = 5, 4096
    rng = np.random.default_rng(99)
    coef = [wl.design(fd.FLT_BT_LRX_LOPASS, 4, 700.0 * (c + 1), 0, 1.0, 0.75)[:8] for c in range(C)]
    x = (rng.standard_normal((6, C, n)) * 0.25).astype(np.float32)

    def run(blocks_call):
        bank = gpu.BiquadBank(C, 8)
        for c in range(C):
            bank.set_chains(c, coef[c], False)
        bufs = [gpu.DeviceBuffer.from_host(x[b]) for b in range(6)]
        ring = [gpu.DeviceBuffer((C, n)) for _ in range(2)]
        scratch = gpu.DeviceBuffer((C, n))
        # in place; a ring of two outputs (every block's output comes round again); a chain through `scratch`
        ins = [bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], scratch, bufs[5]]
        outs = [bufs[0], ring[0], ring[1], ring[0], scratch, ring[1], ring[0]]
        if blocks_call:
            bank.process_blocks(outs, ins, n)
        else:
            for o, i in zip(outs, ins):
                bank.process(o, i, n)
        res = [b.download() for b in bufs + ring + [scratch]]
        st = bank.get_state()
        bank.close()
        return res, st
    a, sa = run(False)
    b, sb = run(True)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    np.testing.assert_array_equal(sa, sb)


def _twin():
    """tests/hip/libtwin.so: the oracle's recurrence as a serial device kernel (test infrastructure)."""
    import ctypes
    import os
    import subprocess
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip")
    if not os.path.exists(os.path.join(base, "libtwin.so")):
        subprocess.check_call(["make", "-s", "-C", base])
    lib = ctypes.CDLL(os.path.join(base, "libtwin.so"))
    fp, up = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)
    lib.twin_biquad_bank.argtypes = [fp, fp, ctypes.c_size_t, ctypes.c_size_t, fp, fp, up, ctypes.c_size_t]
    lib.twin_biquad_bank.restype = ctypes.c_int

    def run(x, coef, nsec, state):
        C, n = x.shape
        y = np.empty_like(x)
        coef = np.ascontiguousarray(coef, np.float32)
        nsec = np.ascontiguousarray(nsec, np.uint32)
        rc = lib.twin_biquad_bank(y.ctypes.data_as(fp), np.ascontiguousarray(x).ctypes.data_as(fp), C, n, coef.ctypes.data_as(fp),
                                  state.ctypes.data_as(fp), nsec.ctypes.data_as(up), coef.shape[1])
        assert rc == 0, rc
        return y
    return run


def test_device_twin_equals_the_oracle(gpu):
    """The oracle's recurrence, operation for operation and without fused multiply-adds, as a serial kernel on the device
    (tests/hip/serial_biquad.hip, one channel per lane): output and filter memory equal oracle/biquad_oracle.c BIT FOR BIT on
    C2's filters (8 sections, cut-offs from 200 Hz up, three blocks with carried memory).  The device's float32 arithmetic
    is the reference's; what separates the product's time-parallel kernel from the oracle is the order of the operations,
    and the product is held to the same noise rule against this on-device twin as against the CPU oracle."""
    twin = _twin()
    C, n, blocks = 16, 4096, 3
    coef, _ = wl.c2_coefficients(C)
    x = wl.c2_input(C, n, blocks=blocks)
    nsec = np.full(C, 8, np.uint32)
    st_cpu = np.zeros((C, 8, 2), np.float32)
    st_dev = np.zeros((C, 8, 2), np.float32)
    y_gpu, _ = run_bank(gpu, x, list(coef))
    for b in range(blocks):
        ref = oracle.biquad_bank(x[b], coef, nsec, st_cpu)
        dev = twin(x[b], coef, nsec, st_dev)
        np.testing.assert_array_equal(dev, ref)
        np.testing.assert_array_equal(st_dev, st_cpu)
        for c in range(C):
            exact = oracle.biquad_cascade_f64(x[:b + 1, c].reshape(-1), coef[c])[b * n:]
            assert_iir_parity(y_gpu[b, c], dev[c], exact, "product against the device twin, block %d ch %d" % (b, c))


def test_linearity_and_determinism_full_size(gpu):
    """Size-independent properties at full size: same input twice -> identical bits; scaling by 2 is exact."""
    C, n = 1024, 4096
    coef, _ = wl.c2_coefficients(C)
    x = wl.c2_input(C, n, blocks=1)
    y1, _ = run_bank(gpu, x, list(coef))
    y2, _ = run_bank(gpu, x, list(coef))
    np.testing.assert_array_equal(y1, y2)
    y3, _ = run_bank(gpu, x * np.float32(2.0), list(coef))
    np.testing.assert_array_equal(y3, y1 * np.float32(2.0))     # power-of-two scaling commutes with rounding


@pytest.mark.parametrize("seed", range(10))
def test_random_operation_sequences(gpu, seed):
    """Differential stress: per-channel cascades of different lengths re-designed at random (same count: memory kept,
    other count or clear: memory cleared, FilterBank.cpp:233-235), resets, rows switched off, in-place calls and lengths
    that exercise every launch variant and the tail kernel -- against the oracle run over the whole history of a channel."""
    rng = np.random.default_rng(11000 + seed)
    C = 4
    bank = gpu.BiquadBank(C, 8)
    types = [fd.FLT_BT_RLC_BELL, fd.FLT_BT_RLC_HISHELF, fd.FLT_BT_LRX_LOPASS, fd.FLT_MT_RLC_BELL, fd.FLT_BT_BWC_HIPASS]
    coef = [None] * C
    hist = [np.zeros(0, np.float32) for _ in range(C)]       # input since the channel's memory was last cleared ...
    state0 = [None] * C                                      # ... under the current coefficients: oracle state at its start
    state64 = [None] * C                                     # ... and the same in exact (float64) arithmetic
    enabled = [True] * C
    recent = [[np.zeros(0), np.zeros(0)] for _ in range(C)]  # the oracle's last 1024 outputs (float32, float64) while the memory lives

    def redesign(c, clear):
        t = int(rng.choice(types))
        slope = int(rng.integers(1, 3))
        q = wl.design(t, slope, float(rng.uniform(700.0, 15000.0)), 0, float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.1, 2.0)))[:8]
        old = coef[c]
        # memory carried over when the section count is unchanged and no clear was asked for
        keep = (old is not None) and (len(old) == len(q)) and not clear
        if keep:
            if hist[c].size:                                 # (nothing processed since the last re-design: state0 stands)
                _, st = oracle.biquad_cascade(hist[c], old, state0[c])
                state0[c] = st
                _, state64[c] = oracle.biquad_cascade_f64_state(hist[c], old, state64[c])
        else:
            state0[c] = None
            state64[c] = None
            recent[c] = [np.zeros(0), np.zeros(0)]
        hist[c] = np.zeros(0, np.float32)
        coef[c] = q
        bank.set_chains(c, q, clear=clear)
    for c in range(C):
        redesign(c, True)
    for step in range(40):
        op = rng.choice(["process", "process", "process", "redesign", "reset", "toggle", "ir"])
        if op == "process":
            n = int(rng.choice([1, 7, 15, 16, 17, 1000, 1024, 1031, 2048, 2049, 4096, int(rng.integers(1, 6000))]))
            x = rng.standard_normal((C, n)).astype(np.float32)
            din = gpu.DeviceBuffer.from_host(x)
            in_place = bool(rng.integers(0, 2))
            dout = din if in_place else gpu.DeviceBuffer.from_host(np.full((C, n), 9.0, np.float32))
            bank.process(dout, din, n)
            y = dout.download()
            for c in range(C):
                if not enabled[c]:
                    assert np.array_equal(y[c], x[c] if in_place else np.full(n, 9.0, np.float32)), (seed, step, c)
                    continue
                hist[c] = np.concatenate([hist[c], x[c]])
                ref, _ = oracle.biquad_cascade(hist[c], coef[c], state0[c])
                # exact arithmetic follows the filter memory across re-designs that keep it (lfilter with zi is the same
                # transposed direct form II), so the noise rule applies to carried memory as well.  Errors are related to
                # the peak of the channel's last 1024 outputs (across re-designs that keep the memory), and the oracle's noise is
                # read over the same stretch: a call of a few samples has no meaningful peak, or maximum of a round-off walk,
                # of its own.
                exact, _ = oracle.biquad_cascade_f64_state(hist[c], coef[c], state64[c])
                recent[c] = [np.concatenate([recent[c][0], ref[-n:]])[-max(n, 1024):], np.concatenate([recent[c][1], exact[-n:]])[-max(n, 1024):]]
                assert_iir_parity(y[c], ref[-n:], exact[-n:], what=str((seed, step, c, n)),
                                  peak=np.abs(recent[c][1]).max(), noise_over=recent[c])
        elif op == "redesign":
            redesign(int(rng.integers(0, C)), bool(rng.integers(0, 2)))
        elif op == "reset":
            c = int(rng.integers(-1, C))
            bank.reset(None if c < 0 else c)
            for k in (range(C) if c < 0 else [c]):
                hist[k] = np.zeros(0, np.float32); state0[k] = None; state64[k] = None
                recent[k] = [np.zeros(0), np.zeros(0)]
        elif op == "toggle":
            c = int(rng.integers(0, C))
            enabled[c] = not enabled[c]
            bank.set_row_enabled(c, enabled[c])
        else:
            out = gpu.DeviceBuffer((C, 300))
            bank.impulse_response(out, 300)                  # must leave every channel's memory as it was
            h = out.download()
            imp = np.zeros(300, np.float32); imp[0] = 1.0
            for c in range(C):
                if enabled[c]:
                    ref, _ = oracle.biquad_cascade(imp, coef[c])
                    assert float(np.abs(h[c] - ref).max()) <= 2e-5 * max(float(np.abs(ref).max()), 1.0), (seed, step, c)
    bank.close()


# ---- the bank's exact mode (mi_biquad_bank_set_exact): the reference's serial recurrence on the device, through the product --------

def _oracle_calls(x_calls, coef_list, nsec, max_sec):
    """The oracle over a list of calls (each [C][n_i]) with carried filter memory: outputs per call and the final state."""
    C = len(coef_list)
    coef = np.zeros((C, max_sec, 5), np.float32)
    for c, q in enumerate(coef_list):
        coef[c, :len(q)] = q
    state = np.zeros((C, max_sec, 2), np.float32)
    outs = [oracle.biquad_bank(np.ascontiguousarray(x), coef, np.asarray(nsec, np.uint32), state) for x in x_calls]
    return outs, state


@pytest.mark.parametrize("sections,C", [(8, 16), (1, 5), (2, 70), (3, 9), (5, 33), (12, 7), (17, 3), (40, 2), (64, 2), (70, 3), (130, 1)])
def test_exact_mode_is_the_oracle_bit_for_bit(gpu, sections, C):
    """mi_biquad_bank_set_exact(1): FilterBank::process's serial recurrence (FilterBank.cpp:256-291), a section per lane, through the
    product's C-ABI -- output AND filter memory equal oracle/biquad_oracle.c BIT FOR BIT: every lane-group width (1 .. 64 lanes per
    channel), cascades longer than a wave (passes of 64 sections, in place), channels with fewer sections than the bank's longest
    and with none (a copy), calls of 1 .. 5000 samples with the memory carried from call to call, a call in place, a row switched
    off (nothing read, nothing written), and calls in the fast mode in between (the same filter memory serves both)."""
    rng = np.random.default_rng(8800 + sections * 7 + C)
    types = [fd.FLT_BT_LRX_LOPASS, fd.FLT_BT_RLC_BELL, fd.FLT_BT_BWC_HISHELF, fd.FLT_MT_RLC_LOPASS]
    chains, nsec = [], []
    for c in range(C):
        q = []
        while len(q) < sections:
            q.extend(wl.design(types[int(rng.integers(len(types)))], int(rng.integers(1, 5)), float(np.exp(rng.uniform(np.log(60), np.log(15000)))),
                               0, float(10 ** (rng.uniform(-6, 6) / 20)), float(rng.uniform(0.3, 2.0))))
        keep = sections if c != 1 else max(sections - 1 - (sections // 3), 0)      # channel 1: a shorter cascade (none at all for sections 1)
        chains.append(np.asarray(q[:keep], np.float32).reshape(-1, 5))
        nsec.append(keep)
    sizes = [4096, 1, 63, 64, 65, 5000, 200]
    x = [(rng.standard_normal((C, n)) * 0.25).astype(np.float32) for n in sizes]
    ref, st_ref = _oracle_calls(x, chains, nsec, sections)
    bank = gpu.BiquadBank(C, sections)
    for c in range(C):
        bank.set_chains(c, chains[c])
    bank.set_exact(True)
    for k, (xb, n) in enumerate(zip(x, sizes)):
        din = gpu.DeviceBuffer.from_host(xb)
        if k in (3, 5):                                         # calls in place: 64 samples (the start-up steps only) and 5000 (whole chunks, their results stored behind them)
            bank.process(din, din, n)
            got = din.download()
        else:
            dout = gpu.DeviceBuffer((C, n))
            bank.process(dout, din, n)
            got = dout.download()
        assert gpu.last_launch().startswith("(biquad_exact_kernel"), gpu.last_launch()
        np.testing.assert_array_equal(got, ref[k], err_msg="call %d (%d samples)" % (k, n))
    np.testing.assert_array_equal(bank.get_state()[:, :sections], st_ref)
    # a row switched off: its output row and its memory stay
    if C >= 3:
        bank.set_row_enabled(2, False)
        xb = (rng.standard_normal((C, 300)) * 0.25).astype(np.float32)
        marked = gpu.DeviceBuffer.from_host(np.full((C, 300), -5.0, np.float32))
        st_before = bank.get_state()
        bank.process(marked, gpu.DeviceBuffer.from_host(xb), 300)
        y = marked.download()
        assert np.all(y[2] == -5.0)
        np.testing.assert_array_equal(bank.get_state()[2], st_before[2])
        coef = np.zeros((C, sections, 5), np.float32)
        for c, q in enumerate(chains):
            coef[c, :len(q)] = q
        st = st_ref.copy()
        want = oracle.biquad_bank(xb, coef, np.asarray(nsec, np.uint32), st)
        for c in range(C):
            if c != 2:
                np.testing.assert_array_equal(y[c], want[c])
        bank.set_row_enabled(2, True)
    # the fast mode on the same memory, then exact again: both continue from what the other left
    bank.set_exact(False)
    xb = (rng.standard_normal((C, 4096)) * 0.25).astype(np.float32)
    dout = gpu.DeviceBuffer((C, 4096))
    bank.process(dout, gpu.DeviceBuffer.from_host(xb), 4096)
    assert not gpu.last_launch().startswith("(biquad_exact_kernel")
    assert np.isfinite(dout.download()).all()
    bank.close()


def test_exact_mode_process_blocks_and_impulse_response(gpu):
    """The exact mode under mi_biquad_bank_process_blocks (a launch per block: the serial recurrence has no run of blocks to gain
    from) and mi_biquad_bank_impulse_response (FilterBank.cpp:293-330: memory saved, zeroed, restored) -- the oracle's bits."""
    C, n, nb = 24, 4096, 3
    coef, _ = wl.c2_coefficients(C)
    x = wl.c2_input(C, n, blocks=nb)
    nsec = np.full(C, 8, np.uint32)
    state = np.zeros((C, 8, 2), np.float32)
    bank = gpu.BiquadBank(C, 8)
    bank.set_all_chains(coef)
    bank.set_exact(True)
    ins = [gpu.DeviceBuffer.from_host(x[b]) for b in range(nb)]
    outs = [gpu.DeviceBuffer((C, n)) for _ in range(nb)]
    bank.process_blocks(outs, ins, n)
    for b in range(nb):
        np.testing.assert_array_equal(outs[b].download(), oracle.biquad_bank(x[b], coef, nsec, state), err_msg="block %d" % b)
    np.testing.assert_array_equal(bank.get_state(), state)
    ir = gpu.DeviceBuffer((C, 700))
    bank.impulse_response(ir, 700)
    np.testing.assert_array_equal(bank.get_state(), state)
    h = ir.download()
    for c in (0, 7, 23):
        np.testing.assert_array_equal(h[c], oracle.biquad_impulse_response(700, coef[c], np.zeros((8, 2), np.float32)))
    bank.close()


def test_c2_full_size_all_channels_exact_mode(gpu):
    """BASELINE config 1 at full size through the product's exact mode: 1024 ch x 4096, 8 sections, 64 consecutive blocks with carried
    state, EVERY channel against the oracle under north_star's own rule -- |gpu - oracle| <= 1e-5 of the block's peak, no noise
    allowance -- and, beyond it, bit for bit; the filter memory after the 64 blocks as well."""
    from conftest import TOL
    C, n, nb = 1024, 4096, 64
    coef, _ = wl.c2_coefficients(C, seed=3)
    x = wl.c2_input(C, n, blocks=nb, seed=2)
    state = np.zeros((C, 8, 2), np.float32)
    nsec = np.full(C, 8, np.uint32)
    bank = gpu.BiquadBank(C, 8)
    bank.set_all_chains(coef)
    bank.set_exact(True)
    din, dout = gpu.DeviceBuffer((C, n)), gpu.DeviceBuffer((C, n))
    worst, differing = 0.0, 0
    for b in range(nb):
        din.upload(x[b])
        bank.process(dout, din, n)
        y = dout.download()
        ref = oracle.biquad_bank(x[b], coef, nsec, state)
        peak = np.maximum(np.abs(ref).max(axis=1), 1e-30)
        worst = max(worst, float((np.abs(y - ref).max(axis=1) / peak).max()))
        differing += int((y != ref).any(axis=1).sum())
    note("C2 full size in the bank's exact mode (mi_biquad_bank_set_exact): 1024 channels x 64 blocks, worst |gpu - oracle| / peak = %.1e, "
         "channel-blocks that differ from the oracle in any bit: %d of %d" % (worst, differing, C * nb))
    assert worst <= TOL, worst
    assert differing == 0, differing
    np.testing.assert_array_equal(bank.get_state(), state)
    bank.close()


def test_c1_readme_filter_in_the_exact_mode_by_default_switch(gpu):
    """BASELINE config 0 (1 ch x 48000, FLT_BT_BWC_HISHELF slope 2 @1 kHz +6 dB, README.md:176-191) with the process-wide switch
    mi_dspu_set_exact_iir_default(1) -- how objects of the class layer are put into the exact mode: a bank created afterwards runs
    biquad_exact_kernel and gives the oracle's output and filter memory bit for bit, in place, in three uneven calls; a bank
    created after the switch is cleared is a fast one again."""
    gain = float(np.float32(np.exp(np.float32(6.0) * np.float32(np.log(10.0)) * np.float32(0.05))))
    bq = wl.design(fd.FLT_BT_BWC_HISHELF, 2, 1000.0, 1000.0, gain, 0.0)
    x = (np.random.default_rng(1).standard_normal((1, 48000)) * 0.25).astype(np.float32)
    ref, st = oracle.biquad_cascade(x[0], bq, None)
    gpu.check(gpu.lib.mi_dspu_set_exact_iir_default(1))
    try:
        bank = gpu.BiquadBank(1, 2)
    finally:
        gpu.check(gpu.lib.mi_dspu_set_exact_iir_default(0))
    bank.set_chains(0, bq)
    buf = gpu.DeviceBuffer.from_host(x)
    pos = 0
    for n in (4096, 40000, 3904):
        seg = gpu.DeviceBuffer.from_host(np.ascontiguousarray(x[:, pos:pos + n]))
        bank.process(seg, seg, n)                               # in place
        assert gpu.last_launch().startswith("(biquad_exact_kernel"), gpu.last_launch()
        np.testing.assert_array_equal(seg.download()[0], ref[pos:pos + n])
        pos += n
    np.testing.assert_array_equal(bank.get_state()[0], np.asarray(st, np.float32).reshape(2, 2))
    bank.close()
    fast = gpu.BiquadBank(1, 2)
    fast.set_chains(0, bq)
    fast.process(buf, buf, 48000)
    assert not gpu.last_launch().startswith("(biquad_exact_kernel")
    fast.close()


def test_exact_mode_by_the_environment_switch(gpu, monkeypatch):
    """MI_DSPU_EXACT_IIR=1, read when a bank is created: a host that cannot be changed to call mi_dspu_set_exact_iir_default gets the
    reference's bits; without the variable (or with 0) a bank is a fast one."""
    gain = float(np.float32(np.exp(np.float32(6.0) * np.float32(np.log(10.0)) * np.float32(0.05))))
    bq = wl.design(fd.FLT_BT_BWC_HISHELF, 2, 1000.0, 1000.0, gain, 0.0)
    x = (np.random.default_rng(11).standard_normal((1, 8192)) * 0.25).astype(np.float32)
    ref, _ = oracle.biquad_cascade(x[0], bq, None)
    for value, exact in (("1", True), ("0", False), (None, False)):
        if value is None:
            monkeypatch.delenv("MI_DSPU_EXACT_IIR", raising=False)
        else:
            monkeypatch.setenv("MI_DSPU_EXACT_IIR", value)
        bank = gpu.BiquadBank(1, 2)
        bank.set_chains(0, bq)
        out = gpu.DeviceBuffer((1, 8192))
        bank.process(out, gpu.DeviceBuffer.from_host(x), 8192)
        assert gpu.last_launch().startswith("(biquad_exact_kernel") == exact, (value, gpu.last_launch())
        if exact:
            np.testing.assert_array_equal(out.download()[0], ref)
        bank.close()
