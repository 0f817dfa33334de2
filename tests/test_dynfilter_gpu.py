"""GPU parity of mi_dynfilter_bank_* (lsp::dspu::DynamicFilters) against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

import oracle
from oracle import dynamic_filters as df
from oracle import filter_design as fd
from conftest import IIR_EXACT_FACTOR, IIR_REF_FACTOR, NOISE_FLOOR, TOL

pytestmark = pytest.mark.gpu
SR = 48000
TYPES = [t for t in range(1, len(fd.FILTER_TYPES)) if df.cascade_count(t, 1) > 0]


def gains(rng, C, n, kind):
    """Per-channel gain curves: what a dynamics processor feeds a dynamic equaliser."""
    t = np.arange(n) / float(SR)
    if kind == "constant":
        g = np.full((C, n), 1.8)
    elif kind == "sweep":                                    # slow envelope, 0.3 .. 3
        g = np.exp(np.log(3.0) * np.sin(2 * np.pi * (3.0 + rng.uniform(0, 4, (C, 1))) * t + rng.uniform(0, 6, (C, 1))))
    else:                                                    # sample-to-sample changes
        g = np.exp(rng.uniform(-1.0, 1.0, (C, n)))
    return g.astype(np.float32)


def check(y, ref, exact, what, coef_tol=0.0):
    """conftest.assert_iir_parity's rule (the recursion's own float32 noise is the yardstick).
    coef_tol: how far the coefficient sets of the device and of the oracle may differ (matched-Z types: the float
    amplitude normalisation next to z = 1, Filter.cpp:2369-2411, turns the last bits of expf into 1e-4 of a numerator --
    tests/test_oracle_dynamic_filters.py::test_product_builders_match_the_oracle shows the same between two host libms)."""
    peak = max(float(np.abs(exact).max()), 1e-30)
    noise = float(np.abs(ref - exact).max()) / peak
    e32 = float(np.abs(y - ref).max()) / peak
    e64 = float(np.abs(y - exact).max()) / peak
    msg = "%s: vs oracle %.2e, vs float64 %.2e, oracle's own noise %.2e" % (what, e32, e64, noise)
    assert np.all(np.isfinite(y)), msg
    # the coefficients themselves come from two libms (device / numpy): one part in 1e6 of the coefficients on top of the
    # recursion's noise
    if noise <= NOISE_FLOOR:
        assert e32 <= 2 * TOL + coef_tol, msg
    else:
        assert e64 <= max(2 * TOL, IIR_EXACT_FACTOR * noise) + coef_tol and e32 <= max(2 * TOL, IIR_REF_FACTOR * noise) + coef_tol, msg
    return e32, noise


@pytest.mark.parametrize("t", TYPES)
def test_every_type_matches_the_oracle(gpu, t):
    """Three blocks with carried memory, a slowly moving gain per channel, 3 channels; bilinear and matched types."""
    rng = np.random.default_rng(100 + t)
    C, n, blocks = 3, 1500, 3
    slope = 2 if df.cascade_count(t, 2) <= 16 else 1
    bank = gpu.DynFilterBank(C, 2)
    bank.set_sample_rate(SR)
    bank.set_params(1, t, slope, 1200.0, 5000.0, 1.0, 0.6)
    bank.set_filter_active(1)
    refs = [df.DynamicFilters(2) for _ in range(C)]
    for r in refs:
        r.set_sample_rate(SR)
        r.set_params(1, t, slope, 1200.0, 5000.0, 1.0, 0.6)
        r.set_filter_active(1, True)
    x = (rng.standard_normal((C, n * blocks)) * 0.25).astype(np.float32)
    g = gains(rng, C, n * blocks, "sweep")
    y = np.empty_like(x)
    ref, exact = np.empty_like(x), np.empty(x.shape, np.float64)
    for b in range(blocks):
        seg = slice(b * n, (b + 1) * n)
        din, dg, dout = gpu.DeviceBuffer.from_host(x[:, seg]), gpu.DeviceBuffer.from_host(g[:, seg]), gpu.DeviceBuffer((C, n))
        bank.process(1, dout, din, dg, n)
        y[:, seg] = dout.download()
        for c in range(C):
            ref[c, seg], exact[c, seg] = refs[c].process(1, x[c, seg], g[c, seg], exact=True)
    for c in range(C):
        check(y[c], ref[c], exact[c], "%s ch %d" % (fd.FILTER_TYPES[t], c), coef_tol=0.0 if (t & 1) else 1e-3)
    bank.close()


@pytest.mark.parametrize("kind", ["constant", "sweep", "jumpy"])
def test_gain_shapes_sizes_and_in_place(gpu, kind):
    """Constant, slowly varying and sample-to-sample gains; ragged call sizes around the 1024-sample block and the
    16-sample chunk; in-place calls; padded rows."""
    rng = np.random.default_rng(7)
    C = 5
    t = fd.FLT_BT_LRX_BELL
    bank = gpu.DynFilterBank(C, 1)
    bank.set_sample_rate(SR)
    bank.set_params(0, t, 2, 900.0, 900.0, 1.0, 1.0)
    bank.set_filter_active(0)
    refs = [df.DynamicFilters(1) for _ in range(C)]
    for r in refs:
        r.set_sample_rate(SR); r.set_params(0, t, 2, 900.0, 900.0, 1.0, 1.0); r.set_filter_active(0, True)
    sizes = [1, 15, 16, 17, 1023, 1024, 1025, 2048, 3000, 4096, 77]
    total = sum(sizes)
    x = (rng.standard_normal((C, total)) * 0.25).astype(np.float32)
    g = gains(rng, C, total, kind)
    y = np.empty_like(x)
    pos = 0
    for i, n in enumerate(sizes):
        pad = 8 if (i & 1) else 0
        hx = np.zeros((C, n + pad), np.float32); hx[:, :n] = x[:, pos:pos + n]
        hg = np.ones((C, n + pad), np.float32); hg[:, :n] = g[:, pos:pos + n]
        din, dg = gpu.DeviceBuffer.from_host(hx), gpu.DeviceBuffer.from_host(hg)
        dout = din if (i % 3 == 0) else gpu.DeviceBuffer((C, n + pad))
        bank.process(0, dout, din, dg, n, n + pad, n + pad, n + pad)
        y[:, pos:pos + n] = dout.download()[:, :n]
        pos += n
    for c in range(C):
        ref, exact = refs[c].process(0, x[c], g[c], exact=True)
        check(y[c], ref, exact, "%s ch %d" % (kind, c))
    bank.close()


def test_constant_gain_equals_the_static_bank(gpu):
    """A constant gain vector filters every sample with the same sections: the dynamic bank then computes what the static
    biquad bank computes with those sections (both within round-off of the oracle; identical operation order per sample)."""
    rng = np.random.default_rng(9)
    C, n = 4, 4096
    t, slope = fd.FLT_BT_RLC_BELL, 3
    bank = gpu.DynFilterBank(C, 1)
    bank.set_sample_rate(SR); bank.set_params(0, t, slope, 2000.0, 2000.0, 1.0, 0.8); bank.set_filter_active(0)
    sec = gpu.dynfilter_sections(t, slope, 2000.0, 2000.0, 0.8, 2.2, SR)
    static = gpu.BiquadBank(C, len(sec))
    for c in range(C):
        static.set_chains(c, sec)
    x = (rng.standard_normal((C, n)) * 0.25).astype(np.float32)
    din, dg = gpu.DeviceBuffer.from_host(x), gpu.DeviceBuffer.from_host(np.full((C, n), 2.2, np.float32))
    d1, d2 = gpu.DeviceBuffer((C, n)), gpu.DeviceBuffer((C, n))
    bank.process(0, d1, din, dg, n)
    static.process(d2, din, n)
    y1, y2 = d1.download(), d2.download()
    for c in range(C):
        ref, _ = oracle.biquad_cascade(x[c], sec)
        exact = oracle.biquad_cascade_f64(x[c], sec)
        check(y1[c], ref, exact, "dynamic ch %d" % c)
        check(y2[c], ref, exact, "static ch %d" % c)
        assert np.abs(y1[c] - y2[c]).max() <= 2 * TOL * np.abs(exact).max()
    bank.close(); static.close()


def test_bypass_clear_and_bad_arguments(gpu):
    """inactive / FLT_NONE / slope 0: a copy (DynamicFilters.cpp:207-212); a change of type clears every filter's memory at
    the next process() (:132-133, 214-219); set_filter_active always activates (.h:147-153); refusals."""
    rng = np.random.default_rng(3)
    C, n = 2, 600
    x = (rng.standard_normal((C, n)) * 0.25).astype(np.float32)
    g = np.full((C, n), 2.0, np.float32)
    bank = gpu.DynFilterBank(C, 2)
    bank.set_sample_rate(SR)
    din, dg, dout = gpu.DeviceBuffer.from_host(x), gpu.DeviceBuffer.from_host(g), gpu.DeviceBuffer((C, n))
    bank.process(0, dout, din, dg, n)                        # FLT_NONE, inactive
    np.testing.assert_array_equal(dout.download(), x)
    bank.set_params(0, fd.FLT_BT_RLC_LOSHELF, 0, 500.0, 500.0, 1.0, 0.0); bank.set_filter_active(0, False)
    assert bank.get_params(0)[1] is True                     # "false" activates as well
    bank.process(0, dout, din, dg, n)                        # slope 0
    np.testing.assert_array_equal(dout.download(), x)
    # memory: two filters run, then filter 1 changes its type -> both memories are cleared at the next process()
    refs = [df.DynamicFilters(2) for _ in range(C)]
    def both(name, *a):
        getattr(bank, name)(*a)
        for r in refs:
            getattr(r, name)(*a)
    both("set_sample_rate", SR)
    both("set_params", 0, fd.FLT_BT_RLC_LOSHELF, 2, 500.0, 500.0, 1.0, 0.0); both("set_filter_active", 0, True)
    both("set_params", 1, fd.FLT_BT_BWC_HIPASS, 3, 300.0, 300.0, 1.0, 0.0); both("set_filter_active", 1, True)
    for step in range(3):
        if step == 2:
            both("set_params", 1, fd.FLT_MT_RLC_BELL, 2, 3000.0, 3000.0, 1.0, 0.5)
        for fid in (0, 1):
            bank.process(fid, dout, din, dg, n)
            y = dout.download()
            for c in range(C):
                ref, exact = refs[c].process(fid, x[c], g[c], exact=True)
                check(y[c], ref, exact, "step %d filter %d ch %d" % (step, fid, c), coef_tol=1e-3 if (step == 2 and fid == 1) else 0.0)
    with pytest.raises(gpu.MiError):
        bank.set_params(5, fd.FLT_BT_RLC_BELL, 1, 1000.0, 1000.0, 1.0, 0.0)
    with pytest.raises(gpu.MiError):
        bank.set_params(0, fd.FLT_BT_RLC_ENVELOPE, 1, 1000.0, 1000.0, 1.0, 0.0)
    with pytest.raises(gpu.MiError):
        bank.process(0, dout, din, None, n)                  # an active filter needs its gain rows
    bank.close()
