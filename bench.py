#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X lsp-dsp-units hot path.

    python bench.py --gpus N --steps K --warmup W [--workload biquad]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  biquad    : BASELINE.json configs[1] -- 1024 channels x 4096-sample block through an 8-stage biquad
              cascade (FilterBank::process), state carried from block to block.  (default)
With N > 1 (launched by torch.distributed.run, one process per GPU) the channels are sharded: every rank
owns its own 1024 channels (weak scaling, no data-path collective: channels are independent).

Prints ONE JSON line on rank 0.  After the W warm-up steps the K-step region (barrier + synchronize on both sides,
max over ranks) is timed `--regions` times back to back and `value` is whole-job Msamples/s of the MEDIAN region, so
a short region (the driver's K = 20 is a quarter of a millisecond) does not rest on one sample; every region's time is
in `region_ms`.  The K steady-state calls of a region are captured once into a hipGraph and replayed (the calls take
no host decision; `launch` says which mode ran).  `roofline` is priced from the dominant kernel's average duration
over >= 16 launches that carry their own HIP event pair (hipExtLaunchKernelGGL on the launch stream), taken in an
untimed pass after the regions.  `cpu_baseline` is the reference's x8 structure built -O3 -march=native for this
host (oracle/cpu_baseline), on one core and on all cores, on a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


KERNEL_PROBES = 16         # launches with their own event pair, after the timed regions
FLT_BT_LRX_LOPASS = 47     # filter_type_t, include/lsp-plug.in/dsp-units/filters/common.h:95


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="all", choices=["all", "biquad", "convolver", "equalizer", "spectral", "stft", "crossover", "splitter", "loudness", "dynfilter"],
                    help="all = headline biquad line with the convolver result attached under \"convolver\"")
    ap.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    ap.add_argument("--samples", type=int, default=4096, help="samples per block")
    ap.add_argument("--ring", type=int, default=16, help="distinct resident blocks cycled through "
                    "(16 x 32 MiB in+out > the 256 MiB Infinity Cache, so steps stream from HBM)")
    ap.add_argument("--regions", type=int, default=0, help="timed repetitions of the K-step region (0 = 25, or 5 when K >= 500)")
    ap.add_argument("--launch", default="blocks", choices=["blocks", "graph", "eager"],
                    help="how the K steps of a region are issued: blocks = one mi_biquad_bank_process_blocks call (the K blocks ride "
                         "one launch of biquad_stream_kernel; `value` is this), graph = one hipGraph of K process() calls (always "
                         "reported beside it under \"per_call\"), eager = K process() calls from Python")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sections", type=int, default=8, help="experiment knob: keep only the first N sections")
    ap.add_argument("--conv-channels", type=int, default=256, help="convolver channels per GPU")
    ap.add_argument("--eq-channels", type=int, default=256, help="equalizer channels per GPU (config 3: 2048 over 8 GPUs)")
    ap.add_argument("--spec-channels", type=int, default=1024, help="analyzer channels per GPU (config 4: 8192 over 8 GPUs)")
    ap.add_argument("--split-channels", type=int, default=256, help="splitter row: channels per GPU")
    ap.add_argument("--call", type=int, default=0, help="convolver workload: also time a stream of calls of this many samples "
                    "(e.g. 256: what a plugin host does; a step is still one 4096-sample frame = 4096 / call calls)")
    ap.add_argument("--no-stream-pair", action="store_true", help="meters row: skip the extra measurement with the two banks on "
                    "a stream each (kernel profiles of the row then hold the one-stream launches only)")
    ap.add_argument("--conv-steps", type=int, default=192, help="steps per timed region of the sub-workloads (a multiple of their launches' 16 / 64 / 128 units)")
    ap.add_argument("--conv-warmup", type=int, default=10)
    return ap.parse_args()


def _effective_cores():
    """CPUs this process can really use: the affinity mask, cut down to the cgroup's CPU quota (a container limited to
    8 CPUs of a 256-thread host still sees 256 in sched_getaffinity)."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                                    # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                                # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(math.ceil(quota))))
    return n


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_biquad(coef, samples, budget_s=5.0):
    """The reference's structure on this host (oracle/cpu_baseline/biquad_x8_host.c): one x8 software-pipelined pass per
    channel and block (FilterBank.cpp:267-273), SIMD across the eight sections, four channels interleaved per thread,
    persistent threads over fixed channel ranges, all blocks of a run inside one parallel region.  Rebuilt here with
    -O3 -march=native for the machine the bench runs on; timed on one core and on all cores."""
    import ctypes
    import subprocess
    import numpy as np
    import workloads as wl
    base = os.path.join(ROOT, "oracle", "cpu_baseline")
    subprocess.check_call(["make", "-s", "-B", "-C", base])
    lib = ctypes.CDLL(os.path.join(base, "libcpubase.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    lib.cpu_biquad_x8_run.argtypes = [fp, fp] + [ctypes.c_size_t] * 4 + [fp, fp, ctypes.c_int]
    C, ring = coef.shape[0], 2
    x = wl.c2_input(C, samples, blocks=ring, seed=12)
    y = np.empty_like(x)
    coef = np.ascontiguousarray(coef, np.float32)
    f = lambda a: a.ctypes.data_as(fp)
    cores = _effective_cores()

    def timed(threads, blocks):
        st = np.zeros((C, 8, 2), np.float32)
        t0 = time.perf_counter()
        used = lib.cpu_biquad_x8_run(f(y), f(x), C, samples, blocks, ring, f(coef), f(st), threads)
        return used, time.perf_counter() - t0

    res = {}
    for name, threads in (("one_core", 1), ("all_cores", cores)):
        used, dt = timed(threads, 64)                               # >= 64 blocks per parallel region; sizes the run
        blocks = int(min(max(64, 64 * budget_s / max(dt, 1e-6)), 1 << 16))
        used, dt = timed(threads, blocks)
        res[name] = {"value": round(blocks * C * samples / dt / 1e6, 1), "threads": used, "blocks": blocks,
                     "seconds": round(dt, 2)}
    allc = res["all_cores"]
    return {
        "value": allc["value"], "unit": "Msamples/s", "cores": allc["threads"], "kind": "port",
        "cores_note": "threads = CPUs this process may use: affinity mask (%d) cut to the cgroup CPU quota" %
                      (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)),
        "per_core": round(allc["value"] / max(1, allc["threads"]), 1),
        "one_core": {"value": res["one_core"]["value"], "unit": "Msamples/s", "cores": 1},
        "cpu_model": _cpu_model(), "build": "gcc -O3 -march=native -ffp-contract=fast -fopenmp (rebuilt on this host)",
        "simd": "x8: eight sections side by side in vector registers",
        "sample": "%d blocks (all cores) / %d (one core) of %d ch x %d samples, x8 pipelined pass per channel and block (FilterBank.cpp:267-273), "
                  "SIMD across the sections; lsp-dsp-lib's own kernels are not available offline"
                  % (allc["blocks"], res["one_core"]["blocks"], C, samples),
    }


def _probe_mean(kernel_ms):
    """Plain mean of the launches that carried an event pair (round 4 trimmed the two lowest and two highest of 16 and so
    did not follow from rocprofv3's average of the same command: VERDICT r04 "weak" 5)."""
    return sum(kernel_ms) / len(kernel_ms)


def _roofline(kernel, alg_bytes, kernel_ms, step_ms, probe_mode, traffic=None, extra=None, launch_steps=1):
    """The `roofline` object of a (sub-)result.  alg_bytes: algorithmic bytes of ONE launch of `kernel` (= of `launch_steps`
    steps); kernel_ms: that launch's durations from the probe pass of _timed_steps (HIP events at the kernel's own begin and
    end, on the launch stream); step_ms: the timed region's time per step.  A launch cannot outlast the steps it carries:
    when the probes say so even after the second (back-to-back) pass, or there are none, the fraction is priced from the
    timed region itself (`source` says which) -- never null."""
    whole = alg_bytes / launch_steps / (step_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic, "kernel": kernel,
         "algorithmic_bytes_per_launch": alg_bytes, "steps_per_launch": launch_steps,
         "whole_step_frac": round(whole / HBM_PEAK_GBS, 4)}
    ks = sorted(kernel_ms)
    if not ks or probe_mode == "inconsistent":
        r.update({"achieved": round(whole, 1), "frac": round(whole / HBM_PEAK_GBS, 4), "source": "whole_step",
                  "reason": "no usable probe pass (%s): priced from the timed region, launch gaps included" %
                            ("kernel_avg_us %.2f > %.2f us of steps x 1.05 in both probe passes" %
                             (_probe_mean(kernel_ms) * 1e3, step_ms * launch_steps * 1e3) if ks else "probes not taken")})
    else:
        avg_ms = _probe_mean(kernel_ms)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        r.update({"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4), "source": "kernel_events",
                  "kernel_avg_us": round(avg_ms * 1e3, 3), "kernel_median_us": round(ks[len(ks) // 2] * 1e3, 3),
                  "kernel_us_per_step": round(avg_ms * 1e3 / launch_steps, 3), "kernel_samples": len(kernel_ms),
                  # (all probes, untrimmed: a rocprofv3 --stats average of the same command also covers the launches of the
                  # timed regions, the first of which run in the power controller's onset dip -- timing.region_ms.in_order)
                  "kernel_min_mean_max_us": [round(ks[0] * 1e3, 3), round(sum(ks) / len(ks) * 1e3, 3), round(ks[-1] * 1e3, 3)],
                  "kernel_avg_of": "plain mean of %d launches with an event pair of their own" % len(ks),
                  "probe": probe_mode})
    if extra:
        r.update(extra)
    return r


def _biquad_issue_side(kernel_ms, C, n, sections, launch_steps, counters, clock_ghz=None):
    """The second roof of the biquad kernels next to the HBM one: VALU issue.  Instructions per launch from the committed SQ
    counters of the SAME kernel (`counters`: a file under profiles/ holding SQ_INSTS_VALU per launch and the blocks that
    launch carried), one wave64 VALU instruction per 4 clocks and SIMD (`v_pk_fma_f32` included: MI355X_MICROARCH.md,
    157.3 TFLOP/s = 1024 SIMDs x 2.4 GHz x 16 lanes x 2 (packed) x 2 (fma)), 1024 SIMDs, 2.4 GHz."""
    sq = _committed_json(counters) or {}
    insts, per = sq.get("SQ_INSTS_VALU"), sq.get("blocks_per_launch", 1)
    if not insts or not kernel_ms or (C, n, sections) != (1024, 4096, 8):
        return {}
    if not _sources_current(sq.get("sources"), counters):
        return {}
    avg_s = _probe_mean(kernel_ms) * 1e-3 / launch_steps
    floor_s = float(insts) / per * 4.0 / 1024.0 / 2.4e9
    r = {"valu_issue_frac": round(floor_s / avg_s, 4), "valu_insts_per_block": round(float(insts) / per),
         "valu_issue_floor_us_per_block": round(floor_s * 1e6, 2), "hbm_floor_us_per_block": round(8.0 * C * n / 6.29e12 * 1e6, 2),
         "counters_from": "profiles/" + counters}
    if clock_ghz:
        # the same count at the shader clock the launch HELD (mi_dspu_last_stream_clock: the part lowers its clock under dense packed
        # arithmetic), not the nominal 2.4 GHz
        r["shader_clock_ghz"] = round(clock_ghz, 3)
        r["valu_issue_frac_at_clock"] = round(floor_s * 2.4 / clock_ghz / avg_s, 4)
    return r



def _issue_side(kernel, kernel_ms, launch_units, counters="r06_kernels_pmc_sq.json"):
    """The second roof of a launch next to the HBM one: vector issue.  VALU instructions per unit (block / frame) of `kernel`
    from the committed rocprofv3 --pmc counts (profiles/<counters>, tests/prof_valu.sh), one wave64 VALU instruction per 4 clocks
    and SIMD, 1024 SIMDs, 2.4 GHz: valu_issue_frac = that floor / the launch's measured time per unit.  A launch far below BOTH
    roofs (the transforms: 0.3 - 0.4 of the issue rate at a fraction of the HBM rate) is bound by neither -- by its passes
    through LDS and the barriers between them (profiles/r05_experiments/fft_kernels_issue_and_occupancy.txt)."""
    doc = _committed_json(counters) or {}
    for name, d in (doc.get("kernels") or {}).items():
        if kernel in name and kernel_ms:
            if not _sources_current(d.get("sources"), "%s: %s" % (counters, kernel)):
                return {}
            per_unit_s = _probe_mean(kernel_ms) * 1e-3 / launch_units
            floor_s = float(d["valu_per_unit"]) * 4.0 / 1024.0 / 2.4e9
            return {"valu_issue_frac": round(floor_s / per_unit_s, 4), "valu_insts_per_unit": round(float(d["valu_per_unit"])),
                    "valu_issue_floor_us_per_unit": round(floor_s * 1e6, 3), "counters_from": "profiles/" + counters}
    return {}


def _pmc_traffic(name, kernel=None, units=1):
    """HBM bytes per launch measured with rocprofv3 --pmc (committed under profiles/), or None.  kernel: the summary must be
    of that kernel; units: what this run's launch carries (blocks of a biquad_stream_kernel launch) -- the summary holds the
    bytes per unit of the launch it was taken from."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        if kernel is not None and kernel not in d.get("kernel", ""):
            return None
        if not _sources_current(d.get("sources"), name):
            return None
        return d["hbm_bytes_per_unit"] * units if "hbm_bytes_per_unit" in d else d.get("hbm_bytes_per_launch")
    except Exception:
        return None


_STALE = set()


def _sources_current(recorded, what):
    """A committed counter summary is quoted only while the library this run loaded was built from the sources the summary was
    measured on (the summary's "sources": file -> sha, tests/prof_sources.py; the library's: mi_dspu_source_sha).  A summary
    without hashes, or with other ones, is stale: its figure stays out of the line and the line says so ("stale_counters")."""
    try:
        mi = importlib.import_module("lsp-dsp-units_amd")
        ok = bool(recorded) and all(mi.source_sha(f) == sha for f, sha in recorded.items())
    except Exception:
        ok = False
    if not ok:
        _STALE.add(what)
    return ok


def _committed_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def _profile_avg_us(workload, kernel):
    """Average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary (profiles/), or None.
    rocprofv3 stamps a dispatch from the moment its packet is picked up, so with back-to-back launches its figure is the
    step period (kernel + dispatch gap); the live HIP events bracket the execution only."""
    import csv
    import glob
    try:
        fn = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_stats.csv" % workload)))[-1]
        for row in csv.DictReader(open(fn)):
            if kernel in row["Name"]:
                return round(float(row["AverageNs"]) / 1e3, 3)
    except Exception:
        pass
    return None


def cpu_baseline_convolver(irs, frame, budget_s=6.0):
    """The reference's Convolver::process on this host (oracle/cpu_baseline/fft_units_host.c: cpu_convolver_bank_* -- the oracle's
    restatement of the non-uniform partitioned algorithm, Convolver.cpp:217-313, one object per channel, on the vectorised
    dsp:: primitives of fft_simd_host.c, OpenMP over the channels; -O3 -march=native, rebuilt here)."""
    import ctypes
    import numpy as np
    lib = _cpubase()
    fp = ctypes.POINTER(ctypes.c_float)
    lib.cpu_convolver_bank_create.restype = ctypes.c_void_p
    lib.cpu_convolver_bank_create.argtypes = [fp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    lib.cpu_convolver_bank_run.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    lib.cpu_convolver_bank_destroy.argtypes = [ctypes.c_void_p]
    f = lambda a: a.ctypes.data_as(fp)
    cores = _effective_cores()
    C, taps, ring = irs.shape[0], irs.shape[1], 2
    irs = np.ascontiguousarray(irs, np.float32)
    x = np.random.default_rng(5).standard_normal((ring, C, frame)).astype(np.float32)
    y = np.empty_like(x)
    bank = lib.cpu_convolver_bank_create(f(irs), taps, C, 13, cores)
    used = [1]

    def timed(frames):
        t0 = time.perf_counter()
        used[0] = lib.cpu_convolver_bank_run(bank, f(y), f(x), frame, frames, ring, cores)
        return time.perf_counter() - t0
    timed(2)                                                # (the ring of frame images fills)
    rate, frames = _cpu_sized_run(timed, C * frame, budget_s, first=2)
    lib.cpu_convolver_bank_destroy(bank)
    assert np.isfinite(y).all() and float(np.abs(y).max()) > 0.0
    return {
        "value": round(rate / 1e6, 2), "unit": "Msamples/s", "cores": used[0], "kind": "port", "simd": "gcc -O3 -march=native vector code",
        "cpu_model": _cpu_model(),
        "sample": "%d frames of %d channels x 4096 samples, 65536-tap IR per channel, rank 13: the reference's non-uniform partitioned "
                  "algorithm (Convolver.cpp:217-313, C restatement) on vectorised four-step FFT primitives, OpenMP over the channels" % (frames, C),
    }


def _cpubase():
    """oracle/cpu_baseline rebuilt -O3 -march=native for this host (measurement infrastructure): the reference's block logic on
    vectorised restatements of the dsp:: primitives -- lsp-dsp-lib's own SIMD kernels are un-vendored (modules.mk:29-33)."""
    import ctypes
    import subprocess
    base = os.path.join(ROOT, "oracle", "cpu_baseline")
    subprocess.check_call(["make", "-s", "-B", "-C", base])
    return ctypes.CDLL(os.path.join(base, "libcpubase.so"))


def _cpu_sized_run(timed, unit_count, budget_s, first=4):
    """timed(reps) -> seconds: sizes a run to about budget_s from a short one and returns (units per second, reps)."""
    dt = timed(first)
    reps = int(min(max(first, first * budget_s / max(dt, 1e-6)), 1 << 20))
    dt = timed(reps)
    return unit_count * reps / dt, reps


def cpu_baseline_equalizer(C, n, budget_s=5.0):
    """The reference's Equalizer::process in EQM_FIR mode on this host (oracle/cpu_baseline/fft_units_host.c:
    Equalizer.cpp:460-571 -- per 4096 samples one fastconv_parse_apply of rank 13 -- on the oracle's scalar FFT primitives,
    -O3 -march=native, OpenMP over the channels, one object per channel)."""
    import ctypes
    import numpy as np
    lib = _cpubase()
    fp = ctypes.POINTER(ctypes.c_float)
    lib.cpu_equalizer_fir_run.argtypes = [fp, fp] + [ctypes.c_size_t] * 4 + [fp, fp, ctypes.c_int]
    lib.cpu_fastconv_image.argtypes = [fp, fp, ctypes.c_size_t]
    f = lambda a: a.ctypes.data_as(fp)
    fir_rank, ring = 12, 2
    rng = np.random.default_rng(6)
    fir = (rng.standard_normal((C, n)) * np.exp(-np.arange(n) / 512.0)).astype(np.float32)   # any FIR: the cost does not depend on it
    conv = np.zeros((C, 4 * n), np.float32)
    for c in range(C):
        lib.cpu_fastconv_image(f(conv[c]), f(fir[c]), fir_rank)
    x = (np.random.default_rng(60).standard_normal((ring, C, n)) * 0.25).astype(np.float32)
    y = np.empty_like(x)
    cores = _effective_cores()
    used = [1]

    def timed(blocks):
        st = np.zeros((C, 4 * n), np.float32)
        t0 = time.perf_counter()
        used[0] = lib.cpu_equalizer_fir_run(f(y), f(x), C, fir_rank, blocks, ring, f(conv), f(st), cores)
        return time.perf_counter() - t0
    rate, blocks = _cpu_sized_run(timed, C * n, budget_s)
    assert np.isfinite(y).all() and float(np.abs(y).max()) > 0.0
    return {"value": round(rate / 1e6, 2), "unit": "Msamples/s", "cores": used[0], "kind": "port", "cpu_model": _cpu_model(),
            "simd": "gcc -O3 -march=native vector code",
            "sample": "%d blocks of %d ch x %d samples, Equalizer EQM_FIR fir_rank 12 (Equalizer.cpp:460-571: one 8192-point "
                      "fastconv_parse_apply per block and channel), vectorised four-step FFT primitives, OpenMP over the channels" % (blocks, C, n)}


def cpu_baseline_analyzer(C, rank_fft, hop, budget_s=5.0):
    """The reference's Analyzer::process on this host (oracle/cpu_baseline/fft_units_host.c: Analyzer.cpp:299-409 -- ring
    ingest, one Hann-windowed 4096-point packed_direct_fft per channel and period, pcomplex_mod, mix2 -- scalar C, OpenMP)."""
    import ctypes
    import numpy as np
    lib = _cpubase()
    fp = ctypes.POINTER(ctypes.c_float)
    lib.cpu_analyzer_run.argtypes = [fp, fp] + [ctypes.c_size_t] * 5 + [fp, ctypes.c_float, fp, ctypes.c_size_t, ctypes.c_int]
    f = lambda a: a.ctypes.data_as(fp)
    n, ring = 1 << rank_fft, 2
    bufsize = 2 * n
    i = np.arange(n, dtype=np.float64)
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * i / (n - 1))).astype(np.float32)      # windows.cpp:152-155
    x = np.random.default_rng(7).standard_normal((ring, C, hop)).astype(np.float32)
    amp = np.zeros((C, n // 2 + 1), np.float32)
    cores = _effective_cores()
    used = [1]

    def timed(frames):
        buf = np.zeros((C, bufsize), np.float32)
        t0 = time.perf_counter()
        used[0] = lib.cpu_analyzer_run(f(amp), f(x), C, rank_fft, hop, frames, ring, f(window), 0.2, f(buf), bufsize, cores)
        return time.perf_counter() - t0
    rate, frames = _cpu_sized_run(timed, C, budget_s)
    assert np.isfinite(amp).all() and float(amp.max()) > 0.0
    return {"value": round(rate, 1), "unit": "channel-frames/s", "cores": used[0], "kind": "port", "cpu_model": _cpu_model(),
            "simd": "gcc -O3 -march=native vector code",
            "sample": "%d frames of %d channels (2048 new samples, one Hann-windowed 4096-point spectrum, magnitude, smoothing: "
                      "Analyzer.cpp:299-409), vectorised four-step FFT primitives, OpenMP over the channels" % (frames, C)}


def _convolver_pass(args, mi, torch, dist, rank, world, dev, C, steps, warmup):
    """One measurement of `C` Convolver channels (65536-tap IR each, rank 13): returns (result dict or None, irs)."""
    import numpy as np
    taps, frame = 65536, 4096
    rng = np.random.default_rng(4 + rank)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 16384.0)).astype(np.float32)
    bank = mi.ConvolverBank(irs, 13)
    info = bank.info()
    P = info["partitions"]
    ring = 16                                               # (a batch of frames needs outputs that lie apart)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(5 + rank)
    xin = torch.randn((ring, C, frame), generator=gen, dtype=torch.float32).to(dev)
    yout = torch.empty_like(xin)
    stream = torch.cuda.current_stream()

    def step(i):
        k = i % ring
        bank.process(yout[k], xin[k], frame, stream=stream)

    # the K frames of a region as ONE mi_convolver_bank_process_blocks call: batches of 16 frames whose tails come out of one pass
    # over the partitions' images (conv_batch_tail_kernel<16>: the probed launch), bit for bit the frame-by-frame samples
    import ctypes
    BATCH = 16
    seq = [(warmup + i) % ring for i in range(steps)]
    po = (ctypes.c_void_p * steps)(*[yout[k].data_ptr() for k in seq])
    pi = (ctypes.c_void_p * steps)(*[xin[k].data_ptr() for k in seq])
    st_ptr = ctypes.c_void_p(stream.cuda_stream)

    def region():
        mi.check(mi.lib.mi_convolver_bank_process_blocks(bank.handle, po, pi, steps, frame, frame, frame, st_ptr))
    batched = steps >= BATCH and ring >= BATCH
    if batched:
        # (15 regions: the first few of a fresh leg run 10 - 15 % longer than the ones behind them, timing.region_ms.in_order)
        b_elapsed, b_kernel_ms, b_info = _timed_steps(mi, torch, dist, world, dev, step, steps, warmup, region=region, regions=15,
                                                      probe_step=lambda j: region(), probe_steps=BATCH)
        b_info["launch"] = ("one mi_convolver_bank_process_blocks call per region: batches of %d frames, three launches each "
                            "(conv_batch_forward / _tail<%d> / _frames; below 256 channels a fourth, _finish)" % (BATCH, BATCH))
    elapsed, kernel_ms, tinfo = _timed_steps(mi, torch, dist, world, dev, step, steps, warmup)
    chk = yout[(warmup + steps - 1) % ring]
    assert bool(torch.isfinite(chk).all()) and float(chk.abs().max()) > 0.0
    bank_faults = bank.faults(stream=stream)
    bank.close()
    del xin, yout
    torch.cuda.empty_cache()
    if rank != 0:
        return None, irs
    # dominant kernel conv_step_kernel = the whole step in one launch (frame role + tail role): per
    # channel-frame the tail role reads (P-1) IR images and (P-1) ring images of 32 KiB each and writes one, the frame role
    # moves the rest of SURVEY.md 8d's 272 B per channel-sample at P = 16
    img = 8 * frame
    step_bytes = float(C) * frame * 16.0 * (P + 1)
    mac_bytes = step_bytes
    # a bank with more channels than the device has CUs goes in launches of one CU-count of channels each; the probe's event
    # pair then spans ALL launches of the step (first launch's start, last launch's stop)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    launches = (C + cus - 1) // cus
    kname = "conv_step_kernel<12>" if launches == 1 else "conv_step_kernel<12> x %d launches of <= %d channels (events span the step)" % (launches, cus)
    assert bank_faults == 0, "the roles of conv_step_kernel gave up waiting for each other %d times" % bank_faults
    res = {
        "value": round(C * frame * world * steps / elapsed / 1e6, 1), "unit": "Msamples/s",
        "ms_per_step": round(elapsed / steps * 1e3, 5), "steps": steps, "warmup": warmup,
        "config": {"workload": "Convolver (partitioned FFT overlap-add), %d channels per GPU, 65536-tap IR per "
                               "channel, rank 13, one 4096-sample frame per step" % C,
                   "channels_per_gpu": C, "taps": taps, "frame": frame, "partitions": P,
                   "images_read_per_step_MiB": round(float(C) * 2 * (P - 1) * img / 2 ** 20, 1)},
        "roofline": _roofline(kname, mac_bytes, kernel_ms, elapsed / steps * 1e3, tinfo["probe"],
                              _pmc_traffic("pmc_convolver_step_latest.json", "conv_step_kernel") if C == 256 else None, {"launches_per_step": launches}),
        "whole_step": {"algorithmic_bytes": step_bytes,
                       "achieved_GBps_incl_launch_gaps": round(step_bytes / (elapsed / steps) / 1e9, 1),
                       "frac": round(step_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    if batched:
        # The batch's own traffic model: per channel and batch of K frames the probed kernel reads the P partitions'
        # images, P - 2 of the ring's older ones, the K new ones and the pending tail, and writes K spectra and the new pending
        # tail (32 KiB each); for the whole batch in + out 16 K KiB each, the overlap-add tail, H and the ring once, the ring's
        # update.
        tail_bytes = float(C) * img * (P + (P - 2) + BATCH + 1 + BATCH + 1)
        batch_bytes = float(C) * (2 * 4 * frame * BATCH + 2 * 8 * frame + img * (P + (P - 1) + min(BATCH, P - 1)))
        per_call = dict(res)
        per_call["what"] = "the same frames as separate mi_convolver_bank_process calls: one launch of conv_step_kernel per frame (SURVEY 8d's streaming model, 272 B per channel-sample)"
        per_call.pop("config", None)
        res = {
            "value": round(C * frame * world * steps / b_elapsed / 1e6, 1), "unit": "Msamples/s",
            "ms_per_step": round(b_elapsed / steps * 1e3, 5), "steps": steps, "warmup": warmup,
            "config": dict(res["config"], call="one mi_convolver_bank_process_blocks call per region (batches of %d frames; "
                                               "bit-identical to %d process() calls -- those are timed under \"per_call\")" % (BATCH, steps)),
            "timing": b_info,
            "roofline": _roofline("conv_batch_tail_kernel<%d> (%d frames per launch; the batch is three launches)" % (BATCH, BATCH),
                                  tail_bytes, b_kernel_ms, b_elapsed / steps * 1e3, b_info["probe"],
                                  _pmc_traffic("pmc_convolver_latest.json", "conv_batch_tail_kernel", BATCH) if C == 256 else None,
                                  dict({"bytes_model": "this kernel per channel and batch: P + (P - 2) + K + 1 images of 32 KiB read, K + 1 written",
                                        "streaming_model_frac": round(step_bytes / (b_elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4)},
                                       **(_issue_side("conv_batch_tail_kernel<16", b_kernel_ms, BATCH) if C == 256 else {})),
                                  launch_steps=BATCH),
            "whole_step": {"algorithmic_bytes": batch_bytes / BATCH, "bytes_model": "a batch of K frames per channel: in + out 16 K KiB each, "
                           "overlap-add tail 32 KiB, H P x 32 KiB, ring (P - 1) x 32 KiB read and min(K, P - 1) x 32 KiB written",
                           "achieved_GBps_incl_launch_gaps": round(batch_bytes / BATCH / (b_elapsed / steps) / 1e9, 1),
                           "frac": round(batch_bytes / BATCH / (b_elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4)},
            "per_call": per_call,
        }
    return res, irs


def _convolver_call_stream(args, mi, torch, dist, rank, world, dev, C, call, steps, warmup):
    """The same bank fed with `call`-sample calls (sub-frame path: conv_small_kernel per aligned 256-sample block, the frame's
    commit and tail every 4096 samples).  A step is one frame's worth of calls."""
    import numpy as np
    taps, frame = 65536, 4096
    assert frame % call == 0
    rng = np.random.default_rng(4 + rank)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 16384.0)).astype(np.float32)
    bank = mi.ConvolverBank(irs, 13)
    info = bank.info()
    period = max(1, info["partitions"] - 1)             # the frame ring's lap: a captured run must cover whole laps of it
    ring = period
    gen = torch.Generator(device="cpu")
    gen.manual_seed(5 + rank)
    xin = torch.randn((ring, C, frame), generator=gen, dtype=torch.float32).to(dev)
    yout = torch.empty_like(xin)
    stream = torch.cuda.Stream(device=dev)              # a created stream: the run is captured into a hipGraph
    import ctypes
    st = ctypes.c_void_p(stream.cuda_stream)
    # raw pointers of every call, made once: the timed loop (or the capture) does nothing but the library calls
    calls = [[(ctypes.c_void_p(yout[k].data_ptr() + 4 * o), ctypes.c_void_p(xin[k].data_ptr() + 4 * o))
              for o in range(0, frame, call)] for k in range(ring)]
    fn, h = mi.lib.mi_convolver_bank_process, bank.handle

    def step(i):
        for po, pi in calls[i % ring]:
            mi.check(fn(h, po, pi, call, frame, frame, st))
    steps = max(period, steps - steps % period)
    torch.cuda.synchronize()
    elapsed, _, tinfo = _timed_steps(mi, torch, dist, world, dev, step, steps, period, profile=False, stream=stream,
                                     graph=(args.launch != "eager"))
    stream.synchronize()
    assert bool(torch.isfinite(yout).all()) and float(yout.abs().max()) > 0.0
    assert bank.faults(stream=stream) == 0
    bank.close()
    del xin, yout
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    return {"call": call, "calls_per_step": frame // call, "steps": steps, "launch": tinfo["launch"],
            "value": round(C * frame * world * steps / elapsed / 1e6, 1), "unit": "Msamples/s",
            "ms_per_step": round(elapsed / steps * 1e3, 5), "us_per_call": round(elapsed / steps / (frame // call) * 1e6, 2),
            "config": {"workload": "Convolver, %d channels per GPU, 65536-tap IR per channel, rank 13, fed in %d-sample calls "
                                   "(one kernel per aligned 256-sample block; commit + tail of the frame every 4096 samples)" % (C, call), "channels_per_gpu": C}}


def run_convolver(args, mi, torch, dist, rank, world, dev):
    """BASELINE.json configs[2]: 256 channels per GPU, 65536-tap IR per channel, rank 13 -> 4096-sample frames.
    At that size the images a step reads (248 MiB) just fit the 256 MiB Infinity Cache, so part of the figure is cache
    bandwidth; the same pass over 512 channels (496 MiB, streamed from HBM with non-temporal loads) is reported next to it
    under "beyond_infinity_cache" so that the BASELINE-size fraction is not read as an HBM figure."""
    C = args.conv_channels
    whole = lambda k: (k // 16) * 16 if k >= 16 else k      # (whole batches of 16 frames in a region)
    res, irs = _convolver_pass(args, mi, torch, dist, rank, world, dev, C, whole(args.conv_steps), args.conv_warmup)
    big = None
    if C == 256:
        big, _ = _convolver_pass(args, mi, torch, dist, rank, world, dev, 512, whole(max(48, args.conv_steps // 2)), args.conv_warmup)
    if rank != 0:
        return None
    stream_res = None
    if args.call > 0:
        stream_res = _convolver_call_stream(args, mi, torch, dist, rank, world, dev, C, args.call, max(20, args.conv_steps // 4), 3)
    if rank != 0:
        return None
    if big is not None:
        res["beyond_infinity_cache"] = {k: big[k] for k in ("value", "ms_per_step", "config", "roofline", "whole_step")}
    if stream_res is not None:
        stream_res["fraction_of_whole_frame_rate"] = round(stream_res["value"] / res.get("per_call", res)["value"], 3)   # (a launch per frame)
        res["call_stream"] = stream_res
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_convolver(irs, 4096)
    return res


def _timed_steps(mi, torch, dist, world, dev, step, steps, warmup, profile=True, regions=5, stream=None, graph=False, region=None,
                 probe_step=None, probe_sync=True, probe_steps=1, fill_seconds=0.0, sample_every=0):
    """W untimed warm-up calls of step(i), then `regions` timed repetitions of the K-step region, each bracketed by
    barrier + synchronize on both sides and reduced with MAX over the ranks.
    Returns (median region seconds, sorted kernel ms list of the probe pass, info).
    probe_step: what the probe pass calls instead of step() when a step has more than one launch -- the dominant kernel's
    launch alone, so that its event pair does not span the other kernels of the step.
    fill_seconds > 0: `regions` is a minimum -- the count is raised (to at most 5000) so that the timed regions cover about
    that much time; sized from 25 untimed regions and agreed between the ranks (MIN), so that every rank runs the same count.
    sample_every = n > 0 (a region that is ONE launch): every n-th timed region's launch carries its own event pair, so the
    kernel durations behind `roofline` come from the timed regions themselves and a rocprofv3 average of the same command is an
    average over the same launches; the probe pass afterwards is then not taken.
    probe_sync: the stream is drained in front of every probed launch (what rocprofv3's serialised kernel durations measure
    too); back to back the start stamp of a pair can be taken while the launch before still drains, or the launches
    overlap their ramps, and the pair reads anything between 0.93 and 1.4 of the kernel (11.8 .. 12.6 us for the 12.8 us
    biquad launch, 11.0 .. 16.4 us for the 11.8 us analysis launch, run to run)."""
    import ctypes
    import gc
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()

    # The K steps of a region, captured once: steady-state process() calls take no host decision.
    exe, mode = None, ("eager" if region is None else "one mi_biquad_bank_process_blocks call per region: the K blocks ride ONE launch "
                       "(biquad_stream_kernel; runs of 128 blocks when K is larger)")
    if graph and stream is not None:
        gc.collect()
        gc.disable()
        try:
            mi.check(mi.lib.mi_dspu_graph_begin_capture(ctypes.c_void_p(stream.cuda_stream)))
            try:
                for i in range(steps):
                    step(warmup + i)
            finally:
                h = ctypes.c_void_p()
                rc = mi.lib.mi_dspu_graph_end_capture(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(h))
            mi.check(rc)
            exe, mode = h, "hipGraph of %d process() calls, one graph launch per region" % steps
        except mi.MiError as e:
            print("bench: graph capture refused (%s), eager launches" % e, file=sys.stderr)
        finally:
            gc.enable()
        torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def one_region():
        if exe is not None:
            mi.check(mi.lib.mi_dspu_graph_launch(exe, ctypes.c_void_p(stream.cuda_stream)))
        elif region is not None:
            region()                                        # the K steps as one library call
        else:
            for i in range(steps):
                step(warmup + i)

    if fill_seconds > 0.0:
        fence()
        t0 = time.perf_counter()
        for _ in range(25):
            one_region()
            fence()
        per = (time.perf_counter() - t0) / 25.0
        want = int(min(max(regions, fill_seconds / max(per, 1e-6)), 5000))
        if world > 1:
            wt = torch.tensor([want], dtype=torch.int64, device=dev)
            dist.all_reduce(wt, op=dist.ReduceOp.MIN)
            want = int(wt.item())
        regions = max(regions, want) | 1                    # (odd: the median is a region that ran)

    def new_event():
        e = ctypes.c_void_p()
        mi.check(mi.lib.mi_dspu_event_create(ctypes.byref(e)))
        return e

    if sample_every < 0:                                    # auto: about 128 sampled launches spread over the run
        sample_every = max(1, regions // 128) if regions >= 16 else 0
    times, pairs = [], []
    gc.collect()
    gc.disable()            # no interpreter housekeeping inside a timed region
    gap_s = float(os.environ.get("MI_BENCH_REGION_GAP_MS", "0")) * 1e-3      # experiment knob: idle time between regions
    for r in range(regions):
        sampled = sample_every > 0 and profile and (r % sample_every) == sample_every - 1 and len(pairs) < 256
        if sampled:
            pairs.append((new_event(), new_event()))
        fence()
        if gap_s > 0.0:
            time.sleep(gap_s)
        if sampled:
            mi.check(mi.lib.mi_dspu_profile_next_launch(pairs[-1][0], pairs[-1][1]))
        t0 = time.perf_counter()
        one_region()
        # the closing bracket: this rank's work is done (synchronize) -> its clock stops -> the ranks meet (barrier).  The
        # region's time is the MAX over the ranks of start-together-to-own-completion (below); with the barrier inside every
        # rank's span each of them would add a collective's latency (tens of microseconds, the size of a K = 20 region) to all.
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        if world > 1:
            dist.barrier()
    gc.enable()
    in_region_ms = []
    for e0, e1 in pairs:
        ms = ctypes.c_float()
        if mi.lib.mi_dspu_event_elapsed_ms(ctypes.byref(ms), e0, e1) == 0:
            in_region_ms.append(float(ms.value))
        mi.lib.mi_dspu_event_destroy(e0)
        mi.lib.mi_dspu_event_destroy(e1)
    if pairs:
        mi.lib.mi_dspu_profile_next_launch(None, None)
    if exe is not None:
        mi.lib.mi_dspu_graph_destroy(exe)
    if world > 1:
        tt = torch.tensor(times, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        times = [float(v) for v in tt.tolist()]

    # Untimed probe pass: the dominant kernel of each call carries its own start/stop event pair (hipExtLaunchKernelGGL).
    # An event pair serialises its launch against its neighbours, which is what rocprofv3 measures too; the first probe
    # still overlaps the launch before it and is dropped.
    kernel_ms, probe_mode = [], None
    st = sorted(times)
    step_ms = st[len(st) // 2] / steps * 1e3
    if profile and len(in_region_ms) >= 8:
        kernel_ms, probe_mode = in_region_ms, "event pairs on every %d-th launch of the timed regions themselves" % sample_every
    elif profile:
        def probe_pass(sync_probes):
            n = KERNEL_PROBES + 1
            starts, stops = [new_event() for _ in range(n)], [new_event() for _ in range(n)]
            for j in range(n):
                if sync_probes:
                    torch.cuda.synchronize()
                mi.check(mi.lib.mi_dspu_profile_next_launch(starts[j], stops[j]))
                (probe_step or step)(warmup + j)
            torch.cuda.synchronize()
            out = []
            for j in range(1, n):
                ms = ctypes.c_float()
                # (an event pair that no launch of the step took up reads as an error: the roofline then falls back to the
                # timed region instead of the run ending here)
                if mi.lib.mi_dspu_event_elapsed_ms(ctypes.byref(ms), starts[j], stops[j]) == 0:
                    out.append(float(ms.value))
            mi.lib.mi_dspu_profile_next_launch(None, None)
            for e in starts + stops:
                mi.lib.mi_dspu_event_destroy(e)
            return out
        # A launch cannot take longer than the steps it carries (probe_steps of them: a K-block launch of the biquad
        # bank carries K).  An event pair whose start stamp is taken while the launch before it is still running reads too
        # long, and a drained stream hands an isolated launch its ramps in full: the drained pass is the default, a pass
        # that contradicts the timed region is repeated the other way, and if that does not help either the caller is told
        # (`probe` says "inconsistent": the roofline is then priced from the timed region itself).
        bound = 1.05 * step_ms * probe_steps
        drained = probe_sync or os.environ.get("MI_BENCH_PROBE_SYNC", "0") == "1"
        kernel_ms = probe_pass(drained)
        probe_mode = "stream drained before each probed launch" if drained else "back to back"
        if not kernel_ms:
            probe_mode = "inconsistent"
        elif _probe_mean(kernel_ms) > bound:
            second = probe_pass(not drained)
            if second and _probe_mean(second) <= bound:
                kernel_ms = second
                probe_mode = ("back to back" if drained else "stream drained before each probed launch") + \
                             " (the first pass read longer than the steps of the launch)"
            else:
                probe_mode = "inconsistent"
        if os.environ.get("MI_BENCH_DUMP_PROBES"):
            print("probes us: " + " ".join("%.2f" % (v * 1e3) for v in kernel_ms), file=sys.stderr)
    def _med(v):
        v = sorted(v)
        return round(v[len(v) // 2] * 1e3, 5)
    info = {"launch": mode, "regions": regions, "probe": probe_mode,
            "region_ms": {"min": round(st[0] * 1e3, 5), "median": round(st[len(st) // 2] * 1e3, 5),
                          "max": round(st[-1] * 1e3, 5), "first": round(times[0] * 1e3, 5),
                          # (the walk of the region times, thinned to <= 128 points; the detail file only -- never the line)
                          "in_order_every": max(1, len(times) // 128),
                          "in_order": [round(v * 1e3, 4) for v in times[::max(1, len(times) // 128)]]}}
    if len(times) >= 50:                                    # the onset transient of the power controller, and the settled state
        info["region_ms"]["median_of_first_25"] = _med(times[:25])
        info["region_ms"]["median_of_last_25"] = _med(times[-25:])
    return st[len(st) // 2], sorted(kernel_ms), info


def run_equalizer(args, mi, torch, dist, rank, world, dev):
    """BASELINE.json configs[3]: 32-band Equalizer (FIR mode, fir_rank 12), 256 channels per GPU, 4096-sample blocks."""
    import numpy as np
    C, nfilt, fir_rank, n = args.eq_channels, 32, 12, 4096
    rng = np.random.default_rng(6 + rank)
    eq = mi.EqualizerBank(C, nfilt, fir_rank)
    eq.set_mode(mi.EqualizerBank.FIR)
    eq.set_sample_rate(48000)
    freqs = np.exp(np.linspace(np.log(20.0), np.log(20000.0), nfilt))
    FLT_BT_RLC_BELL = 11
    for c in range(C):
        gains = 10.0 ** (rng.uniform(-12.0, 12.0, nfilt) / 20.0)
        for i in range(nfilt):
            eq.set_params(i, FLT_BT_RLC_BELL, 1, float(freqs[i]), float(freqs[i]), float(gains[i]), 2.0, channel=c)
    # a buffer of its own for every block of a region (192): nothing comes back out of the 256 MB last-level cache -- an HBM figure --
    # and the waves take a run's units in a row (MI_BENCH_EQ_RING=8: round 5's ring of eight, which sat in that cache)
    ring = int(os.environ.get("MI_BENCH_EQ_RING", "192"))
    gen = torch.Generator(device="cpu")
    gen.manual_seed(60 + rank)
    xin = (torch.randn((ring, C, n), generator=gen, dtype=torch.float32) * 0.25).to(dev)
    yout = torch.empty_like(xin)
    stream = torch.cuda.current_stream()
    eq.get_latency(stream)                                  # reconfigure (designer + FIR synthesis) outside the timing

    def step(i):
        eq.process(yout[i % ring], xin[i % ring], n, stream=stream)
    # the K steps of a region as ONE mi_equalizer_bank_process_blocks call: runs of up to 128 blocks ride one launch of
    # conv_frames_wave_kernel (a wave per block on the wave-resident 4096-point transform, overlap-save: DESIGN.md 3.3)
    import ctypes
    K = args.conv_steps
    seq = [(args.conv_warmup + i) % ring for i in range(K)]
    po = (ctypes.c_void_p * K)(*[yout[k].data_ptr() for k in seq])
    pi = (ctypes.c_void_p * K)(*[xin[k].data_ptr() for k in seq])
    st_ptr = ctypes.c_void_p(stream.cuda_stream)
    launch_steps = K if K <= 128 else 127                   # (mi::conv_frames_chunk: a full launch is 127 blocks = 128 units of the waves' work)

    def region():
        mi.check(mi.lib.mi_equalizer_bank_process_blocks(eq.handle, po, pi, K, n, n, n, st_ptr))
    elapsed, kernel_ms, tinfo = _timed_steps(mi, torch, dist, world, dev, step, K, args.conv_warmup, region=region,
                                             probe_step=lambda j: region(), probe_steps=launch_steps)
    tinfo["launch"] = "one mi_equalizer_bank_process_blocks call per region: runs of up to 128 blocks ride ONE launch (conv_frames_wave_kernel)"
    pc_elapsed, pc_kernel_ms, pc_info = _timed_steps(mi, torch, dist, world, dev, step, K, 0)
    assert bool(torch.isfinite(yout[0]).all())
    eq.close()
    if rank != 0:
        return None
    step_bytes = 24.0 * C * n                               # 24 B per channel-sample
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline_equalizer(C, n)
    res = {
        "value": round(C * n * world * args.conv_steps / elapsed / 1e6, 1), "unit": "Msamples/s",
        "ms_per_step": round(elapsed / args.conv_steps * 1e3, 5), "steps": args.conv_steps, "warmup": args.conv_warmup,
        "config": {"workload": "Equalizer EQM_FIR, 32 x FLT_BT_RLC_BELL per channel, fir_rank 12, %d channels per GPU, "
                               "4096-sample blocks" % C, "channels_per_gpu": C},
        # the step is ONE launch: conv_frame_kernel<12> pulls the frame out of the delay line, transforms, multiplies with
        # the channel's FIR image, transforms back, overlap-adds and emits
        "timing": tinfo,
        # the region is ONE launch per 127 blocks: a wave of conv_frames_wave_kernel per block -- two blocks of samples in 64
        # registers per lane, forward transform, split-product-merge against the response's (alpha, beta) table, inverse, the upper
        # half out (overlap-save)
        "roofline": _roofline("conv_frames_wave_kernel (%d blocks per launch)" % launch_steps, step_bytes * launch_steps, kernel_ms,
                              elapsed / args.conv_steps * 1e3, tinfo["probe"],
                              _pmc_traffic("pmc_equalizer_latest.json", "conv_frames_wave_kernel", launch_steps) if C == 256 else None,
                              _issue_side("conv_frames_wave_kernel", kernel_ms, launch_steps) if C == 256 else None,
                              launch_steps=launch_steps),
        "whole_step": {"algorithmic_bytes": step_bytes,
                       "achieved_GBps_incl_launch_gaps": round(step_bytes / (elapsed / args.conv_steps) / 1e9, 1),
                       "frac": round(step_bytes / (elapsed / args.conv_steps) / 1e9 / HBM_PEAK_GBS, 4)},
        "per_call": {"what": "the same blocks as separate mi_equalizer_bank_process calls (one launch of conv_frame_kernel per block)",
                     "value": round(C * n * world * args.conv_steps / pc_elapsed / 1e6, 1), "unit": "Msamples/s",
                     "ms_per_step": round(pc_elapsed / args.conv_steps * 1e3, 5),
                     "roofline": _roofline("conv_frame_kernel<12>", step_bytes, pc_kernel_ms, pc_elapsed / args.conv_steps * 1e3, pc_info["probe"])},
    }
    if cpu is not None:
        res["cpu_baseline"] = cpu
    return res


def run_spectral(args, mi, torch, dist, rank, world, dev):
    """BASELINE.json configs[4]: 4096-point Hann spectrum of 1024 channels per GPU every 2048 samples, per-bin sum
    over ALL channels of the job: local reduction on the device + one RCCL all-reduce per batch of frames."""
    sharding = importlib.import_module("lsp-dsp-units_amd.sharding")
    C, rank_fft, hop, batch = args.spec_channels, 12, 2048, int(os.environ.get("MI_BENCH_SPEC_BATCH", "16"))
    sr = 48000
    an = mi.AnalyzerBank(C, rank_fft, sr, 1.0, 0)
    for what, v in ((an.SAMPLE_RATE, sr), (an.RATE, sr / float(hop)), (an.RANK, rank_fft), (an.WINDOW, 0),
                    (an.REACTIVITY, 0.2), (an.SHIFT, 1.0)):
        an.configure(what, v)
    ring = 8
    gen = torch.Generator(device="cpu")
    gen.manual_seed(7 + rank)
    xin = torch.randn((ring, C, hop), generator=gen, dtype=torch.float32).to(dev)
    bins = (1 << (rank_fft - 1)) + 1
    # two buffers of per-bin sums: batch k's collective runs beside batch k + 1's analysis, which fills the other one
    sums2 = torch.zeros((2, batch, bins), dtype=torch.float32, device=dev)
    sums = sums2[0]
    stream = torch.cuda.current_stream()
    an.process(xin[0], hop, stream=stream)
    info = an.info()
    assert info["period"] == hop, info
    # the exchange step runs inside the library: its own RCCL communicator (ncclAllReduce from the C++ host side);
    # the rehearsal mode (several ranks on one device, gloo) keeps the torch collective
    state = {"comm": None, "collective": "none (one rank)"}
    if world > 1:
        state["collective"] = "torch.distributed all_reduce (%s)" % dist.get_backend()
        if dist.get_backend() == "nccl":
            try:
                state["comm"] = sharding.library_comm(mi)   # (a communicator on every rank or on none: the ranks agree inside)
                if state["comm"] is not None:
                    state["collective"] = ("mi_analyzer_bank_allreduce_bins_begin: ncclAllReduce from the library's host side on its side stream "
                                           "(RCCL over xGMI), beside the next batch's analysis; the sums double-buffered")
            except Exception as e:                          # the measurement goes on with torch's communicator
                print("bench: library communicator refused (%s); torch.distributed all_reduce instead" % e, file=sys.stderr)

    def step(i):
        # one frame: ingest + strobe analysis of every channel, then the local per-bin sum over this GPU's channels
        an.process_reduce(xin[i % ring], hop, sums[i % batch], stream=stream)
        if (i % batch) == batch - 1:                        # one collective per `batch` frames (RCCL over xGMI)
            if state["comm"] is not None:
                an.allreduce_bins(sums, batch, state["comm"], stream=stream)
            else:
                sharding.allreduce_bins(sums)

    def batch_step(i):
        # the same `batch` frames as ONE mi_analyzer_bank_process_reduce_frames call: the strobes as one launch, their reductions as one
        # launch, then the collective -- on the library's side stream, beside the next batch (which fills the other buffer): the
        # stream only waits for the collective that used THIS buffer two batches ago
        b = (i // batch) & 1
        if state["comm"] is not None:
            state["comm"].wait(b, stream=stream)
        an.process_reduce_frames([xin[(i + j) % ring] for j in range(batch)], hop, sums2[b], stream=stream)
        if state["comm"] is not None:
            an.allreduce_bins_begin(sums2[b], sums2[b], batch, state["comm"], b, stream=stream)
        else:
            sharding.allreduce_bins(sums2[b])
    steps = args.conv_steps - (args.conv_steps % batch) or batch
    # probes: the analysis launch alone, the stream drained in front of each (behind a step's own bin_reduce_kernel -- or
    # behind another analysis launch that is still draining -- the start stamp of the event pair is taken early and the pair
    # reads more than the kernel: 16.4 us in one run, 11.0 in the next, against rocprofv3's 11.8)
    def region():
        for i in range(0, steps, batch):
            batch_step(i)
        if state["comm"] is not None:                       # the region ends when its last collectives have
            state["comm"].wait(0, stream=stream)
            state["comm"].wait(1, stream=stream)
    # probes: the analysis launch of a batch alone (analyzer_frames_wave_kernel takes the event pair, the reduction and the
    # collective behind it do not), the stream drained in front of each
    elapsed, kernel_ms, tinfo = _timed_steps(mi, torch, dist, world, dev, step, steps, batch, region=region,
                                             probe_step=lambda i: batch_step(i * batch), probe_sync=True, probe_steps=batch)
    tinfo["launch"] = ("%d mi_analyzer_bank_process_reduce_frames calls of %d frames each per region (the %d strobes as ONE launch of "
                       "analyzer_frames_wave_kernel -- raw magnitudes -- then bin_smooth_reduce_kernel + bin_combine_kernel: the smoothing "
                       "walked over the strobes where the per-bin sums are formed)" % (steps // batch, batch, batch))
    pc_elapsed, pc_kernel_ms, pc_info = _timed_steps(mi, torch, dist, world, dev, step, steps, batch,
                                                     probe_step=lambda i: an.process(xin[i % ring], hop, stream=stream), probe_sync=True)
    assert bool(torch.isfinite(sums2).all()) and float(sums2.abs().max()) > 0.0
    if state["comm"] is not None:
        state["comm"].close()
    an.close()
    if rank != 0:
        return None
    frame_bytes = float(C) * (4096 * 4 + bins * 4)          # 24 580 B per channel-frame
    cpu = cpu_baseline_analyzer(C, rank_fft, hop) if (not args.no_cpu_baseline and world == 1) else None
    res = {
        "value": round(C * world * steps / elapsed, 1), "unit": "channel-frames/s",
        "msamples_per_s": round(C * world * hop * steps / elapsed / 1e6, 1),
        "ms_per_step": round(elapsed / steps * 1e3, 5), "steps": steps, "warmup": batch,
        "config": {"workload": "Analyzer: 4096-point Hann spectrum per channel every 2048 samples, %d channels per GPU, "
                               "per-bin sum over all channels (all-reduce of %d x %d floats per %d frames)"
                               % (C, batch, bins, batch), "channels_per_gpu": C, "collective": state["collective"]},
        "timing": tinfo,
        "per_call": {"what": "a mi_analyzer_bank_process_reduce call per frame (analysis launch + reduction launch)",
                     "ms_per_step": round(pc_elapsed / steps * 1e3, 5), "value": round(C * world * steps / pc_elapsed, 1), "unit": "channel-frames/s",
                     "roofline": _roofline("analyzer_kernel<11>", frame_bytes, pc_kernel_ms, pc_elapsed / steps * 1e3, pc_info["probe"])},
        "roofline": _roofline("analyzer_frames_wave_kernel (%d frames per launch)" % batch, frame_bytes * batch, kernel_ms,
                              elapsed / steps * 1e3, tinfo["probe"],
                              _pmc_traffic("pmc_spectral_latest.json", "analyzer_frames_wave_kernel", batch) if C == 1024 else None,
                              _issue_side("analyzer_frames_wave_kernel", kernel_ms, batch) if C == 1024 else None,
                              launch_steps=batch),
        "whole_step": {"algorithmic_bytes": frame_bytes,
                       "achieved_GBps_incl_launch_gaps": round(frame_bytes / (elapsed / steps) / 1e9, 1),
                       "frac": round(frame_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    if cpu is not None:
        res["cpu_baseline"] = cpu
    return res


def run_spectral_processor(args, mi, torch, dist, rank, world, dev):
    """Row a10: SpectralProcessor with a fused gain mask, rank 12 (frames of 4096 samples, hops of 2048), 1024 channels x
    4096 samples per step = two hops per channel.  Algorithmic bytes: 4 B in + 4 B out per channel-sample."""
    import numpy as np
    C, rank_fft, n = 1024, 12, 4096
    sp = mi.SpectralBank(C, rank_fft)
    sp.set_rank(rank_fft)
    sp.bind_mask(np.linspace(1.0, 0.25, (1 << (rank_fft - 1)) + 1).astype(np.float32))
    # a buffer of its own for every block of a run of 64 (2 x 1 GiB: nothing comes back out of the 256 MiB last-level cache,
    # and stft_wave_blocks_kernel -- whose waves may take a channel's run in segments side by side -- wants the run's buffers apart)
    ring = int(os.environ.get("MI_BENCH_STFT_RING", "64"))
    gen = torch.Generator(device="cpu"); gen.manual_seed(90 + rank)
    xin = (torch.randn((ring, C, n), generator=gen, dtype=torch.float32) * 0.25).to(dev)
    yout = torch.empty_like(xin)
    stream = torch.cuda.current_stream()

    def step(i):
        sp.process(yout[i % ring], xin[i % ring], n, stream=stream)
    K = args.conv_steps

    # the K blocks of a region as ONE library call (runs of 64 per launch); the pointer tables are made once, outside the timed
    # regions -- a C / C++ host has them at hand, and building 2 K tensor views per region in Python cost 5 us per block of the
    # figures of rounds 3 - 4 (kernel 13.5 us per block under rocprofv3, step 18.9)
    import ctypes
    seq = [(args.conv_warmup + i) % ring for i in range(K)]
    po = (ctypes.c_void_p * K)(*[yout[k].data_ptr() for k in seq])
    pi = (ctypes.c_void_p * K)(*[xin[k].data_ptr() for k in seq])
    st_ptr = ctypes.c_void_p(stream.cuda_stream)

    def region():
        mi.check(mi.lib.mi_spectral_bank_process_blocks(sp.handle, po, pi, K, n, n, n, st_ptr))
    elapsed, _, _ = _timed_steps(mi, torch, dist, world, dev, step, K, args.conv_warmup, profile=False, region=region)
    pc_elapsed, _, _ = _timed_steps(mi, torch, dist, world, dev, step, K, args.conv_warmup, profile=False)
    assert bool(torch.isfinite(yout).all()) and float(yout.abs().max()) > 0.0
    sp.close()
    if rank != 0:
        return None
    return _step_result("spectral_processor", "SpectralProcessor with a gain mask, rank 12, %d channels per GPU, 4096-sample blocks "
                        "(two hops each: the two frames of a block as ONE 4096-point complex transform pair, a wave per channel; "
                        "a buffer of its own for every block of a run)" % C,
                        C, n, K, elapsed, world, 8.0,
                        {"call": "one mi_spectral_bank_process_blocks call per region: runs of 64 blocks ride stft_wave_blocks_kernel "
                                 "(fft_wave.h: within 1e-6 of %d process() calls; MI_DSPU_COMPAT_BITS=1: stft_stream_blocks_kernel, their bits)" % K,
                         "per_call": {"what": "the same blocks as separate mi_spectral_bank_process calls (one launch of stft_stream_kernel per block)",
                                      "value": round(C * n * world * K / pc_elapsed / 1e6, 1), "ms_per_step": round(pc_elapsed / K * 1e3, 5),
                                      "whole_step_frac": round(8.0 * C * n / (pc_elapsed / K) / 1e9 / HBM_PEAK_GBS, 4)}})


def _step_result(name, workload, C, n, steps, elapsed, world, bytes_per_sample, extra=None):
    """Sub-result of a SURVEY 8f row: whole-step throughput against the algorithmic bytes (several launches per step)."""
    step_bytes = float(bytes_per_sample) * C * n
    res = {"value": round(C * n * world * steps / elapsed / 1e6, 1), "unit": "Msamples/s",
           "ms_per_step": round(elapsed / steps * 1e3, 5), "steps": steps,
           "config": {"workload": workload, "channels_per_gpu": C, "block": n},
           "whole_step": {"algorithmic_bytes": step_bytes, "bytes_per_channel_sample": bytes_per_sample,
                          "achieved_GBps_incl_launch_gaps": round(step_bytes / (elapsed / steps) / 1e9, 1),
                          "frac": round(step_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4)}}
    if extra:
        res.update(extra)
    # (the row's dominant launch against the vector unit's issue rate, priced from the whole step: the rows carry no kernel probes)
    kern = {"crossover": "biquad_stream_chain_kernel", "splitter": "splitter_wave_blocks_kernel", "spectral_processor": "stft_wave_blocks_kernel",
            "dynfilter": "dynfilter_kernel"}.get(name)
    side = _issue_side(kern, [elapsed / steps * 1e3], 1) if kern else {}
    if side and (C, n) == {"splitter": (256, 4096)}.get(name, (1024, 4096)):
        res["whole_step"]["valu_issue_frac"] = side["valu_issue_frac"]
    return res


def run_crossover(args, mi, torch, dist, rank, world, dev):
    """IIR Crossover, 4 bands LR4 (3 split points), 1024 channels x 4096 samples per step.
    Algorithmic bytes: 4 B in + 4 bands x 4 B out = 20 B per channel-sample."""
    C, bands, n = 1024, 4, 4096
    xo = mi.CrossoverBank(C, bands)
    xo.set_sample_rate(48000)
    for i, f in enumerate((200.0, 1500.0, 7000.0)):
        xo.set_slope(i, 2); xo.set_frequency(i, f)
    ring = 4
    gen = torch.Generator(device="cpu"); gen.manual_seed(70 + rank)
    xin = (torch.randn((ring, C, n), generator=gen, dtype=torch.float32) * 0.25).to(dev)
    outs = [[torch.empty((C, n), dtype=torch.float32, device=dev) for _ in range(bands)] for _ in range(ring)]
    stream = torch.cuda.current_stream()

    def step(i):
        xo.process(outs[i % ring], xin[i % ring], n, stream=stream)
    K = args.conv_steps

    # the K blocks of a region as ONE library call (runs of 64 per launch), its pointer tables made once outside the timed regions
    import ctypes
    seq = [(args.conv_warmup + i) % ring for i in range(K)]
    po = (ctypes.c_void_p * (K * bands))(*[outs[k][b].data_ptr() for k in seq for b in range(bands)])
    pi = (ctypes.c_void_p * K)(*[xin[k].data_ptr() for k in seq])
    st_ptr = ctypes.c_void_p(stream.cuda_stream)

    def region():
        mi.check(mi.lib.mi_crossover_bank_process_blocks(xo.handle, po, pi, K, n, n, n, st_ptr))
    elapsed, _, _ = _timed_steps(mi, torch, dist, world, dev, step, K, args.conv_warmup, profile=False, region=region)
    pc_elapsed, _, _ = _timed_steps(mi, torch, dist, world, dev, step, K, args.conv_warmup, profile=False)
    assert all(bool(torch.isfinite(o).all()) for o in outs[0])
    xo.close()
    if rank != 0:
        return None
    return _step_result("crossover", "Crossover (IIR), 4 bands LR4, %d channels per GPU, 4096-sample blocks" % C,
                        C, n, K, elapsed, world, 20.0,
                        {"call": "one mi_crossover_bank_process_blocks call per region: the %d blocks ride biquad_stream_chain_kernel "
                                 "(runs of 64 blocks per launch), bit-identical to %d process() calls" % (K, K),
                         "per_call": {"what": "the same blocks as separate mi_crossover_bank_process calls (one launch of biquad_chain_kernel per block)",
                                      "value": round(C * n * world * K / pc_elapsed / 1e6, 1), "ms_per_step": round(pc_elapsed / K * 1e3, 5),
                                      "whole_step_frac": round(20.0 * C * n / (pc_elapsed / K) / 1e9 / HBM_PEAK_GBS, 4)}})


def run_splitter(args, mi, torch, dist, rank, world, dev):
    """FFTCrossover / SpectralSplitter, rank 12 (frames of 2048 new samples), 4 bands of real gains,
    256 channels x 4096 samples per step.  Algorithmic bytes: 4 B in + 4 x 4 B out = 20 B per channel-sample."""
    C, bands, rank_fft, n = args.split_channels, 4, 12, 4096
    sp = mi.SplitterBank(C, rank_fft, bands)
    edges = [(None, (300.0, -32.0)), ((300.0, -32.0), (2000.0, -32.0)), ((2000.0, -32.0), (8000.0, -32.0)), ((8000.0, -32.0), None)]
    for b, (hp, lp) in enumerate(edges):
        sp.bind_mask(b, mi.crossover_fft_mask(hp, lp, 1.0, 1.0, 48000, rank_fft))
    ring = int(os.environ.get("MI_BENCH_SPLIT_RING", "64"))  # a buffer of its own for every block of a run of 64 (1.3 GB)
    gen = torch.Generator(device="cpu"); gen.manual_seed(80 + rank)
    xin = (torch.randn((ring, C, n), generator=gen, dtype=torch.float32) * 0.25).to(dev)
    outs = [[torch.empty((C, n), dtype=torch.float32, device=dev) for _ in range(bands)] for _ in range(ring)]
    stream = torch.cuda.current_stream()

    def step(i):
        sp.process(outs[i % ring], xin[i % ring], n, stream=stream)
    K = args.conv_steps

    # the K blocks of a region as ONE library call (runs of 64 per launch), its pointer tables made once outside the timed regions
    import ctypes
    seq = [(args.conv_warmup + i) % ring for i in range(K)]
    po = (ctypes.c_void_p * (K * bands))(*[outs[k][b].data_ptr() for k in seq for b in range(bands)])
    pi = (ctypes.c_void_p * K)(*[xin[k].data_ptr() for k in seq])
    st_ptr = ctypes.c_void_p(stream.cuda_stream)

    def region():
        mi.check(mi.lib.mi_splitter_bank_process_blocks(sp.handle, po, pi, K, n, n, n, st_ptr))
    elapsed, _, _ = _timed_steps(mi, torch, dist, world, dev, step, K, args.conv_warmup, profile=False, region=region)
    pc_elapsed, _, _ = _timed_steps(mi, torch, dist, world, dev, step, K, args.conv_warmup, profile=False)
    assert all(bool(torch.isfinite(o).all()) for o in outs[0])
    sp.close()
    if rank != 0:
        return None
    return _step_result("splitter", "FFTCrossover / SpectralSplitter, rank 12, 4 bands, %d channels per GPU, 4096-sample "
                        "blocks (a wave per channel and quarter of the run: the two frames of a block as ONE 4096-point complex "
                        "transform, one forward and four inverse transforms per channel and step, the spectrum and the bands' "
                        "overlap-add tails in registers; a buffer of its own for every block of a run)" % C,
                        C, n, K, elapsed, world, 20.0,
                        {"call": "one mi_splitter_bank_process_blocks call per region: runs of 64 blocks ride splitter_wave_blocks_kernel "
                                 "(fft_wave.h: within 1e-6 of %d process() calls; MI_DSPU_COMPAT_BITS=1: splitter_hops_blocks_kernel, one "
                                 "workgroup per channel and band, their bits)" % K,
                         "per_call": {"what": "the same blocks as separate mi_splitter_bank_process calls (one launch of splitter_hop_kernel per block)",
                                      "value": round(C * n * world * K / pc_elapsed / 1e6, 1), "ms_per_step": round(pc_elapsed / K * 1e3, 5),
                                      "whole_step_frac": round(20.0 * C * n / (pc_elapsed / K) / 1e9 / HBM_PEAK_GBS, 4)}})


def run_loudness(args, mi, torch, dist, rank, world, dev):
    """512 stereo LoudnessMeter (K weighting, 400 ms) + 512 stereo ILUFSMeter on the same 1024 channel
    rows, 4096 samples per step.  Algorithmic bytes per channel-sample: 4 B in + 2 B of the meter's output row (one row
    per two channels) for each of the two meters = 8 B."""
    M, K, n = 512, 2, 4096
    lm = mi.LoudnessBank(M, K, 400.0)
    lm.set_sample_rate(48000)
    im = mi.ILUFSBank(M, K, 10.0, 400.0)
    im.set_sample_rate(48000)
    ring = 4
    gen = torch.Generator(device="cpu"); gen.manual_seed(90 + rank)
    xin = (torch.randn((ring, M * K, n), generator=gen, dtype=torch.float32) * 0.25).to(dev)
    o1 = torch.empty((M, n), dtype=torch.float32, device=dev)
    o2 = torch.empty((M, n), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    def step(i):
        lm.process(o1, None, xin[i % ring], n, stream=stream)
        im.process(o2, xin[i % ring], n, stream=stream)
    elapsed, _, _ = _timed_steps(mi, torch, dist, world, dev, step, args.conv_steps, args.conv_warmup, profile=False)
    assert bool(torch.isfinite(o1).all()) and bool(torch.isfinite(o2).all())
    # The two banks are independent objects: a host that runs both may give each a stream of its own.  Reported beside the
    # one-stream figure, never instead of it (events on the forking stream around 200 steps, best of five).
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    best = None
    for _ in range(0 if args.no_stream_pair else 5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        s1.wait_event(e0); s2.wait_event(e0)
        for i in range(200):
            lm.process(o1, None, xin[i % ring], n, stream=s1)
            im.process(o2, xin[i % ring], n, stream=s2)
        stream.wait_stream(s1); stream.wait_stream(s2)
        e1.record(stream)
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 200.0
        best = t if best is None else min(best, t)
    lm.close(); im.close()
    if rank != 0:
        return None
    return _step_result("loudness", "LoudnessMeter + ILUFSMeter (K weighting), %d stereo meters each per GPU, 4096-sample "
                        "blocks" % M, M * K, n, args.conv_steps, elapsed, world, 8.0,
                        extra=None if best is None else {"banks_on_two_streams": {"ms_per_step": round(best, 5), "note": "each "
                               "bank on a stream of its own; not the figure `value` is computed from"}})


def run_dynfilter(args, mi, torch, dist, rank, world, dev):
    """DynamicFilters, one FLT_BT_RLC_BELL filter (slope 2: two sections) per channel whose gain follows a
    per-sample curve (what a dynamic equalizer feeds it), 1024 channels x 4096 samples per step.  Algorithmic bytes:
    4 B in + 4 B gain + 4 B out = 12 B per channel-sample."""
    C, n = 1024, 4096
    FLT_BT_RLC_BELL = 11
    df = mi.DynFilterBank(C, 1)
    df.set_sample_rate(48000)
    df.set_params(0, FLT_BT_RLC_BELL, 2, 1000.0, 1000.0, 1.0, 2.0)
    df.set_filter_active(0, True)                  # (filters start inactive, DynamicFilters.cpp:108-120: process() copies)
    ring = 4
    gen = torch.Generator(device="cpu"); gen.manual_seed(95 + rank)
    xin = (torch.randn((ring, C, n), generator=gen, dtype=torch.float32) * 0.25).to(dev)
    t = torch.arange(n, dtype=torch.float32) / n
    curve = (1.0 + 0.8 * torch.sin(2.0 * 3.14159265 * (3.0 * t[None, :] + torch.rand((C, 1), generator=gen)))).contiguous().to(dev)
    out = torch.empty((C, n), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    def step(i):
        df.process(0, out, xin[i % ring], curve, n, stream=stream)
    elapsed, _, _ = _timed_steps(mi, torch, dist, world, dev, step, args.conv_steps, args.conv_warmup, profile=False)
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0.0
    assert float((out - xin[(args.conv_steps - 1) % ring]).abs().max()) > 1e-3, "the filter did not act on the signal"
    df.close()
    if rank != 0:
        return None
    return _step_result("dynfilter", "DynamicFilters, FLT_BT_RLC_BELL slope 2 with a per-sample gain curve, %d channels per GPU, "
                        "4096-sample blocks" % C, C, n, args.conv_steps, elapsed, world, 12.0)



LINE_LIMIT = 4000          # bytes of the final stdout line (the driver keeps the last ~8 KB of output: VERDICT r04)


def _r(v, nd=4):
    return round(float(v), nd) if isinstance(v, (int, float)) and not isinstance(v, bool) else v


def _short_roofline(rf):
    """The roofline keys the driver's contract names, plus the second (issue) roof and the measured-over-algorithmic traffic."""
    if not rf:
        return None
    out = {"kernel": str(rf.get("kernel", ""))[:64], "bound": rf.get("bound"), "peak": rf.get("peak"), "unit": rf.get("unit"),
           "achieved": rf.get("achieved"), "frac": rf.get("frac"), "traffic": _r(rf.get("traffic"), 0),
           "source": rf.get("source")}
    alg = rf.get("algorithmic_bytes_per_launch")
    if alg:
        out["algorithmic"] = _r(alg, 0)
        if rf.get("traffic"):
            out["traffic_ratio"] = _r(rf["traffic"] / alg, 3)
    for k in ("kernel_avg_us", "kernel_samples", "steps_per_launch", "whole_step_frac", "valu_issue_frac", "shader_clock_ghz",
              "valu_issue_frac_at_clock"):
        if rf.get(k) is not None:
            out[k] = rf[k]
    if rf.get("traffic") is None and rf.get("stale"):
        out["stale"] = True
    return out


def _short_cpu(cb, full=False):
    if not cb:
        return None
    out = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind")}
    if full:
        if cb.get("one_core"):
            out["one_core"] = cb["one_core"].get("value")
        out["cpu_model"] = cb.get("cpu_model")
        out["sample"] = str(cb.get("sample", ""))[:220]
    else:
        out["simd"] = cb.get("simd")
    return {k: v for k, v in out.items() if v is not None}


def _short_sub(sub):
    """A sub-workload in the line: value, time per step, kernel fraction (+ second roof), whole-step fraction, CPU figure."""
    if not sub:
        return None
    out = {"value": sub.get("value"), "unit": sub.get("unit"), "ms_per_step": sub.get("ms_per_step")}
    rf = sub.get("roofline")
    if rf:
        out["roofline"] = {k: v for k, v in {
            "kernel": str(rf.get("kernel", "")).split(" (")[0][:40], "frac": rf.get("frac"),
            "traffic_ratio": _r(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 3)
                             if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch") else None,
            "valu_issue_frac": rf.get("valu_issue_frac")}.items() if v is not None}
    ws = sub.get("whole_step") or {}
    if ws.get("frac") is not None:
        out["whole_step_frac"] = ws["frac"]
    pc = sub.get("per_call") or {}
    if pc.get("ms_per_step") is not None:
        out["per_call_ms"] = pc["ms_per_step"]
    if sub.get("cpu_baseline"):
        out["cpu_baseline"] = _short_cpu(sub["cpu_baseline"])
    return out


def compact_line(full, detail_path=None):
    """The ONE stdout line, <= LINE_LIMIT bytes: the contract's keys, `roofline` and `cpu_baseline` of the headline, and per
    sub-workload only value / time per step / fractions / CPU figure.  Everything else lives in the detail file."""
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: (str(v)[:200] if isinstance(v, str) else v) for k, v in cfg.items()
                      if k in ("workload", "channels_per_gpu", "block", "sections", "blocks_per_call", "parallelism", "taps", "frame")}
    tm = full.get("timing") or {}
    if tm:
        rm = tm.get("region_ms") or {}
        line["timing"] = {k: v for k, v in {"regions": tm.get("regions"), "region_ms_median": rm.get("median"),
                                            "region_ms_first25": rm.get("median_of_first_25"),
                                            "region_ms_last25": rm.get("median_of_last_25"),
                                            "value_is": str(tm.get("value_is", ""))[:100] or None}.items() if v is not None}
    if full.get("roofline"):
        line["roofline"] = _short_roofline(full["roofline"])
    if full.get("cpu_baseline"):
        line["cpu_baseline"] = _short_cpu(full["cpu_baseline"], full=True)
    pc = full.get("per_call")
    if pc:
        line["per_call"] = {"value": pc.get("value"), "ms_per_step": pc.get("ms_per_step"),
                            "frac": (pc.get("roofline") or {}).get("frac", pc.get("whole_step_frac"))}
    em = full.get("exact_mode")
    if em:
        line["exact_mode"] = {"value": em.get("value"), "ms_per_step": em.get("ms_per_step"), "frac": em.get("frac")}
    if full.get("headline_definition_changed_in"):
        line["headline_definition_changed_in"] = full["headline_definition_changed_in"]
    if full.get("stale_counters"):
        line["stale_counters"] = len(full["stale_counters"])        # (which ones: the detail file)
        if line.get("roofline") and line["roofline"].get("traffic") is None:
            line["roofline"]["stale"] = True
    for name in ("convolver", "equalizer", "spectral"):
        if full.get(name):
            line[name] = _short_sub(full[name])
    sp = (full.get("spectral") or {}).get("spectral_processor")
    nxt = dict(full.get("next_rows") or {})
    if sp:
        nxt["spectral_processor"] = sp
    if nxt:
        line["next_rows"] = {k: {kk: vv for kk, vv in {"value": v.get("value"), "ms_per_step": v.get("ms_per_step"),
                                                       "whole_step_frac": (v.get("whole_step") or {}).get("frac"),
                                                       "valu_issue_frac": (v.get("whole_step") or {}).get("valu_issue_frac")}.items()
                                 if vv is not None} for k, v in nxt.items() if v}
    if detail_path:
        line["detail"] = detail_path
    # never over the limit: shed the optional parts in order of (un)importance
    for drop in (("next_rows",), ("timing",), ("per_call",), ("exact_mode",), ("equalizer", "cpu_baseline"), ("spectral", "cpu_baseline"),
                 ("convolver", "cpu_baseline"), ("spectral",), ("equalizer",), ("convolver",), ("cpu_baseline", "sample")):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k) or {}
        d.pop(drop[-1], None)
    return line


def emit(full):
    if _STALE:
        full["stale_counters"] = sorted(_STALE)
    _emit(full)


def _emit(full):
    """Writes the full result to gpurun_out/bench_detail.json (nothing of it goes to stdout or stderr: the driver's tail is
    one buffer for both) and prints the compact line as the last thing on stdout."""
    detail = None
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        name = os.environ.get("MI_BENCH_DETAIL", "bench_detail.json")
        with open(os.path.join(out, name), "w") as f:
            json.dump(full, f, indent=1)
        detail = "gpurun_out/" + name
    except OSError:
        pass
    text = json.dumps(compact_line(full, detail))
    assert len(text) <= LINE_LIMIT, len(text)
    sys.stderr.flush()
    print(text, flush=True)


def _spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start N ranks through torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1) BEFORE this process has touched the GPU, and leave with their status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                                   # nothing of this process has initialised the GPU yet
            sys.exit(_spawn_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: the two must agree (n_gpus in the "
                         "JSON line is the number of ranks that ran)" % (args.gpus, os.environ["WORLD_SIZE"]))
    # (the driver's tail of the run is one buffer for stdout and stderr: the c10d socket warnings of a one-node rendezvous --
    # a dozen lines per rank -- stay out of it)
    os.environ.setdefault("TORCH_CPP_LOG_LEVEL", "ERROR")
    import numpy as np
    import torch                                   # torch first: one HIP runtime per process
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Rehearsal knob for boxes with fewer GPUs than ranks (MI_BENCH_REHEARSAL=1): the ranks share the devices there are
    # and talk over gloo -- RCCL refuses two ranks on one device.  It exercises the N > 1 control flow (sharding,
    # barriers, max over ranks, the bin all-reduce); its numbers mean nothing and the JSON line says so.
    rehearsal = world > 1 and os.environ.get("MI_BENCH_REHEARSAL", "0") == "1"
    if world > 1 and os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        # one node: the collectives' bootstrap goes over loopback (the data over xGMI); without this RCCL probes every
        # interface of the box first, which takes minutes where there is no network
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU fallback)")
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)                       # (before the process group: its collectives run on the current device)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="gloo" if rehearsal else "nccl")

    mi = importlib.import_module("lsp-dsp-units_amd")
    mi.check(mi.lib.mi_dspu_set_device(local_rank))
    import workloads as wl

    if args.workload in ("convolver", "equalizer", "spectral", "stft", "crossover", "splitter", "loudness", "dynfilter"):
        runner = {"convolver": run_convolver, "equalizer": run_equalizer, "spectral": run_spectral, "stft": run_spectral_processor,
                  "crossover": run_crossover, "splitter": run_splitter, "loudness": run_loudness,
                  "dynfilter": run_dynfilter}[args.workload]
        res = runner(args, mi, torch, dist, rank, world, dev)
        if rank == 0:
            line = {"metric": "Msamples/sec per GPU (biquad-x8 1024ch; Convolver 65536-tap) + HBM roofline %",
                    "n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "f32", "data": "synthetic"}
            line.update(res)
            if rehearsal:
                line["data"] = "synthetic; REHEARSAL: ranks share one device over gloo, not a measurement"
            emit(line)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    C, n, ring = args.channels, args.samples, args.ring
    # per-rank channel shard: rank r owns global channels [r*C, (r+1)*C).  BASELINE config C2: FLT_BT_LRX_LOPASS slope 4
    # (8 biquads), Q 0.75, cutoff log-uniform 200 Hz .. 18 kHz per channel (seed 3), designed by the PRODUCT's designer.
    fc_all = np.exp(np.random.default_rng(3).uniform(np.log(200.0), np.log(18000.0), size=C * world))
    coef = np.zeros((C, 8, 5), np.float32)
    for c in range(C):
        f = float(fc_all[rank * C + c])
        _, _, sec = mi.design_filter(FLT_BT_LRX_LOPASS, 4, f, f, 1.0, 0.75, 48000)
        assert sec.shape == (8, 5), sec.shape
        coef[c] = sec
    coef = np.ascontiguousarray(coef[:, :args.sections])
    bank = mi.BiquadBank(C, max(1, coef.shape[1]))
    bank.set_all_chains(coef)

    gen = torch.Generator(device="cpu")
    gen.manual_seed(2 + rank)
    xin = (torch.randn((ring, C, n), generator=gen, dtype=torch.float32) * 0.25).to(dev)
    yout = torch.empty_like(xin)
    stream = torch.cuda.Stream(device=dev)                  # a created stream: the region is captured into a hipGraph
    bank.commit(stream)
    torch.cuda.synchronize()

    def step(i):
        k = i % ring
        bank.process(yout[k], xin[k], n, stream=stream)

    # Short regions (K = 20: 0.16 - 0.2 ms each) are timed 101 times: the chip's power controller answers the onset of the
    # load with a dip of the shader clock that lasts 20 - 30 regions (160 -> 215 us per region) and settles after 60 - 80
    # (165 - 175 us; timing.region_ms.in_order shows the walk, profiles/r04_experiments/bench_k20_regions.txt) -- 25 regions
    # measured that transient, not the state a stream of blocks runs in.  `value` is the median region.
    regions = args.regions or (5 if args.steps >= 500 else 101)
    region, launch_steps = None, 1
    if args.launch == "blocks":
        seq = [(args.warmup + i) % ring for i in range(args.steps)]
        import ctypes
        po = (ctypes.c_void_p * args.steps)(*[yout[k].data_ptr() for k in seq])
        pi = (ctypes.c_void_p * args.steps)(*[xin[k].data_ptr() for k in seq])
        st_ptr = ctypes.c_void_p(stream.cuda_stream)
        launch_steps = min(args.steps, 128)                 # blocks the call's FIRST launch carries (the probed one)

        def region():
            mi.check(mi.lib.mi_biquad_bank_process_blocks(bank.handle, po, pi, args.steps, n, n, n, st_ptr))
    elapsed, kernel_ms, tinfo = _timed_steps(mi, torch, dist, world, dev, step, args.steps, args.warmup, regions=regions,
                                             stream=stream, graph=(args.launch == "graph"), region=region,
                                             probe_step=(lambda j: region()) if region is not None else None,
                                             probe_steps=launch_steps,
                                             fill_seconds=(0.0 if args.regions else 1.0),
                                             sample_every=(-1 if region is not None else 0))
    # the shader clock the headline's launches hold: one more region, then the stamps its first workgroup left
    stream_clock = None
    if region is not None and hasattr(mi.lib, "mi_dspu_last_stream_clock"):
        try:
            for _ in range(3):
                region()
            stream.synchronize()
            stream_clock = mi.last_stream_clock()[0]
        except Exception:
            stream_clock = None
    # the same steps as separate process() calls (one launch per block, a hipGraph of the K calls): reported beside `value`
    per_call = None
    if args.launch == "blocks":
        pc_elapsed, pc_kernel_ms, pc_info = _timed_steps(mi, torch, dist, world, dev, step, args.steps, 0,
                                                         regions=max(3, min(regions, 101) // 5), stream=stream, graph=True)
        per_call = (pc_elapsed, pc_kernel_ms, pc_info)

    # the bank's exact mode (mi_biquad_bank_set_exact: the reference's serial recurrence, bit for bit) on the same blocks, one
    # launch per block -- reported beside `value`, never part of it
    exact_mode = None
    if args.launch == "blocks" and hasattr(mi.lib, "mi_biquad_bank_set_exact"):
        bank.set_exact(True)
        reps = 16
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            step(i)
        stream.synchronize()
        exact_mode = (time.perf_counter() - t0) / reps
        bank.set_exact(False)

    # sanity: the output of the last step is finite and non-trivial
    chk = yout[(args.warmup + args.steps - 1) % ring]
    assert bool(torch.isfinite(chk).all()) and float(chk.abs().max()) > 0.0
    assert args.sections == 8 or args.no_cpu_baseline, "--sections is an experiment knob; the headline config has 8"

    if rank == 0:
        samples_per_step = C * n * world
        alg_bytes = 8.0 * C * n                     # 4 B in + 4 B out per channel-sample
        streamed = args.launch == "blocks" and args.steps >= 2
        kname = "biquad_stream_kernel<4> (%d blocks per launch)" % launch_steps if streamed else "biquad_bank_kernel<16,2>"
        committed = {"note": "read from files committed under profiles/ (collected by tests/prof_round.sh in an earlier "
                             "run of the same command), not measured by this run",
                     "traffic": _pmc_traffic("pmc_biquad_latest.json", "biquad_stream_kernel" if streamed else "biquad_bank_kernel", launch_steps),
                     "rocprofv3_avg_us": _profile_avg_us("biquad", "biquad_stream_kernel" if streamed else "biquad_bank_kernel<16, 2")
                                         if (C, n) == (1024, 4096) else None,
                     "parity": _committed_json("c2_parity_latest.json")}
        line = {
            "metric": "Msamples/sec per GPU (biquad-x8 1024ch; Convolver 65536-tap) + HBM roofline %",
            "value": round(samples_per_step * args.steps / elapsed / 1e6, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "biquad-x8 cascade (FilterBank::process), %d channels x %d-sample blocks per GPU, "
                            "FLT_BT_LRX_LOPASS slope 4 per-channel cutoffs, state carried across blocks" % (C, n),
                "channels_per_gpu": C, "block": n, "sections": int(coef.shape[1]),
                "resident_ring_blocks": ring, "parallelism": "channel-shard x%d, no collective" % world,
                "blocks_per_call": args.steps if args.launch == "blocks" else 1,
                "call": ("mi_biquad_bank_process_blocks: the %d blocks of a timed region in one call (one launch per run of <= 128 blocks), "
                         "bit-identical to %d mi_biquad_bank_process calls -- those are timed beside it under \"per_call\"" % (args.steps, args.steps))
                        if args.launch == "blocks" else "mi_biquad_bank_process: one call and one launch per block",
            },
            "per_gpu_msamples_s": round(C * n * args.steps / elapsed / 1e6, 1),
            "timing": tinfo,
            "roofline": _roofline(kname, alg_bytes * launch_steps, kernel_ms, elapsed / args.steps * 1e3, tinfo["probe"],
                                  committed["traffic"],
                                  _biquad_issue_side(kernel_ms, C, n, coef.shape[1], launch_steps,
                                                     "r06_biquad_stream_pmc_sq.json" if streamed else "r06_biquad_pmc_sq.json", clock_ghz=stream_clock),
                                  launch_steps=launch_steps),
            "committed_profile": committed,
        }
        line["timing"]["value_is"] = ("the K blocks of a region as ONE mi_biquad_bank_process_blocks call (one launch)" if streamed
                                      else "one launch per block (%s)" % tinfo["launch"])
        if per_call is not None:
            pe, pk, pinfo = per_call
            line["per_call"] = {
                "what": "the same blocks as separate mi_biquad_bank_process calls: one launch per block, the K calls of a region as one hipGraph",
                "value": round(samples_per_step * args.steps / pe / 1e6, 1), "unit": "Msamples/s",
                "ms_per_step": round(pe / args.steps * 1e3, 5), "timing": pinfo,
                "roofline": _roofline("biquad_bank_kernel<16,2>", alg_bytes, pk, pe / args.steps * 1e3, pinfo["probe"], None,
                                      _biquad_issue_side(pk, C, n, coef.shape[1], 1, "r06_biquad_pmc_sq.json")),
            }
        if exact_mode is not None:
            line["exact_mode"] = {
                "what": "mi_biquad_bank_set_exact(1): FilterBank::process's serial recurrence, a section per lane -- the reference's bits "
                        "(tests/test_biquad_gpu.py::test_c2_full_size_all_channels_exact_mode: 0 of 65 536 channel-blocks differ); one launch per block",
                "value": round(samples_per_step / exact_mode / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(exact_mode * 1e3, 5),
                "frac": round(alg_bytes / exact_mode / 1e9 / HBM_PEAK_GBS, 4)}
        # `value` counts the K blocks of a region as ONE call since round 4 (288 K -> 528 K Msamples/s between rounds 3 and 5 is
        # that change of what is measured; the launch per block -- per_call -- has been 285 K since round 2)
        line["headline_definition_changed_in"] = "r04"
        if stream_clock:
            line["roofline"]["shader_clock_ghz"] = round(stream_clock, 3)
        if not args.no_cpu_baseline and world == 1:         # the CPU figure is a 1-process measurement (rank 0 at N = 1 only)
            line["cpu_baseline"] = cpu_baseline_biquad(coef, n)

    bank.close()
    del xin, yout
    torch.cuda.empty_cache()
    if args.workload == "all" and args.sections == 8:
        conv = run_convolver(args, mi, torch, dist, rank, world, dev)
        eqr = run_equalizer(args, mi, torch, dist, rank, world, dev)
        spr = run_spectral(args, mi, torch, dist, rank, world, dev)
        spp = run_spectral_processor(args, mi, torch, dist, rank, world, dev)
        if spr is not None:
            spr["spectral_processor"] = spp
        nxt = {name: fn(args, mi, torch, dist, rank, world, dev)
               for name, fn in (("dynfilter", run_dynfilter), ("crossover", run_crossover), ("splitter", run_splitter),
                                ("loudness", run_loudness))}
        if rank == 0:
            line["convolver"] = conv
            line["equalizer"] = eqr
            line["spectral"] = spr
            line["next_rows"] = nxt
    if rank == 0:
        if rehearsal:
            line["data"] = "synthetic; REHEARSAL: ranks share one device over gloo, not a measurement"

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(line)                                          # the last thing this job writes


if __name__ == "__main__":
    main()
