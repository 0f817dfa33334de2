"""Oracle jobs that run in worker PROCESSES (spawned, so that they never share the parent's GPU runtime): a full-size
parity test hands every channel of a configuration to the CPU oracle and still finishes in seconds.  Test infrastructure."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def c4_channel(job):
    """One channel of BASELINE config 3 through oracle.Equalizer (EQM_FIR): returns (output with the float32 impulse response,
    output with the float64 one) -- the second tells how far the FIR's own synthesis noise moves the output."""
    import numpy as np
    import oracle
    from oracle import equalizer as oe
    from oracle import filter_design as fd
    curve, x, nfilt, rank = job

    def exact_ir(n_, coef, state):
        imp = np.zeros(n_)
        imp[0] = 1.0
        return oracle.biquad_cascade_f64(imp, coef).astype(np.float32)
    outs = []
    for ir_func in (None, exact_ir):
        o = oe.Equalizer(nfilt, rank)
        o.set_mode(oe.FIR)
        o.set_sample_rate(48000)
        for i, p in enumerate(curve):
            o.set_params(i, fd.Params(*p))
        if ir_func is not None:
            o.ir_func = ir_func
        outs.append(o.process(x))
    return outs[0], outs[1]


def run_pool(func, jobs, workers=None):
    """func over jobs in spawned worker processes, results in order."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
        workers = max(1, min(workers, 16, len(jobs)))
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(func, jobs, chunksize=max(1, len(jobs) // (4 * workers))))
