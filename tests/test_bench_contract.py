"""bench.py's launcher contract, checked without a GPU: `--gpus N` must be the number of ranks that run.

* WORLD_SIZE set by a launcher and different from --gpus: refuse, loudly, before anything else happens;
* WORLD_SIZE unset and --gpus N > 1: bench.py starts the N ranks itself through torch.distributed.run (here: on a box
  without a HIP device every rank ends with "needs a HIP device", which is the proof that N rank processes were started and
  that the parent, which must never touch the GPU before spawning, left with their status)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert b"--gpus 2" in r.stderr and b"WORLD_SIZE=4" in r.stderr, r.stderr[-400:]
    assert not r.stdout.strip()                             # no JSON line from a refused run


def test_effective_cores_respects_the_cgroup_quota():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    n = bench._effective_cores()
    affinity = len(os.sched_getaffinity(0))
    assert 1 <= n <= affinity
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    except OSError:
        return
    if q != "max":
        assert n <= -(-int(q) // int(per))


def test_gpus_n_spawns_n_ranks_without_a_launcher():
    import importlib
    mi = importlib.import_module("lsp-dsp-units_amd")
    if mi.device_count() > 0:
        import pytest
        pytest.skip("GPU box: the rehearsal of this path is a gpu test (tests/test_sharding_gloo.py covers the sharded arithmetic)")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "biquad", "--no-cpu-baseline"], {})
    assert r.returncode != 0                                # the ranks' status, not the parent's own
    assert r.stderr.count(b"bench.py needs a HIP device") >= 2, r.stderr[-800:]


def _canned_full_result():
    """A full (detail) result as bench.py builds it: round 4's 20 KB line, committed under profiles/."""
    import json
    return json.load(open(os.path.join(ROOT, "profiles", "r04_driver_cmd_bench_line.json")))


def test_final_line_fits_the_drivers_tail(tmp_path, capsys, monkeypatch):
    """the driver keeps the last ~8 KB of stdout + stderr and parses the final line; round 4's line had grown to
    20 KB and BENCH_r04 was `parsed: null`.  The line must stay under 4 KB, parse from the last 4096 bytes of stdout, and carry
    the contract's keys with `roofline` and `cpu_baseline`."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    full = _canned_full_result()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))       # (the detail file goes to <ROOT>/gpurun_out)
    print("x" * 20000)                                      # whatever was on stdout before the line
    bench.emit(full)
    out = capsys.readouterr()
    assert out.err == ""                                    # nothing on stderr: the driver's tail is one buffer for both
    tail = out.out.encode()[-4096:].decode()
    last = tail.rstrip("\n").split("\n")[-1]
    assert len(last) < 4096
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == full["roofline"]["frac"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "workload" in line["config"] and "model" not in line["config"]
    for sub in ("convolver", "equalizer", "spectral"):
        assert line[sub]["value"] == full[sub]["value"] and "frac" in line[sub]["roofline"]
    detail = json.load(open(os.path.join(str(tmp_path), line["detail"])))
    assert detail["timing"]["region_ms"]["in_order"] == full["timing"]["region_ms"]["in_order"]     # nothing is lost, only moved


def test_final_line_sheds_parts_rather_than_grow():
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    full = _canned_full_result()
    full["next_rows"] = {"row%d" % i: {"value": 1.0, "ms_per_step": 0.1, "whole_step": {"frac": 0.5}} for i in range(200)}
    full["config"]["workload"] = "w" * 5000
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT
    assert "roofline" in line and "cpu_baseline" in line and "value" in line
