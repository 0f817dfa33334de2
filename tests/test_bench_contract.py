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
