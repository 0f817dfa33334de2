"""Prints the sorted per-launch event times of the analyzer probe pass of bench.py (MI_BENCH_PROBE_SYNC=0/1)."""
import json, subprocess, sys, os
for sync in ("0", "1"):
    env = dict(os.environ, MI_BENCH_PROBE_SYNC=sync, MI_BENCH_DUMP_PROBES="1")
    out = subprocess.run([sys.executable, "bench.py", "--workload", "spectral", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    print("sync", sync, [l for l in out.stderr.splitlines() if l.startswith("probes")][:2], out.stdout[-200:])
