#!/bin/bash
# A/B of an environment knob inside one gpurun call: tests/ab_env.sh VAR  (runs the biquad headline with and without VAR=1)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "" "$1=1"; do
  for k in 20 1000; do
  env $v python bench.py --workload biquad --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"[$v] K=$k\", round(d[\"ms_per_step\"]*1e3,3), d[\"roofline\"].get(\"kernel_us_per_step\"))"
  done
done; done
