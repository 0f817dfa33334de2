#!/bin/bash
# time per block of the K-block launch against the number of sections (0 = the kernel as a copy): where the launch's time goes
R=${GRAFT_REPO_ROOT:-/root/repo}
for S in 0 1 2 4 6 8; do
  for V in "" "MI_BIQUAD_NO_WIDE=1"; do
    env $V MI_BENCH_DETAIL=sweep_detail.json python3 $R/bench.py --workload biquad --no-cpu-baseline --steps ${1:-20} --sections $S 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('sections $S %-22s step %.2f us  kernel %.2f us per block (%s)' % ('$V' or 'wide', d['ms_per_step']*1e3, r.get('kernel_avg_us',0)/r.get('steps_per_launch',1), r['kernel']))
"
  done
done
