#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --workload convolver --call 256 --conv-steps 60 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:80].ljust(80), r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
PY
