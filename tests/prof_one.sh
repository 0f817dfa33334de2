#!/bin/bash
# kernel-trace stats of one bench workload:  tests/prof_one.sh <workload> [tag]   (through gpurun, from the repo root)
W=$1; TAG=${2:-one}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rm -rf $O/stats_$W; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_$W --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline > $O/stats_$W.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/stats_$W/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
PY
tail -c 400 $O/stats_$W.log
