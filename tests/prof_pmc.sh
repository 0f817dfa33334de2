#!/bin/bash
# SQ counters of one bench workload:  tests/prof_pmc.sh <workload> "<counters>" [kernel-substring]   (through gpurun)
W=$1; CTRS=$2; K=${3:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_$W
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace -d $O --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline --steps 50 --conv-steps 50 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "$K" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("   %-28s n=%d avg=%.1f" % (c, len(v), sum(v) / len(v)))
PY
