#!/bin/bash
# usage: tests/prof_stats.sh <name> <bench args...>   -> gpurun_out/<name>/ (rocprofv3 --kernel-trace --stats, csv)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$N --output-format csv -- python3 $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/$N.log 2>&1
tail -1 $R/gpurun_out/$N.log | cut -c1-300
