#!/bin/bash
# usage: tests/prof_pmc.sh <outdir-under-gpurun_out> <counters...> -- <bench args>
# runs one rocprofv3 --pmc pass (csv) of bench.py from /tmp as the guide prescribes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
CTRS=()
while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${CTRS[@]}" --kernel-trace -d $OUT --output-format csv -- python3 $R/bench.py "$@" --no-cpu-baseline > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-200
