#!/bin/bash
# round 3: the whole GPU suite on the tree with the per-type DynamicFilters kernels, the bench row, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3dyn
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
grep -E "passed|failed" $O/pytest_all.txt | tail -2
MI_DYNFILTER_GENERIC=1 timeout 1200 python3 -m pytest tests/test_dynfilter_gpu.py -x -q -m gpu > $O/pytest_dyn_generic.txt 2>&1
grep -E "passed|failed" $O/pytest_dyn_generic.txt | tail -2
python3 bench.py --workload dynfilter --no-cpu-baseline > $O/bench_dynfilter.json 2> $O/bench_dynfilter.err; cat $O/bench_dynfilter.json | cut -c1-400
MI_DYNFILTER_GENERIC=1 python3 bench.py --workload dynfilter --no-cpu-baseline > $O/bench_dynfilter_any_type.json 2> $O/bench_dynfilter_any.err; cut -c1-200 $O/bench_dynfilter_any_type.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_dynfilter --output-format csv -- python3 $R/bench.py --workload dynfilter --no-cpu-baseline > $O/stats_dynfilter.log 2>&1
find $O/stats_dynfilter -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/dynfilter_kernel_stats.csv
head -3 $O/dynfilter_kernel_stats.csv | cut -c1-300
