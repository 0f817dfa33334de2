#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DRIVER's bench command (python3 bench.py --gpus 1 --steps 20 --warmup 5), through gpurun:
#   tests/prof_driver_cmd.sh r04   -> gpurun_out/profiles_r04/r04_driver_cmd_kernel_stats.csv (+ the bench line of the profiled run)
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/driver_cmd_$TAG
rm -rf $O; mkdir -p $O $R/gpurun_out/profiles_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
f=$(find $O -name "*kernel_stats.csv" | head -1)
head -40 "$f" > $R/gpurun_out/profiles_$TAG/${TAG}_driver_cmd_kernel_stats.csv
tail -1 $O/bench_line.json > $R/gpurun_out/profiles_$TAG/${TAG}_driver_cmd_bench_line.json
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
