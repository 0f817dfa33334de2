#!/bin/bash
# round 3: DynamicFilters per-type kernel -- samples per lane x register budget (waves per SIMD) x builders interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3dyn
mkdir -p $O
cd $R
row() {
  python3 bench.py --workload dynfilter --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1.json").read())
    print("$1", d.get("ms_per_step"), d.get("value"))
except Exception as e:
    print("$1 failed", e); print(open("$O/bench_$1.err").read()[-600:])
PY
}
for CFG in "8 2 2" "8 2 8" "12 2 2" "16 2 2" "16 2 4" "16 1 2" "16 1 8"; do
set -- $CFG
touch lsp-dsp-units_amd/csrc/dynfilter.hip
make -s -C lsp-dsp-units_amd EXTRA="-DMI_DYN_SAMPLES_PER_LANE=$1 -DMI_DYN_WAVES_PER_SIMD=$2 -DMI_DYN_BUILDERS_IN_FLIGHT=$3" > $O/make.txt 2>&1
row lc$1_w$2_b$3
done
