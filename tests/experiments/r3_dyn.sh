#!/bin/bash
# round 3: DynamicFilters kernels compiled per base type (sections in registers) -- parity, then the bench row at
# register budgets of 3 and 4 waves per SIMD with the any-type kernel beside them, and the division probe
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3dyn
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_dynfilter_gpu.py tests/test_golden_vectors.py -x -q -m gpu > $O/pytest_dyn.txt 2>&1
grep -E "passed|failed" $O/pytest_dyn.txt | tail -2
MI_DYNFILTER_GENERIC=1 timeout 1200 python3 -m pytest tests/test_dynfilter_gpu.py -x -q -m gpu > $O/pytest_dyn_generic.txt 2>&1
grep -E "passed|failed" $O/pytest_dyn_generic.txt | tail -2
row() {
  python3 bench.py --workload dynfilter --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1.json").read())
    print("$1", d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("kernel_avg_us"))
except Exception as e:
    print("$1 failed", e); print(open("$O/bench_$1.err").read()[-600:])
PY
}
row per_type_${MI_DYN_DEFAULT:-default}
MI_DYNFILTER_GENERIC=1 row any_type
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ilsp-dsp-units_amd/csrc -Ilsp-dsp-units_amd/include -ffp-contract=on -fno-slp-vectorize -w tests/experiments/dyn_div_probe.hip -o /tmp/dyn_div_probe -Llsp-dsp-units_amd -lmi_dspu -Wl,-rpath,$R/lsp-dsp-units_amd 2>&1 | tail -5
/tmp/dyn_div_probe | tee $O/div_probe.txt
for WV in 3 4 2; do
touch lsp-dsp-units_amd/csrc/dynfilter.hip
make -s -C lsp-dsp-units_amd EXTRA=-DMI_DYN_WAVES_PER_SIMD=$WV > $O/make$WV.txt 2>&1
row per_type_${WV}waves
done
