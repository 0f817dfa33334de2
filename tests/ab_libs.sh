#!/bin/bash
# A/B of builds of the library on the headline inside one gpurun call: tests/ab_libs.sh tag ... (tests/ab/lib_<tag>.so, built by hand)
cd $GRAFT_REPO_ROOT
cp lsp-dsp-units_amd/libmi_dspu.so /tmp/lib_keep.so
for r in 1 2; do for v in "$@"; do
  cp tests/ab/lib_$v.so lsp-dsp-units_amd/libmi_dspu.so
  for k in 20 1000; do
  python bench.py --workload biquad --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$v K=$k\", 'step_us', round(d[\"ms_per_step\"]*1e3,3), 'kernel_us', d[\"roofline\"].get(\"kernel_avg_us\"), 'frac', d[\"roofline\"].get(\"frac\"), 'per_call', d['per_call']['ms_per_step'] if d.get('per_call') else None)"
  done
done; done
cp /tmp/lib_keep.so lsp-dsp-units_amd/libmi_dspu.so
