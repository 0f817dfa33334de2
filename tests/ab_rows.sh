#!/bin/bash
# A/B of builds of the library on chosen bench rows inside one gpurun call: tests/ab_rows.sh "convolver equalizer" tag ...
cd $GRAFT_REPO_ROOT
ROWS=$1; shift
cp lsp-dsp-units_amd/libmi_dspu.so /tmp/lib_keep.so
for r in 1 2; do for v in "$@"; do
  cp tests/ab/lib_$v.so lsp-dsp-units_amd/libmi_dspu.so
  for w in $ROWS; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pc=d.get('per_call') or {}; print(\"$v $w\", 'step_us', round(d[\"ms_per_step\"]*1e3,3), 'kernel_us/step', (d.get(\"roofline\") or {}).get(\"kernel_us_per_step\"), 'frac', (d.get(\"roofline\") or {}).get(\"frac\"), 'per_call', pc.get('ms_per_step'))"
  done
done; done
cp /tmp/lib_keep.so lsp-dsp-units_amd/libmi_dspu.so
