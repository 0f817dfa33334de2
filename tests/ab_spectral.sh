#!/bin/bash
# A/B of builds of the library on the C5 row inside one gpurun call: tests/ab/lib_<tag>.so (built by hand; *.so travel with the snapshot)
cd $GRAFT_REPO_ROOT
cp lsp-dsp-units_amd/libmi_dspu.so /tmp/lib_keep.so
for v in "$@"; do
  cp tests/ab/lib_$v.so lsp-dsp-units_amd/libmi_dspu.so
  for r in 1 2; do
  python bench.py --workload spectral --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$v\", 'step_us', round(d[\"ms_per_step\"]*1e3,3), 'kernel_us', d[\"roofline\"].get(\"kernel_avg_us\"), 'frac', d[\"roofline\"].get(\"frac\"))"
  done
done
cp /tmp/lib_keep.so lsp-dsp-units_amd/libmi_dspu.so
