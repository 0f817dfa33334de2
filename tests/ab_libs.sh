#!/bin/bash
# A/B of two builds of the library inside one gpurun call: tests/ab/lib_old.so vs lib_new.so (built by hand; *.so travel with the snapshot)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in old new; do
  cp tests/ab/lib_$v.so lsp-dsp-units_amd/libmi_dspu.so
  for k in 20 1000; do
  python bench.py --workload biquad --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$v K=$k\", round(d[\"ms_per_step\"]*1e3,3), d[\"roofline\"].get(\"kernel_avg_us\"))"
  done
done; done
