#!/bin/bash
# kernel-trace averages of the C5 row for several builds of the library (tests/ab/lib_<tag>.so) inside one gpurun call
cd $GRAFT_REPO_ROOT
cp lsp-dsp-units_amd/libmi_dspu.so /tmp/lib_keep.so
for v in "$@"; do
  cp tests/ab/lib_$v.so lsp-dsp-units_amd/libmi_dspu.so
  echo "== $v"
  bash tests/prof_one.sh spectral ab_$v 2>&1 | grep "analyzer_frames\|smooth\|combine"
done
cp /tmp/lib_keep.so lsp-dsp-units_amd/libmi_dspu.so
