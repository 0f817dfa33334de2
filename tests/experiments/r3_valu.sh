#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r3valu
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w tests/experiments/valu_probe2.hip -o /tmp/valu_probe2 && /tmp/valu_probe2 | tee gpurun_out/r3valu/valu_issue_rates.txt
