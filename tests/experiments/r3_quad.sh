#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ilsp-dsp-units_amd/csrc -Ilsp-dsp-units_amd/include -ffp-contract=on -w tests/experiments/biquad_quad_probe.hip -o /tmp/bqp -Llsp-dsp-units_amd -lmi_dspu -Wl,-rpath,$R/lsp-dsp-units_amd 2>&1 | tail -3
timeout 300 /tmp/bqp
