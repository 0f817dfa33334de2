#!/bin/bash
# round 3: crossover (biquad_chain_kernel) with the bands' stores dripping under the following sections, rows per section
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for D in 0 1 2 4 8; do
touch lsp-dsp-units_amd/csrc/biquad.hip
make -s -C lsp-dsp-units_amd EXTRA=-DMI_CHAIN_DRIP=$D > /dev/null 2>&1
for i in 1 2; do python3 bench.py --workload crossover --no-cpu-baseline | python3 tests/experiments/ms.py drip_$D; done
done
