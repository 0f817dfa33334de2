#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
./tests/experiments/biquad_phase_probe 8 4096 > $O/phase_probe.txt 2>&1
./tests/experiments/biquad_phase_probe 0 4096 >> $O/phase_probe.txt 2>&1
for W in spectral stft equalizer splitter; do
  bash tests/prof_pmc.sh $W "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" > $O/sq1_$W.txt 2>&1
  bash tests/prof_pmc.sh $W "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" > $O/sq2_$W.txt 2>&1
done
python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
