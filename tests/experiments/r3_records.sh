#!/bin/bash
# round 3, final tree: rocprofv3 kernel stats + HBM counters (prof_round.sh), SQ counters of the FFT-shaped kernels and the
# DynamicFilters kernel, then the GPU suite and the bench records (r3_final.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tests/prof_round.sh r03 > gpurun_out/prof_round_r03.log 2>&1
O=$R/gpurun_out/r3sq
mkdir -p $O
for W in spectral stft equalizer splitter dynfilter; do
  bash tests/prof_pmc.sh $W "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" > $O/sq1_$W.txt 2>&1
  bash tests/prof_pmc.sh $W "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" > $O/sq2_$W.txt 2>&1
done
python3 tests/prof_sq_summarize.py $O $R/gpurun_out/r03_fft_pmc_sq.json "rocprofv3 --pmc, two passes of eight SQ counters each per workload (tests/experiments/r3_records.sh -> tests/prof_pmc.sh), averages per dispatch over the bench run; final tree of round 3 (padded affine FFT layouts, per-type DynamicFilters kernels); cycle counters in units of four clocks on gfx950"
bash tests/experiments/r3_final.sh
