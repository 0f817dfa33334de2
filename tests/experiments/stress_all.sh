#!/bin/bash
# Every differential stress test over a fresh range of seeds (experiment; run on the GPU box through gpurun):
#   tests/experiments/stress_all.sh <first seed> <last seed>
A=${1:-1000}; B=${2:-1200}
for T in "test_biquad_gpu test_random_operation_sequences" "test_convolver_gpu test_random_geometry_and_call_sizes" \
         "test_crossover_gpu test_random_retune_scripts" "test_delay_gpu test_delay_random_operation_sequences_bit_exact" \
         "test_delay_gpu test_ring_random_operation_sequences_bit_exact" "test_delay_gpu test_delay_lines_with_positions_of_their_own" \
         "test_ilufs_gpu test_random_operation_sequences" "test_loudness_gpu test_random_operation_sequences" \
         "test_spectral_gpu test_spectral_random_operation_sequences" "test_spectral_gpu test_analyzer_random_settings" \
         "test_splitter_gpu test_random_operation_sequences_match_oracle" "test_dynfilter_gpu test_random_operation_sequences"; do
    timeout 1500 python tests/experiments/stress_sweep.py $T $A $B 2>&1 | tail -4 | cut -c1-400
done
for R in 7 9; do                                          # (the equalizer's scripts take the FIR rank as well)
    timeout 1500 python tests/experiments/stress_sweep.py test_equalizer_gpu test_random_operation_sequences_match_oracle $A $B $R 2>&1 | tail -4 | cut -c1-400
done
