cd $GRAFT_REPO_ROOT
S="python tests/experiments/stress_sweep.py"
timeout 500 $S test_biquad_gpu test_process_blocks_random_geometries 100 160 2>&1 | tail -1
timeout 500 $S test_biquad_gpu test_random_operation_sequences 100 160 2>&1 | tail -1
timeout 500 $S test_convolver_gpu test_process_blocks_random_scripts 100 160 2>&1 | tail -1
timeout 500 $S test_crossover_gpu test_random_retune_scripts 100 160 2>&1 | tail -1
timeout 500 $S test_crossover_gpu test_process_blocks_random_geometries 100 160 2>&1 | tail -1
timeout 500 $S test_delay_gpu test_delay_random_operation_sequences_bit_exact 100 200 2>&1 | tail -1
timeout 500 $S test_delay_gpu test_ring_random_operation_sequences_bit_exact 100 200 2>&1 | tail -1
timeout 500 $S test_dynfilter_gpu test_random_operation_sequences 100 140 2>&1 | tail -1
timeout 500 $S test_equalizer_gpu test_random_operation_sequences_match_oracle 100 130 10 2>&1 | tail -1
timeout 500 $S test_ilufs_gpu test_random_operation_sequences 100 160 2>&1 | tail -1
timeout 500 $S test_loudness_gpu test_random_operation_sequences 100 160 2>&1 | tail -1
timeout 500 $S test_spectral_gpu test_spectral_random_operation_sequences 100 160 2>&1 | tail -1
timeout 500 $S test_spectral_gpu test_analyzer_random_settings 100 160 2>&1 | tail -1
timeout 500 $S test_splitter_gpu test_random_operation_sequences_match_oracle 100 140 2>&1 | tail -1
