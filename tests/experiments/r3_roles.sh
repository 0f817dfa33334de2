#!/bin/bash
# round 3: the two-role biquad kernel -- parity tests, in-kernel timeline, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_biquad_gpu.py tests/test_streams_gpu.py tests/test_crossover_gpu.py tests/test_equalizer_gpu.py -x -q -m gpu > $O/pytest_biquad.txt 2>&1
tail -5 $O/pytest_biquad.txt
./tests/experiments/biquad_roles_probe 8 4096 > $O/roles_probe.txt 2>&1
cat $O/roles_probe.txt
python3 bench.py --workload biquad --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_biquad_k20.json 2> $O/bench_biquad_k20.err
python3 bench.py --workload biquad --steps 1000 --warmup 5 --no-cpu-baseline > $O/bench_biquad_k1000.json 2> $O/bench_biquad_k1000.err
MI_BIQUAD_WAVES=2 python3 bench.py --workload biquad --steps 1000 --warmup 5 --no-cpu-baseline > $O/bench_biquad_old_k1000.json 2> $O/bench_biquad_old_k1000.err
cat $O/bench_biquad_k20.json $O/bench_biquad_k1000.json $O/bench_biquad_old_k1000.json | cut -c 1-600
