#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_spectral_gpu.py tests/test_golden_vectors.py tests/test_graph_capture_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -E "passed|failed" $O/pytest.txt | tail -2
for V in fused; do
  if [ $V = onehop ]; then export MI_SPECTRAL_ONE_HOP=1; else unset MI_SPECTRAL_ONE_HOP; fi
  python3 bench.py --workload stft --no-cpu-baseline > $O/bench_stft_$V.json 2> $O/bench_stft_$V.err
  python3 -c "
import json
d=json.loads(open('$O/bench_stft_$V.json').read()); print('$V', d.get('ms_per_step'), d.get('value'), d['whole_step']['frac'])"
done
