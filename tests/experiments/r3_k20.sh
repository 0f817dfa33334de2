#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
python3 -m pytest tests/test_biquad_gpu.py -x -q -m gpu -k "process_blocks or full_size" 2>&1 | grep -E "passed|failed"
for L in blocks graph eager; do
  python3 bench.py --workload biquad --steps 20 --warmup 5 --launch $L --no-cpu-baseline > $O/bench_k20_$L.json 2> $O/err_$L.txt
  python3 -c "
import json
d=json.loads(open('$O/bench_k20_$L.json').read()); print('$L', 'K=20', d['ms_per_step'], d['value'], d['roofline'].get('kernel_avg_us'), d['timing']['region_ms'])"
done
python3 bench.py --workload biquad --steps 1000 --warmup 5 --launch blocks --no-cpu-baseline > $O/bench_k1000_blocks.json 2> $O/err_k1000.txt
python3 -c "
import json
d=json.loads(open('$O/bench_k1000_blocks.json').read()); print('blocks K=1000', d['ms_per_step'], d['value'])"
