#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_convolver_gpu.py -x -q -m gpu -k "subframe or stream_of or random_geometry or utest" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
grep -E "passed|failed|Error|assert" $O/pytest.txt | tail -8
python3 bench.py --workload convolver --call 256 --no-cpu-baseline > $O/bench_conv_call256.json 2> $O/bench_conv_call256.err
python3 -c "
import json
d=json.loads(open('$O/bench_conv_call256.json').read()); print('whole', d['ms_per_step'], d['value']); print('stream', d.get('call_stream'))"
