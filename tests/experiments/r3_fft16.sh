#!/bin/bash
# round 3: radix-16 FFT core switched in -- parity of everything FFT-shaped, then the bench rows
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_spectral_gpu.py tests/test_convolver_gpu.py tests/test_equalizer_gpu.py tests/test_splitter_gpu.py tests/test_golden_vectors.py -x -q -m gpu > $O/pytest_fft.txt 2>&1
grep -E "passed|failed" $O/pytest_fft.txt | tail -2
for W in spectral stft equalizer splitter convolver; do
  python3 bench.py --workload $W --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/bench_$W.json").read())
    print("$W", d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("kernel_avg_us"), (d.get("whole_step") or {}).get("frac"))
except Exception as e:
    print("$W failed", e); print(open("$O/bench_$W.err").read()[-600:])
PY
done
