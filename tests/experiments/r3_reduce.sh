#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
python3 -m pytest tests/test_spectral_gpu.py tests/test_graph_capture_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -E "passed|failed" $O/pytest.txt | tail -2
grep -E "^E " $O/pytest.txt | head -12
python3 bench.py --workload spectral --no-cpu-baseline > $O/fused.json 2> $O/fused.err
MI_ANALYZER_NO_FUSED_REDUCE=1 python3 bench.py --workload spectral --no-cpu-baseline > $O/two.json 2> $O/two.err
python3 - <<PY
import json
for n in ("fused","two"):
    try:
        d=json.loads(open("$O/%s.json"%n).read()); print(n, d["ms_per_step"], d["whole_step"]["frac"], d["roofline"].get("kernel_avg_us"), d["roofline"].get("reason"))
    except Exception as e:
        print(n, "failed", e, open("$O/%s.err"%n).read()[-500:])
PY
