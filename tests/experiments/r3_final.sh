#!/bin/bash
# round 3 records: the GPU suite, the driver's bench command, the long bench, the convolver call stream
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/bench_k1000.json 2> $O/bench_k1000.err
python3 bench.py --workload convolver --call 256 --no-cpu-baseline > $O/bench_conv_call256.json 2> $O/bench_conv_call256.err
python3 - <<PY
import json
d=json.loads(open("$O/bench_k20.json").read())
print("k20 biquad", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("valu_issue_frac"), d["roofline"].get("whole_step_frac"))
for k in ("convolver","equalizer","spectral"):
    s=d.get(k) or {}
    print(k, s.get("ms_per_step"), s.get("value"), (s.get("roofline") or {}).get("frac"), (s.get("whole_step") or {}).get("frac"))
print("stft", d["spectral"]["spectral_processor"]["ms_per_step"], d["spectral"]["spectral_processor"]["whole_step"]["frac"])
for k,v in d["next_rows"].items(): print(k, v["ms_per_step"], v["whole_step"]["frac"])
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"), d.get("cpu_baseline",{}).get("one_core"))
d=json.loads(open("$O/bench_k1000.json").read()); print("k1000 biquad", d["ms_per_step"], d["value"], d["roofline"]["frac"])
d=json.loads(open("$O/bench_conv_call256.json").read()); print("conv stream", d["call_stream"]["ms_per_step"], d["call_stream"]["fraction_of_whole_frame_rate"])
PY
