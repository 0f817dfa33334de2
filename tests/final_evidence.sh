#!/bin/bash
# Everything the round's profiles/ directory is made from, in one gpurun call (from the repo root): tests/final_evidence.sh r06
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/profiles_$TAG
python3 -m pytest tests -q -m gpu 2>&1 | tail -45 > gpurun_out/profiles_$TAG/${TAG}_pytest_gpu_tail.log
cp gpurun_out/parity_report.json gpurun_out/profiles_$TAG/${TAG}_parity_report.json 2>/dev/null
cp gpurun_out/c2_parity.json gpurun_out/profiles_$TAG/${TAG}_c2_parity.json 2>/dev/null
for f in gpurun_out/c2_parity_blocks_*.json; do cp $f gpurun_out/profiles_$TAG/${TAG}_$(basename $f) 2>/dev/null; done
bash tests/prof_round.sh $TAG < /dev/null > gpurun_out/prof_round.log 2>&1
bash tests/prof_sq.sh $TAG < /dev/null > gpurun_out/prof_sq.log 2>&1
bash tests/prof_valu.sh $TAG < /dev/null > gpurun_out/prof_valu.log 2>&1
bash tests/prof_comm_overlap.sh $TAG < /dev/null > gpurun_out/prof_comm_overlap.log 2>&1
bash tests/prof_sq_spectral.sh < /dev/null > gpurun_out/profiles_$TAG/${TAG}_analyzer_wave_pmc_sq.txt 2>&1
python3 tests/experiments/conv_small_rank_rate.py 2>/dev/null | grep "^rank" > gpurun_out/profiles_$TAG/${TAG}_conv_small_rank_rate.txt
# the counters of THIS tree go where bench.py looks for them (profiles/ of the box's copy of the repo) before the lines are made:
# a line measured next to counters of an older source would say "stale" and carry no traffic
cp gpurun_out/profiles_$TAG/pmc_*_latest.json gpurun_out/profiles_$TAG/${TAG}_*pmc_sq.json profiles/ 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profiles_$TAG/${TAG}_bench_k20.json 2> gpurun_out/bench_k20.err
cp gpurun_out/bench_detail.json gpurun_out/profiles_$TAG/${TAG}_bench_k20_detail.json 2>/dev/null
python3 bench.py --workload biquad --steps 1000 --warmup 50 > gpurun_out/profiles_$TAG/${TAG}_bench_k1000.json 2> gpurun_out/bench_k1000.err
for W in crossover stft dynfilter splitter loudness; do
    python3 bench.py --workload $W --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/profiles_$TAG/${TAG}_bench_$W.json
done
MI_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/profiles_$TAG/${TAG}_bench_rehearsal_gpus2.json 2> gpurun_out/bench_gpus2.err
bash tests/prof_driver_cmd.sh $TAG < /dev/null > gpurun_out/prof_driver_cmd.log 2>&1
# what comes back is capped at 64 MiB: the raw rocprofv3 output stays on the box
du -sh gpurun_out/* 2>/dev/null | sort -h | tail -8
rm -rf gpurun_out/prof_$TAG gpurun_out/sq_$TAG gpurun_out/valu_$TAG gpurun_out/comm_overlap gpurun_out/sq_spectral gpurun_out/driver_cmd_$TAG gpurun_out/prof_driver_$TAG
for d in gpurun_out/*/; do [ "$d" != "gpurun_out/profiles_$TAG/" ] && [ $(du -sm "$d" | cut -f1) -gt 8 ] && rm -rf "$d"; done
ls gpurun_out/profiles_$TAG
grep -E "passed|failed|error" gpurun_out/profiles_$TAG/${TAG}_pytest_gpu_tail.log | tail -3
tail -3 gpurun_out/profiles_$TAG/${TAG}_pytest_gpu_tail.log
