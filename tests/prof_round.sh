#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   tests/prof_round.sh r01
# kernel-trace stats of the bench workloads, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md: never combine counters with traces other than --kernel-trace; one counter group per pass).
# Summaries land in gpurun_out/profiles_<tag>/ ; copy them into profiles/ afterwards.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in biquad convolver equalizer spectral; do
    rocprofv3 --kernel-trace --stats -d $O/stats_$W --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline > $O/stats_$W.log 2>&1
    for C in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $C --kernel-trace -d $O/pmc_${W}_$C --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline --steps 50 --conv-steps 50 > $O/pmc_${W}_$C.log 2>&1
    done
done
for W in stft dynfilter crossover splitter loudness; do       # row a10's streaming hop and the SURVEY 8f rows: kernel summary only
    rocprofv3 --kernel-trace --stats -d $O/stats_$W --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline --no-stream-pair > $O/stats_$W.log 2>&1
done
python3 $R/tests/prof_summarize.py $O $R/gpurun_out/profiles_$TAG $TAG
