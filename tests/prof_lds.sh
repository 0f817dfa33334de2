#!/bin/bash
# LDS-side SQ counters of the transform launches (rocprofv3 --pmc, two passes per workload, --kernel-trace only), through gpurun:
#   tests/prof_lds.sh r05  -> gpurun_out/profiles_r05/r05_fft_lds_pmc_sq.json
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lds_$TAG
rm -rf $O; mkdir -p $O $R/gpurun_out/profiles_$TAG
cd /tmp && export TMPDIR=/tmp
for W in equalizer spectral stft splitter; do
    P=0
    for CTRS in "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
                "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT"; do
        P=$((P+1))
        rocprofv3 --pmc $CTRS --kernel-trace -d $O/${W}_$P --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline --conv-steps 128 > $O/${W}_$P.log 2>&1
    done
done
python3 - "$O" "$R/gpurun_out/profiles_$TAG/${TAG}_fft_lds_pmc_sq.json" <<'PY'
import csv, glob, json, os, sys, collections
src, dst = sys.argv[1], sys.argv[2]
KEEP = ("conv_frames_kernel", "conv_frames_wave_kernel", "analyzer_frames_wave_kernel", "bin_smooth_reduce_kernel", "stft_stream_blocks_kernel", "stft_wave_blocks_kernel", "splitter_hops_blocks_kernel", "splitter_wave_blocks_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        for k in KEEP:
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc, two passes per workload (tests/prof_lds.sh: bench.py --workload W --conv-steps 128), averages per dispatch; "
               "SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in units of four clocks summed over the waves, SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT in clocks "
               "summed over the CUs (256)", "kernels": {}}
for k, d in acc.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["dispatches"] = len(next(iter(d.values())))
    w = e.get("SQ_WAVES", 0)
    if w and "SQ_WAVE_CYCLES" in e:
        life = e["SQ_WAVE_CYCLES"] * 4.0 / w                 # clocks a wave lives
        cus = 256.0
        e["derived"] = {
            "wave_lifetime_clocks": round(life),
            "lds_pipe_active_frac_of_cu_time": round(e.get("SQ_LDS_IDX_ACTIVE", 0) / cus / life, 3),
            "lds_bank_conflict_frac_of_lds_active": round(e.get("SQ_LDS_BANK_CONFLICT", 0) / max(e.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3),
            "valu_active_frac_of_simd_time": round(e.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / 1024.0 / life, 3),
            "wait_any_over_wave_cycles": round(e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"], 3),
            "wait_inst_any_over_wave_cycles": round(e.get("SQ_WAIT_INST_ANY", 0) / e["SQ_WAVE_CYCLES"], 3),
        }
    out["kernels"][k] = e
json.dump(out, open(dst, "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, e.get("derived"))
PY
