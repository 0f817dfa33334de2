#!/bin/bash
# Instruction counts of every bench workload's kernels (rocprofv3 --pmc, one pass per workload, --kernel-trace only), through gpurun:
#   tests/prof_valu.sh r05   -> gpurun_out/profiles_r05/r05_kernels_pmc_sq.json
# bench.py prices the second (issue) roof of the FFT-shaped launches from the committed copy under profiles/.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/valu_$TAG
rm -rf $O; mkdir -p $O $R/gpurun_out/profiles_$TAG
cd /tmp && export TMPDIR=/tmp
for W in convolver equalizer spectral stft splitter crossover dynfilter loudness; do
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $O/$W --output-format csv -- \
        python3 $R/bench.py --workload $W --no-cpu-baseline --no-stream-pair --conv-steps 128 > $O/$W.log 2>&1
done
python3 - "$O" "$R/gpurun_out/profiles_$TAG/${TAG}_kernels_pmc_sq.json" <<'PY'
import csv, glob, json, os, sys, collections
src, dst = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import prof_sources
# units (blocks / frames) a launch of the kernel carries in these passes (--conv-steps 128)
UNITS = {"conv_frames_kernel": 128, "conv_frames_wave_kernel": 128, "conv_batch_tail_kernel": 16, "conv_batch_forward_kernel": 16, "conv_batch_frames_kernel": 16,
         "analyzer_frames_wave_kernel": 16, "bin_smooth_reduce_kernel": 16, "bin_reduce_frames_kernel": 16, "stft_stream_blocks_kernel": 64, "stft_wave_blocks_kernel": 64, "splitter_hops_blocks_kernel": 64, "splitter_wave_blocks_kernel": 64,
         "biquad_stream_chain_kernel": 64}
out = {"note": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR, one pass per bench workload "
               "(tests/prof_valu.sh: bench.py --workload W --conv-steps 128), averages per dispatch of the smallest grid a kernel ran with "
               "(the convolver row's 512-channel pass doubles it)", "kernels": {}}
for wl in sorted(os.listdir(src)):
    if not os.path.isdir(os.path.join(src, wl)):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for fn in glob.glob(os.path.join(src, wl, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[name][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, grids in acc.items():
        grid = min(grids)
        d = {c: sum(v) / len(v) for c, v in grids[grid].items()}
        if d.get("SQ_INSTS_VALU", 0) * len(grids[grid].get("SQ_INSTS_VALU", [])) < 1e7:      # (the set-up kernels of a run)
            continue
        d["dispatches"] = len(grids[grid]["SQ_INSTS_VALU"])
        d["grid_threads"] = grid
        short = name.split("<")[0]
        d["units_per_launch"] = UNITS.get(short, 1)
        d["sources"] = prof_sources.sources_for(name)
        d["valu_per_unit"] = d["SQ_INSTS_VALU"] / d["units_per_launch"]
        d["issue_floor_us_per_unit_at_2.4GHz"] = round(d["valu_per_unit"] * 4.0 / 1024.0 / 2400.0, 3)
        out["kernels"]["%s: %s" % (wl, name)] = d
json.dump(out, open(dst, "w"), indent=1)
for k, d in out["kernels"].items():
    print("%-70s VALU/launch %12.0f  units %3d  issue floor %.3f us/unit" % (k[:70], d["SQ_INSTS_VALU"], d["units_per_launch"], d["issue_floor_us_per_unit_at_2.4GHz"]))
PY
