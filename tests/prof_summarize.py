"""Condenses the rocprofv3 output of tests/prof_round.sh into the small files kept under profiles/.

usage: prof_summarize.py <rocprof output dir> <destination dir> <tag>
  <tag>_<workload>_kernel_stats.csv   the --stats kernel summary (top rows)
  <tag>_pmc_hbm_raw.json              average FETCH_SIZE / WRITE_SIZE (KB) per dispatch of the dominant kernels
  pmc_<workload>_latest.json          HBM bytes per launch = FETCH_SIZE x 2 (gfx950 half-report, MI355X_MICROARCH.md)
                                      + WRITE_SIZE, KB -> bytes; bench.py reads it for roofline.traffic
"""
import csv
import glob
import json
import os
import sys

KERNELS = {"biquad": "biquad_stream_kernel", "convolver": "conv_batch_tail_kernel<16,", "equalizer": "conv_frames_kernel",
           "spectral": "analyzer_frames_kernel"}
# second kernels of a workload's PMC passes (bench.py's per_call legs): summary name -> (workload, kernel)
EXTRA = {"convolver_step": ("convolver", "conv_step_kernel<12, false>")}
# the PMC passes run `bench.py --steps 50`: the headline's launch (biquad_stream_kernel) then carries 50 blocks
UNITS_PER_LAUNCH = {"biquad": 50, "equalizer": 50, "spectral": 16, "convolver": 16}


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    raw = {}
    for wl in list(KERNELS) + ["stft", "dynfilter", "crossover", "splitter", "loudness"]:
        stats = glob.glob(os.path.join(src, "stats_" + wl, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            with open(stats[0]) as f:
                rows = f.readlines()[:12]
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)), "w") as f:
                f.writelines(rows)
    jobs = [(wl, wl, k) for wl, k in KERNELS.items()] + [(name, wl, k) for name, (wl, k) in EXTRA.items()]
    for name, wl_dir, kname in jobs:
        wl = name
        per = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            files = glob.glob(os.path.join(src, "pmc_%s_%s" % (wl_dir, ctr), "**", "*counter_collection.csv"), recursive=True)
            vals = []
            for fn in files:
                with open(fn) as f:
                    for r in csv.DictReader(f):
                        if kname in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                            vals.append(float(r["Counter_Value"]))
            if vals:
                per[ctr] = {"dispatches": len(vals), "avg_KB": sum(vals) / len(vals)}
        raw[wl] = per
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            hbm = (per["FETCH_SIZE"]["avg_KB"] * 2.0 + per["WRITE_SIZE"]["avg_KB"]) * 1024.0
            with open(os.path.join(dst, "pmc_%s_latest.json" % wl), "w") as f:
                units = UNITS_PER_LAUNCH.get(wl, 1)
                json.dump({"hbm_bytes_per_launch": hbm, "kernel": kname, "units_per_launch": units, "hbm_bytes_per_unit": hbm / units,
                           "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (%s_pmc_hbm_raw.json); "
                                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced "
                                   "reads); KB -> bytes" % tag}, f, indent=1)
    with open(os.path.join(dst, "%s_pmc_hbm_raw.json" % tag), "w") as f:
        json.dump(raw, f, indent=1)
    print(json.dumps(raw))


if __name__ == "__main__":
    main()
