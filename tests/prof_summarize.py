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

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import prof_sources                                         # noqa: E402  (the hashes of the sources the loaded library was built from)

KERNELS = {"biquad": "biquad_stream_kernel", "convolver": "conv_batch_tail_kernel<16,", "equalizer": "conv_frames_wave_kernel",
           "spectral": "analyzer_frames_wave_kernel"}
# second kernels of a workload's PMC passes (bench.py's per_call legs): summary name -> (workload, kernel)
EXTRA = {"convolver_step": ("convolver", "conv_step_kernel<12, false>")}
# the PMC passes run `bench.py --steps 50`: the headline's launch (biquad_stream_kernel) then carries 50 blocks
UNITS_PER_LAUNCH = {"biquad": 50, "equalizer": 50, "spectral": 16, "convolver": 16}


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    raw = {}
    regs = {}
    for wl in list(KERNELS) + ["stft", "dynfilter", "crossover", "splitter", "loudness"]:
        stats = glob.glob(os.path.join(src, "stats_" + wl, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            with open(stats[0]) as f:
                rows = f.readlines()[:12]
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)), "w") as f:
                f.writelines(rows)
        # the same kernel at several grid sizes (the convolver row also runs a 512-channel pass beside BASELINE's 256): the
        # --stats summary averages them together, so the trace is summed up per (kernel, grid) as well
        traces = glob.glob(os.path.join(src, "stats_" + wl, "**", "*kernel_trace.csv"), recursive=True)
        if traces:
            per = {}
            with open(traces[0]) as f:
                for r in csv.DictReader(f):
                    key = (r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0],
                           int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
                    per.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            grids = {}
            for (name, grid) in per:
                grids.setdefault(name, set()).add(grid)
            rows = [(name, grid, v) for (name, grid), v in per.items() if len(grids[name]) > 1 and sum(v) > 1e6]
            if rows:
                with open(os.path.join(dst, "%s_%s_kernel_by_grid.csv" % (tag, wl)), "w") as f:
                    f.write('"Name","GridThreads","Calls","AverageNs","MinNs","MaxNs"\n')
                    for name, grid, v in sorted(rows):
                        f.write('"%s",%d,%d,%.1f,%d,%d\n' % (name, grid, len(v), sum(v) / len(v), min(v), max(v)))
    jobs = [(wl, wl, k) for wl, k in KERNELS.items()] + [(name, wl, k) for name, (wl, k) in EXTRA.items()]
    for name, wl_dir, kname in jobs:
        wl = name
        per = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            files = glob.glob(os.path.join(src, "pmc_%s_%s" % (wl_dir, ctr), "**", "*counter_collection.csv"), recursive=True)
            by_grid = {}
            for fn in files:
                with open(fn) as f:
                    for r in csv.DictReader(f):
                        if kname in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                            by_grid.setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
                            regs[kname] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size") if k in r}
            if by_grid:
                # (several grid sizes: BASELINE's configuration is the smallest -- the row's extra pass doubles the channels)
                grid = min(by_grid)
                vals = by_grid[grid]
                per[ctr] = {"dispatches": len(vals), "avg_KB": sum(vals) / len(vals), "grid_threads": grid,
                            "other_grids": {str(g): {"dispatches": len(v), "avg_KB": sum(v) / len(v)} for g, v in by_grid.items() if g != grid}}
        raw[wl] = per
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            hbm = (per["FETCH_SIZE"]["avg_KB"] * 2.0 + per["WRITE_SIZE"]["avg_KB"]) * 1024.0
            with open(os.path.join(dst, "pmc_%s_latest.json" % wl), "w") as f:
                units = UNITS_PER_LAUNCH.get(wl, 1)
                json.dump({"hbm_bytes_per_launch": hbm, "kernel": kname, "units_per_launch": units, "hbm_bytes_per_unit": hbm / units,
                           "sources": prof_sources.sources_for(kname), "registers": regs.get(kname),
                           "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (%s_pmc_hbm_raw.json); "
                                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced "
                                   "reads); KB -> bytes" % tag}, f, indent=1)
    with open(os.path.join(dst, "%s_pmc_hbm_raw.json" % tag), "w") as f:
        json.dump(raw, f, indent=1)
    print(json.dumps(raw))


if __name__ == "__main__":
    main()
