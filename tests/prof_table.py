"""profiles/<tag>_roofline_table.md from the committed rocprofv3 kernel summaries: kernel, calls, average duration, algorithmic bytes per
launch (SURVEY.md 8d x the units a launch carries in the profiled bench.py run) and the fraction of 8 TB/s.   python tests/prof_table.py r06"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
C = 1024 * 4096
# (file, kernel name part, units per launch, algorithmic bytes per unit, what a unit is)
ROWS = [
    ("driver_cmd", "biquad_stream_kernel<4, true>", 20, 8.0 * C, "C2 block (1024 ch x 4096, 8 B per channel-sample), 20 per launch"),
    ("biquad", "biquad_bank_kernel<16, 2, true>", 1, 8.0 * C, "C2 block, a launch per process()"),
    ("biquad", "biquad_exact_kernel<8>", 1, 8.0 * C, "C2 block, exact mode"),
    ("convolver", "conv_step_kernel<12, false>", 1, 272.0 * 256 * 4096, "C3 frame (256 ch, 272 B per channel-sample)"),
    ("convolver@524288", "conv_batch_tail_kernel<16, true>", 1, 536870912.0, "C3 batch of 16 frames at 256 ch: the tail launch (H once per batch, 15 ring images + 16 new ones, 16 tails out)"),
    ("equalizer", "conv_frames_wave_kernel", None, 24.0 * 256 * 4096, "C4 block (256 ch, 24 B per channel-sample model); launches of 127 / 65 blocks"),
    ("equalizer", "conv_frame_kernel<12>", 1, 24.0 * 256 * 4096, "C4 block, a launch per process()"),
    ("spectral", "analyzer_frames_wave_kernel", 16, 24580.0 * 1024, "C5 strobe (1024 ch x 24 580 B), 16 per launch"),
    ("spectral", "bin_smooth_reduce_kernel<true>", 16, (17 * 4 + 8) * 1024 * 2049 / 16.0, "C5 batch: 16 planes + vAmp read, vAmp + vData written (not part of the 8d model)"),
    ("spectral", "analyzer_kernel<11>", 1, 24580.0 * 1024, "C5 strobe, a launch per frame"),
    ("stft", "stft_wave_blocks_kernel", 64, 8.0 * C, "SpectralProcessor block (1024 ch x 4096, 8 B per channel-sample), 64 per launch"),
    ("splitter", "splitter_wave_blocks_kernel", 64, 20.0 * 256 * 4096, "splitter block (256 ch, 4 bands: 20 B per channel-sample), 64 per launch"),
    ("crossover", "biquad_stream_chain_kernel", 64, 20.0 * C, "crossover block (1024 ch, 4 bands: 20 B per channel-sample), 64 per launch"),
    ("dynfilter", "dynfilter_kernel", 1, 12.0 * C, "dynamic filter block (12 B per channel-sample)"),
    ("loudness", "loudness_block4_kernel", 1, 6.0 * C, "LoudnessMeter window kernel (filtered samples in, one output row per two channels)"),
    ("loudness", "biquad_sumsq_ilufs_pair_kernel", 1, 4.0 * C, "ILUFSMeter block (samples in)"),
]
out = ["# %s: rocprofv3 averages against the byte models (made by tests/prof_table.py from the `%s_*_kernel_stats.csv` files)" % (TAG, TAG), "",
       "| kernel | calls | average, us | unit | algorithmic MB per launch | fraction of 8 TB/s |", "|---|---|---|---|---|---|"]
for wl, part, units, per_unit, what in ROWS:
    grid = None
    if "@" in wl:                                           # (the convolver row profiles 256 and 512 channels: by grid size)
        wl, grid = wl.split("@")
    fn = os.path.join(ROOT, "profiles", "%s_%s_kernel_%s.csv" % (TAG, wl, "by_grid" if grid else "stats"))
    if not os.path.exists(fn):
        continue
    for r in csv.DictReader(open(fn)):
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        if not name.startswith(part) or (grid and r["GridThreads"] != grid):
            continue
        avg = float(r["AverageNs"]) / 1e3
        if units is None:
            out.append("| `%s` | %s | %.1f | %s | (per block: %.1f MB; see the bench line for the per-block time) | |" % (part, r["Calls"], avg, what, per_unit / 1e6))
        else:
            mb = per_unit * units / 1e6
            out.append("| `%s` | %s | %.2f | %s | %.1f | %.3f |" % (part, r["Calls"], avg, what, mb, mb * 1e6 / (avg * 1e-6) / 8e12))
        break
open(os.path.join(ROOT, "profiles", "%s_roofline_table.md" % TAG), "w").write("\n".join(out) + "\n")
print("\n".join(out))
