"""Condenses the SQ counter passes of tests/prof_pmc.sh (two passes of eight counters per workload, their printed
averages kept as gpurun_out/<dir>/sq{1,2}_<workload>.txt) into one JSON for profiles/.

usage: prof_sq_summarize.py <dir with sq1_*.txt / sq2_*.txt> <output json> "<note>"
"""
import glob
import json
import os
import re
import sys

KEEP = ("analyzer_kernel", "bin_reduce_kernel", "stft_stream_kernel", "conv_frame_kernel", "splitter_hop_kernel",
        "dynfilter_kernel", "biquad_bank_kernel", "biquad_chain_kernel", "loudness_block4_kernel", "conv_step_kernel")


def main():
    src, dst, note = sys.argv[1], sys.argv[2], sys.argv[3]
    kernels = {}
    for fn in sorted(glob.glob(os.path.join(src, "sq[12]_*.txt"))):
        wl = re.match(r"sq[12]_(\w+)\.txt", os.path.basename(fn)).group(1)
        name = None
        for line in open(fn):
            if not line.startswith(" "):
                name = line.strip()
                continue
            m = re.match(r"\s+(\w+)\s+n=(\d+) avg=([\d.]+)", line)
            if not m or name is None or not any(k in name for k in KEEP):
                continue
            short = next(k for k in KEEP if k in name)
            d = kernels.setdefault("%s: %s" % (wl, short), {"name_in_trace": name})
            d[m.group(1)] = float(m.group(3))
            d["dispatches"] = int(m.group(2))
    for d in kernels.values():
        w = d.get("SQ_WAVES")
        if not w:
            continue
        der = {}
        for key, out in (("SQ_INSTS_VALU", "valu_per_wave"), ("SQ_INSTS_LDS", "lds_per_wave"), ("SQ_INSTS_SALU", "salu_per_wave")):
            if key in d:
                der[out] = round(d[key] / w, 1)
        if "SQ_INSTS_VMEM_RD" in d and "SQ_INSTS_VMEM_WR" in d:
            der["vmem_per_wave"] = round((d["SQ_INSTS_VMEM_RD"] + d["SQ_INSTS_VMEM_WR"]) / w, 1)
        if "SQ_WAVE_CYCLES" in d:
            wc = d["SQ_WAVE_CYCLES"]
            der["wave_lifetime_us_at_2.4GHz"] = round(wc * 4.0 / w / 2400.0, 2)        # counter in units of four clocks
            for key, out in (("SQ_WAIT_ANY", "wait_any_over_wave_cycles"), ("SQ_WAIT_INST_ANY", "wait_inst_any_over_wave_cycles"),
                             ("SQ_ACTIVE_INST_ANY", "active_inst_any_over_wave_cycles")):
                if key in d:
                    der[out] = round(d[key] / wc, 3)
        d["derived"] = der
    with open(dst, "w") as f:
        json.dump({"note": note, "kernels": kernels}, f, indent=1)
    for k, d in kernels.items():
        print(k, d.get("derived"))


if __name__ == "__main__":
    main()
