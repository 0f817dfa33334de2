#!/bin/bash
# SQ counters of the headline launch (biquad_stream_kernel, 50 blocks per launch) in passes of <= 8 counters, through gpurun:
#   tests/prof_sq.sh r04      -> gpurun_out/profiles_r04/r04_biquad_stream_pmc_sq.json
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq_$TAG
rm -rf $O; mkdir -p $O $R/gpurun_out/profiles_$TAG
cd /tmp && export TMPDIR=/tmp
P=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    P=$((P+1))
    rocprofv3 --pmc $CTRS --kernel-trace -d $O/pass$P --output-format csv -- python3 $R/bench.py --workload biquad --no-cpu-baseline --steps 50 > $O/pass$P.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections, os, sys
sys.path.insert(0, "$R/tests")
import prof_sources
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        for key in ("biquad_stream_kernel", "biquad_bank_kernel"):
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, name, per in (("biquad_stream_kernel", "${TAG}_biquad_stream_pmc_sq.json", 50), ("biquad_bank_kernel", "${TAG}_biquad_pmc_sq.json", 1)):
    d = {"kernel": key, "config": "C2: 1024 channels x 4096 samples, 8 sections", "blocks_per_launch": per, "sources": prof_sources.sources_for(key),
         "note": "rocprofv3 --pmc in three passes (tests/prof_sq.sh), averages per dispatch of bench.py --workload biquad --steps 50; "
                 "cycle counters are in units of four clocks on gfx950"}
    for c, v in acc[key].items():
        d[c] = sum(v) / len(v)
        d["dispatches"] = len(v)
    if "SQ_WAVES" in d and d["SQ_WAVES"]:
        w = d["SQ_WAVES"]
        der = {"valu_per_wave": round(d.get("SQ_INSTS_VALU", 0) / w, 1), "valu_per_block": round(d.get("SQ_INSTS_VALU", 0) / per)}
        if "SQ_WAVE_CYCLES" in d:
            wc = d["SQ_WAVE_CYCLES"]
            for k, o in (("SQ_WAIT_ANY", "wait_any_over_wave_cycles"), ("SQ_WAIT_INST_ANY", "wait_inst_any_over_wave_cycles"),
                         ("SQ_ACTIVE_INST_ANY", "active_inst_any_over_wave_cycles")):
                if k in d:
                    der[o] = round(d[k] / wc, 3)
        d["derived"] = der
    json.dump(d, open("$R/gpurun_out/profiles_$TAG/" + name, "w"), indent=1)
    print(key, json.dumps(d.get("derived")))
PY
