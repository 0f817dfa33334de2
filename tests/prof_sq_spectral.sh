#!/bin/bash
# SQ counters of the C5 analysis launch (analyzer_frames_wave_kernel, 16 strobes per launch), passes of <= 7 counters, through gpurun
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq_spectral
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
            "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT"; do
    P=$((P+1))
    rocprofv3 --pmc $CTRS --kernel-trace -d $O/pass$P --output-format csv -- python3 $R/bench.py --workload spectral --no-cpu-baseline --conv-steps 64 > $O/pass$P.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "analyzer_frames_wave_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%-32s n=%d avg=%.0f" % (c, len(v), sum(v) / len(v)))
PY
