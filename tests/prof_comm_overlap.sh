#!/bin/bash
# What runs beside what when the C5 batch's collective goes out on the library's side stream (one GPU, a single-rank RCCL
# communicator: tests/comm_overlap_demo.py): rocprofv3 kernel + memory-copy trace, then the overlaps of every collective-side
# operation with the compute stream's kernels.   tests/prof_comm_overlap.sh [tag]   (through gpurun, from the repo root)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/comm_overlap
rm -rf $O; mkdir -p $O $R/gpurun_out/profiles_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O --output-format csv -- python3 $R/tests/comm_overlap_demo.py 8 1024 > $O/log.txt 2>&1
python3 - "$O" "$R/gpurun_out/profiles_$TAG/${TAG}_comm_overlap.txt" <<'PY'
import csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
ops = []
for fn in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60], r.get("Queue_Id", "?")))
for fn in glob.glob(os.path.join(src, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "copy"), "-"))
ops.sort()
t0 = ops[0][0] if ops else 0
main_q = next((o[4] for o in ops if "analyzer_frames_wave_kernel" in o[3]), None)
first = next((o[0] for o in ops if "analyzer_frames_wave_kernel" in o[3]), 0)
# the collective's side: what runs on another queue than the analysis while batches are in flight (a single-rank all-reduce out of
# place is RCCL's device-to-device copy -- __amd_rocclr_copyBuffer; with ranks to talk to it is an ncclDevKernel)
side = [o for o in ops if o[2] == "kernel" and o[4] != main_q and o[0] >= first and ("copyBuffer" in o[3] or "nccl" in o[3].lower())]
with open(dst, "w") as f:
    f.write("C5 batches with the collective on the library's side stream (tests/comm_overlap_demo.py 8 1024; one GPU, single-rank communicator):\n"
            "operations of the side queue and the compute-queue kernels running at the same time, us since the first operation\n")
    n_over = 0
    for s in side:
        f.write("%-34s %12.1f .. %12.1f  (%5.1f us) queue %s\n" % (s[3], (s[0] - t0) / 1e3, (s[1] - t0) / 1e3, (s[1] - s[0]) / 1e3, s[4]))
        for o in ops:
            if o[2] == "kernel" and o[4] == main_q and o[0] < s[1] and s[0] < o[1]:
                n_over += 1
                f.write("      beside %-40s %12.1f .. %12.1f  queue %s\n" % (o[3], (o[0] - t0) / 1e3, (o[1] - t0) / 1e3, o[4]))
    f.write("\n%d side-queue operations, %d overlaps with compute-queue kernels\n" % (len(side), n_over))
print(open(dst).read()[:3000])
PY
