#!/bin/bash
# kernel trace of the Convolver's sub-frame call streams (run through gpurun from the repo root)
R=$(pwd); O=$R/gpurun_out/conv_calls; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "10 256" "13 128" "13 64"; do
    set -- $c
    timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/p_$1_$2 --output-format csv -- python3 $R/tests/experiments/conv_call_stream.py $1 $2 > $O/log_$1_$2.txt 2>&1
    f=$(find /tmp/p_$1_$2 -name "*kernel_stats.csv" | head -1)
    echo "== rank $1 call $2"
    if [ -n "$f" ]; then head -8 "$f" > $O/stats_$1_$2.csv; python3 -c "import csv,sys; [print(\"  %-50s calls %6s avg %9.0f ns %6s%%\" % (r[\"Name\"][:50], r[\"Calls\"], float(r[\"AverageNs\"]), r[\"Percentage\"])) for r in csv.DictReader(open(sys.argv[1]))]" $O/stats_$1_$2.csv; else echo "no stats file"; fi
done < /dev/null
