for C in 512 1024 2048 4096 8192; do
 for S in 0 8; do
  python bench.py --workload biquad --no-cpu-baseline --steps 200 --warmup 20 --channels $C --ring 4 --sections $S 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('C=$C S=$S kernel_med_us', r['kernel_median_us'], 'frac', r['frac'], 'step_us', round(d['ms_per_step']*1e3,2))"
 done
done
