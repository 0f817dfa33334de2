# prints ms_per_step of a bench.py JSON line read from stdin (helper of the experiment scripts)
import json, sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("kernel_avg_us"))
