import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "value", d["value"], "us/step", round(d["ms_per_step"]*1e3,3), "frac", r["frac"], r["source"], "us/step kernel", r.get("kernel_us_per_step"), "whole", r["whole_step_frac"], r.get("probe"))
    if "per_call" in d: print("  per_call us", round(d["per_call"]["ms_per_step"]*1e3,3), d["per_call"]["roofline"]["frac"], d["per_call"]["roofline"]["source"])
    print("  timing", d["timing"]["region_ms"], d["timing"]["launch"][:60])
    print("  cpu", d.get("cpu_baseline",{}).get("value"))
    for k in ("convolver","equalizer","spectral"):
        if k in d and d[k]: print("  ",k, d[k]["value"], round(d[k]["ms_per_step"]*1e3,3), d[k]["roofline"]["frac"], d[k]["roofline"]["source"], d[k]["roofline"].get("kernel_avg_us"), "cpu", d[k].get("cpu_baseline",{}).get("value"))
