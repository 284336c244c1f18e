import sys, json
d = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
r = d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["sclk_mhz"])
