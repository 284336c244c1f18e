#!/usr/bin/env python
""" print the top kernels of a rocprofv3 kernel_stats.csv found below a directory """
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    print(f'{n:64s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} '
          f'avg_us {float(r["AverageNs"])/1e3:10.1f} pct {float(r["Percentage"]):5.1f}')
print("total ms", tot / 1e6)
