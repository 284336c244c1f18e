#!/usr/bin/env python
"""
Turn a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into the short
per-kernel table kept under profiles/ (top kernels, calls, total ms, average us, share).

    python scripts/summarize_rocprof.py <kernel_stats.csv> <out.md> [title]
"""
import csv
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats"
    rows = list(csv.DictReader(open(src)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nsource: `{src}` -- total kernel time {total / 1e6:.1f} ms\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:24]:
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0][:80]
            f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                    f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")


if __name__ == "__main__":
    main()
