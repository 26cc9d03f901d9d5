"""kernel_stats.csv of `rocprofv3 --kernel-trace --stats -- python3 scripts/bench_rows.py` + the bytes it printed -> a table."""
import csv
import json
import re
import sys

stats, rows = sys.argv[1], json.load(open(sys.argv[2]))
print("| kernel | calls | avg us | algorithmic bytes / call | GB/s | note |\n|---|---|---|---|---|---|")
for r in csv.DictReader(open(stats)):
    name = r["Name"]
    m = re.search(r"\b(k_[a-z0-9_]+)\(", name)
    if not m:
        continue
    short = m.group(1)
    avg = float(r["AverageNs"]) / 1e3
    if short in rows:
        b = rows[short]["bytes"]
        print(f"| `{short}` | {r['Calls']} | {avg:.1f} | {b/1e6:.1f} MB | {b/avg/1e3:.0f} | {rows[short]['note']} |")
    elif short.startswith("k_") and float(r["Percentage"]) > 0.3:
        print(f"| `{short}` | {r['Calls']} | {avg:.1f} | | | |")
