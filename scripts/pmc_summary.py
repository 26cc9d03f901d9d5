"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch); raw CSVs are too big to pull."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "anonymous" not in k:
        continue
    k = k.split("(anonymous namespace)::")[1].split("(")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    n = len(next(iter(cs.values())))
    print(k, "n=%d" % n, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())})
    if k == "k_cg_iter" or k.startswith("k_ras_sweep"):   # launches that find the solve converged move little data: report the active ones too
        for c, v in sorted(cs.items()):
            act = [x for x in v if x > 0.5 * max(v)]
            print("   %s ACTIVE launches only:" % k, c, "n=%d" % len(act), "mean", round(sum(act) / len(act), 1), "max", max(v))
