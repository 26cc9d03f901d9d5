#!/bin/bash
# per-dispatch durations AND the idle gap in front of every kernel for a few outer iterations of the default bench (rocprofv3 kernel trace):
# where the step's time goes that no kernel accounts for
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --no-cpu-baseline > /dev/null 2> /tmp/kt.err || { tail -5 /tmp/kt.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/kt/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'k_assoc_local' in n]
def short(n):
    return n.split('::')[-1].split('(')[0].replace('void ', '')
for step in (10, 11):
    a, b = idx[step], idx[step + 1]
    out, busy, gaps = [], 0.0, 0.0
    prev_end = None
    for r in rows[a:b + 1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        if r is not rows[b]:
            out.append(f"{short(r['Kernel_Name'])}:{(e - s) / 1e3:.1f}(+{gap:.1f})")
            busy += (e - s) / 1e3
        gaps += gap
        prev_end = e
    print(' '.join(out))
    print(f"step {step}: span {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us, kernels {busy:.1f} us, gaps {gaps:.1f} us, launches {b - a}")
# gap statistics by the kernel that FOLLOWS the gap, over the steps 5..25
agg = collections.defaultdict(list)
a, b = idx[5], idx[25]
for i in range(a + 1, b):
    agg[short(names[i])].append((int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k:40s} n={len(v):4d} gap median {v[len(v)//2]:6.2f} mean {sum(v)/len(v):6.2f} total/step {sum(v)/20:7.1f} us")
PY
