#!/bin/bash
# per-dispatch durations of k_ras_sweep of the last outer iteration of the default bench (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --no-cpu-baseline > /dev/null 2> /tmp/kt.err || { tail -5 /tmp/kt.err; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the timed region of the main pass: take the 6th k_assoc_local from the start of the patch-solver section
idx = [i for i, n in enumerate(names) if 'k_assoc_local' in n]
a, b = idx[10], idx[11]
out = []
for r in rows[a:b]:
    n = r['Kernel_Name']
    short = n.split('::')[-1].split('(')[0].split('<')[0]
    out.append(f"{short}:{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}")
print(' '.join(out))
print('span us', (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
PY
