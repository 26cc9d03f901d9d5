#!/bin/bash
# quick per-kernel durations of the default bench under rocprofv3 (engine kernels only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 bench.py --no-cpu-baseline ${KS_ARGS} > /dev/null 2> /tmp/ks.err || { tail -5 /tmp/ks.err; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ks/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if '::k_' in n and 'at::native' not in n and int(r["Calls"]) >= int(__import__("os").environ.get("KS_MIN","40")):
        nm = n.split('::k_')[1].split('(')[0]
        print(f"k_{nm:28s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:7.2f} us min {float(r['MinNs'])/1e3:7.2f}")
PY
