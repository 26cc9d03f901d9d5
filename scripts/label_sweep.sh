#!/bin/bash
# k_label_nn under rocprofv3 for a list of MVS_LABEL_SURF values (EXPERIMENTS build) -> gpurun_out/r04/label_sweep.log
R=$PWD
mkdir -p $R/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do export MVS_LABEL_SHELLS=$c
  rm -rf /tmp/prof_label
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_label -o lab -- python3 $R/scripts/label_sweep.py > /tmp/label_run.log 2>&1 || { tail -5 /tmp/label_run.log; exit 1; }
  f=$(find /tmp/prof_label -name '*kernel_stats.csv' | head -1)
  echo "== MVS_LABEL_SHELLS=$c"
  t=$(find /tmp/prof_label -name '*kernel_trace.csv' | head -1)
  python3 - "$t" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_label" in n:
        print("   ", n.split("(")[1].split("::")[-1] if n.startswith("(") else n.split("(")[0], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1), "us")
PY
done
