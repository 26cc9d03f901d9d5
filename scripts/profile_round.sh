#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command      -> <tag>/kernel_stats.csv (ENGINE kernels only; the full table,
#      topped by torch's scene-generation kernels, goes to kernel_stats_all.csv) + the bench line printed under the profiler
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes  -> <tag>/pmc_{fetch,write}.txt (per-kernel means) and
#      <tag>/pmc_traffic_ras.json (HBM bytes per active sweep launch, x2 fetch correction calibrated on k_srt_apply)
#   3. the plain bench line                                      -> <tag>/bench.json
# Raw traces stay in /tmp (too large to pull); only the summaries land under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$TAG /tmp/prof_cg_$TAG /tmp/pmcf_$TAG /tmp/pmcw_$TAG /tmp/prof_rows_$TAG      # (a box may be reused: stale traces would make the copies below ambiguous)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 bench.py --warmup 5 --steps 20 --no-cpu-baseline --no-alt-solver --no-single-solve --no-cold-process --no-tolerance-headroom > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
cp /tmp/prof_$TAG/*/*_kernel_stats.csv $OUT/kernel_stats_all.csv
# the same command with the alternative solver (one kernel per CG iteration) as the main run: its own table
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cg_$TAG -- python3 bench.py --warmup 5 --steps 20 --no-cpu-baseline --only-alt-solver --no-cold-process --no-tolerance-headroom > $OUT/bench_cg_under_rocprof.json 2> $OUT/bench_cg_under_rocprof.err || exit 1
cp /tmp/prof_cg_$TAG/*/*_kernel_stats.csv $OUT/kernel_stats_cg_all.csv
python3 - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
for src, dst in (("kernel_stats_all.csv", "kernel_stats.csv"), ("kernel_stats_cg_all.csv", "kernel_stats_cg.csv")):
    rows = list(csv.DictReader(open(f"{out}/{src}")))
    eng = [r for r in rows if "at::native" not in r["Name"] and ("anonymous namespace" in r["Name"] or r["Name"].startswith("k_"))]
    tot = sum(float(r["TotalDurationNs"]) for r in eng)
    with open(f"{out}/{dst}", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentageOfEngineTime", "MinNs", "MaxNs", "StdDev"])
        for r in sorted(eng, key=lambda r: -float(r["TotalDurationNs"])):
            w.writerow([r["Name"].replace("(anonymous namespace)::", "").split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        f"{100 * float(r['TotalDurationNs']) / tot:.2f}", r["MinNs"], r["MaxNs"], r["StdDev"]])
PY
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf_$TAG -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver --no-cold-process --no-tolerance-headroom > /dev/null 2> $OUT/pmc_fetch.err || exit 1
python3 scripts/pmc_summary.py /tmp/pmcf_$TAG > $OUT/pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw_$TAG -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver --no-cold-process --no-tolerance-headroom > /dev/null 2> $OUT/pmc_write.err || exit 1
python3 scripts/pmc_summary.py /tmp/pmcw_$TAG > $OUT/pmc_write.txt
python3 - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
def grab(path, kernel, counter, active):
    want = None
    for ln in open(path):
        if active and ln.strip().startswith(kernel) and "ACTIVE launches only" in ln and counter in ln:
            m = re.search(r"n=(\d+) mean ([0-9.]+)", ln); want = (int(m.group(1)), float(m.group(2)))
        if not active and ln.startswith(kernel + " "):
            m = re.search(r"n=(\d+) .*'%s': ([0-9.]+)" % counter, ln); want = (int(m.group(1)), float(m.group(2)))
    return want
f = grab(f"{out}/pmc_fetch.txt", "k_ras_sweep<6, 0>", "FETCH_SIZE", True) or grab(f"{out}/pmc_fetch.txt", "k_ras_sweep<8, 0>", "FETCH_SIZE", True)
w = grab(f"{out}/pmc_write.txt", "k_ras_sweep<6, 0>", "WRITE_SIZE", True) or grab(f"{out}/pmc_write.txt", "k_ras_sweep<8, 0>", "WRITE_SIZE", True)
cf = grab(f"{out}/pmc_fetch.txt", "k_srt_apply", "FETCH_SIZE", False)
cw = grab(f"{out}/pmc_write.txt", "k_srt_apply", "WRITE_SIZE", False)
b = json.load(open(f"{out}/bench_under_rocprof.json"))
corr = round(cw[1] / cf[1]) if cf and cw else 2
json.dump({"kernel": "k_ras_sweep", "config": 3, "vertices": b["config"]["vertices"],
           "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver",
           "fetch_size_kb_reported": f[1], "write_size_kb": w[1], "fetch_correction": float(corr),
           "calibration": f"k_srt_apply in the same runs reads and writes the same number of bytes: WRITE_SIZE {cw[1]} KB, FETCH_SIZE {cf[1]} KB -> x{corr} fetch correction (MI355X_MICROARCH.md)",
           "launches": f"mean over the {f[0]} ACTIVE k_ras_sweep dispatches of the run (launches that find their solve finished return before any operand load since round 2)",
           "bytes_per_active_launch": int(1024 * (corr * f[1] + w[1])), "algorithmic_bytes_per_launch": b["roofline"]["bytes_per_launch"]},
          open(f"{out}/pmc_traffic_ras.json", "w"), indent=1)
PY
python3 bench.py --warmup 5 --steps 20 > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo bench done
# 4. the secondary rows at full size + the per-kernel table of the hot path -> <tag>/rows.md
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rows_$TAG -- python3 scripts/bench_rows.py > $OUT/rows.json 2> $OUT/rows.err || exit 1
cp /tmp/prof_rows_$TAG/*/*_kernel_stats.csv $OUT/rows_kernel_stats.csv
python3 scripts/rows_hotpath.py $OUT > $OUT/rows.md || exit 1
echo rows done
