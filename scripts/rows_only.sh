set -o pipefail
TAG=r03; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rows_$TAG -- python3 scripts/bench_rows.py > $OUT/rows.json 2> $OUT/rows.err || { tail -5 $OUT/rows.err; exit 1; }
cp /tmp/prof_rows_$TAG/*/*_kernel_stats.csv $OUT/rows_kernel_stats.csv
python3 scripts/rows_hotpath.py $OUT > $OUT/rows.md || exit 1
tail -9 $OUT/rows.md
for k in 1 2 3; do python3 bench.py --no-cpu-baseline --no-alt-solver 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'])"; done
