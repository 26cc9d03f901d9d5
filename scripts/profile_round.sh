#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command      -> profiles/<tag>/kernel_stats.csv (+ bench json)
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes  -> profiles/<tag>/pmc_{fetch,write}.txt (per-kernel means)
# Raw traces stay in /tmp (too large to pull); only the summaries land under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
cp /tmp/prof_$TAG/*/*_kernel_stats.csv $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf_$TAG -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err || exit 1
python3 scripts/pmc_summary.py /tmp/pmcf_$TAG > $OUT/pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw_$TAG -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err || exit 1
python3 scripts/pmc_summary.py /tmp/pmcw_$TAG > $OUT/pmc_write.txt
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo done
