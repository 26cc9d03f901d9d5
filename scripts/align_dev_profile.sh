#!/bin/bash
# kernel and HIP-API statistics of mvs_align_dev at scan scale -> gpurun_out/r04/align_dev_{kernel,hip}_stats.csv
set -e
mkdir -p gpurun_out/r04
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_align
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d /tmp/prof_align -o align -- python3 $R/scripts/align_dev_profile.py 4 > $R/gpurun_out/r04/align_dev_profile.log 2>&1 || { tail -20 $R/gpurun_out/r04/align_dev_profile.log; exit 1; }
f=$(find /tmp/prof_align -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/r04/align_dev_kernel_stats.csv
f=$(find /tmp/prof_align -name '*hip_api_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r04/align_dev_hip_stats.csv
grep "mvs_align_dev call" $R/gpurun_out/r04/align_dev_profile.log
head -25 $R/gpurun_out/r04/align_dev_kernel_stats.csv | cut -c1-150
head -14 $R/gpurun_out/r04/align_dev_hip_stats.csv | cut -c1-150
