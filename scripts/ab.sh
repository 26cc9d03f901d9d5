#!/bin/bash
# A/B of prebuilt libraries on ONE box (box-to-box variation is ~2 %, more than most kernel changes):
#   ab/<name>.so built here beforehand; usage on the box: bash scripts/ab.sh A B [C ...]  (rounds: AB_ROUNDS, default 3)
set -e
LIB=multiviewstitch_amd/libmvs_hip.so
cp $LIB /tmp/libmvs_keep.so
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for v in "$@"; do
    cp ab/$v.so $LIB
    python3 bench.py --no-cpu-baseline --no-alt-solver ${AB_ARGS} 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['last_launch_of_a_solve']['avg_launch_us'])"
  done
done
cp /tmp/libmvs_keep.so $LIB
