# A/B on the GPU box: k_assoc_all with 16- and 8-wave workgroups (library rebuilt with -DMVS_STAMPS each time): parity of the
# bounded passes, then the per-section timeline of a steady pass (scripts/assoc_all_timeline.py)
set -o pipefail
for W in ${WAVES:-16 8}; do
  echo "=== HEAVY_WAVES $W"
  (cd multiviewstitch_amd/csrc && make clean > /dev/null && make -j16 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DMVS_STAMPS -DMVS_HEAVY_WAVES=$W" > /tmp/build_$W.log 2>&1 || { tail -5 /tmp/build_$W.log; exit 1; }) || exit 1
  timeout -k 10 300 python -m pytest tests/test_gpu_deform.py tests/test_gpu_scale.py -m gpu -x -q -k "bounded or association_matches or far_and_nan or sample" 2>&1 | tail -3 || exit 1
  timeout -k 10 300 python scripts/assoc_all_timeline.py 3 2>&1 | grep -v amdgpu.ids | tail -7 || exit 1
  timeout -k 10 300 python bench.py --warmup 5 --steps 20 --no-cpu-baseline --no-alt-solver --no-single-solve 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step', d['ms_per_step'], 'reference schedule', d['reference_schedule']['gpu_ms'], d['reference_schedule']['phases_ms'])" || exit 1
done
