# GPU box: the round's remaining measurements -> gpurun_out/<tag>/ (copied into profiles/<tag>/ afterwards)
set -o pipefail
TAG=${1:-r04}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python scripts/box_probe.py 2>&1 | grep "box probe" | tee $OUT/box_probe.log
timeout -k 10 300 python scripts/predict_check.py 2>&1 | grep "^config" | tee $OUT/configs124.log || exit 1
timeout -k 10 300 python scripts/config5.py 2>/dev/null | tail -1 | tee $OUT/config5.json || exit 1
timeout -k 10 300 python scripts/config4.py 2>/dev/null | tail -1 | tee $OUT/config4.json || exit 1
timeout -k 10 300 python scripts/soak.py 2>&1 | grep -E "^outer|soak|worst" | tee $OUT/soak_400_outer.log || exit 1
timeout -k 10 300 python scripts/soak_config5.py 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_config5.log | tail -3 || exit 1
echo measured
