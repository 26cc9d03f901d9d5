# GPU box: the whole -m gpu suite, then (only if it passed) the driver's bench command; summaries under gpurun_out/<tag>/
set -o pipefail
TAG=${1:-r04}; N=${2:-x}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/gpu_$N.log 2>&1; rc=$?
tail -5 gpurun_out/$TAG/gpu_$N.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --warmup 5 --steps 20 > gpurun_out/$TAG/bench_$N.json 2> gpurun_out/$TAG/bench_$N.err || { tail -5 gpurun_out/$TAG/bench_$N.err; exit 1; }
python - gpurun_out/$TAG/bench_$N.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(d["value"], d["ms_per_step"], d["parity"], d["reference_schedule"]["gpu_ms"], d["reference_schedule"]["first_rep_ms"], d["solver"])
print({k: r[k] for k in ("frac", "frac_active", "avg_launch_us", "active_launches", "avg_active_launch_us", "idle_launches", "avg_idle_launch_us", "bracket_overhead_us", "separation")})
PY
