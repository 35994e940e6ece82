# round-3 dev aid: a subset of GPU tests named on the command line (-k expression), then the bench line
mkdir -p gpurun_out
TAG=$1; shift
timeout -k 10 900 python -m pytest tests -m gpu -q -x "$@" > gpurun_out/r3_q_$TAG.log 2>&1
rc=$?
tail -8 gpurun_out/r3_q_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 600 python bench.py --no-cpu-baseline --no-dropin --no-full-step > gpurun_out/r3_bench_$TAG.json 2> gpurun_out/r3_bench_$TAG.err || exit $?
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_$TAG.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "fwd", d["forward_ms"], d["stage_ms_per_view"])
print("c5", d["c5"]["ms_per_step"], d["c5"]["stage_ms_per_step"])
PY
