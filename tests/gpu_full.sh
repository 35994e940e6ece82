# round-3 dev aid: the whole -m gpu suite + the bench line (run on the GPU box through gpurun)
mkdir -p gpurun_out
TAG=${1:-a}
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/r3_gpu_$TAG.log 2>&1
rc=$?
tail -22 gpurun_out/r3_gpu_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 600 python bench.py --no-c5 --no-cpu-baseline > gpurun_out/r3_bench_$TAG.json 2> gpurun_out/r3_bench_$TAG.err || exit $?
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_$TAG.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["forward_ms"], d["stage_ms_per_view"], d["full_train_step"]["ms_per_step"], d["dropin"]["train_step_ms"])
PY
