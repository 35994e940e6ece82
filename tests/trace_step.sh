# Development aid (GPU box): kernel trace of a few headline steps -> gpurun_out/trace_step/kt_kernel_trace.csv; tests/trace_step.py prints one step's timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/trace_step
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_step -o kt --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-full-step --no-c5 --no-c4 --no-dropin "$@" > gpurun_out/trace_step.log 2>&1 || exit 1
find gpurun_out/trace_step -name "*kernel_trace.csv" -exec cp {} gpurun_out/trace_step_kernel_trace.csv \;
python3 tests/trace_step.py gpurun_out/trace_step_kernel_trace.csv
