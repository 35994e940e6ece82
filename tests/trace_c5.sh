# Development aid (GPU box): kernel trace of a few C5 steps (bench.py --only-c5) -> one step's timeline in gpurun_out/trace_c5.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/trace_c5
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/trace_c5 -o kt --output-format csv -- python3 bench.py --only-c5 --steps 6 > gpurun_out/trace_c5.log 2>&1 || exit 1
find gpurun_out/trace_c5 -name "*kernel_trace.csv" -exec cp {} gpurun_out/trace_c5_kernel_trace.csv \;
python3 tests/trace_step.py gpurun_out/trace_c5_kernel_trace.csv 12 gauss_preprocess_kernel > gpurun_out/trace_c5.txt
cat gpurun_out/trace_c5.txt
