# Development aid (GPU box): kernel trace of a few C3 steps -> one step's timeline in gpurun_out/trace_c3_<tag>.txt
#   [GSR_LIB=/path/to/variant.so GSR_BINDING=ctypes] bash tests/trace_c3.sh tag [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tag=${1:-x}; shift
rm -rf gpurun_out/trace_c3_$tag
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/trace_c3_$tag -o kt --output-format csv -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-full-step --no-c5 --no-c4 --no-heavy --no-dropin "$@" > gpurun_out/trace_c3_$tag.log 2>&1 || exit 1
find gpurun_out/trace_c3_$tag -name "*kernel_trace.csv" -exec cp {} gpurun_out/trace_c3_${tag}_kernel_trace.csv \;
python3 tests/trace_step.py gpurun_out/trace_c3_${tag}_kernel_trace.csv 6 > gpurun_out/trace_c3_$tag.txt
rm -rf gpurun_out/trace_c3_$tag gpurun_out/trace_c3_${tag}_kernel_trace.csv
cat gpurun_out/trace_c3_$tag.txt
