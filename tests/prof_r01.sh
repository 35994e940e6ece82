cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r01
cd $R
python bench.py --steps 50 --warmup 10 > gpurun_out/prof_r01/bench_c3.json.log 2>gpurun_out/prof_r01/bench_c3.err && tail -1 gpurun_out/prof_r01/bench_c3.json.log &&
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01/kt -o kt --output-format csv -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r01/kt.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_r01/pmc_f -o f --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_r01/pmc_f.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_r01/pmc_w -o w --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_r01/pmc_w.log 2>&1 &&
python tests/pmc_summary.py gpurun_out/prof_r01/pmc_f gpurun_out/prof_r01/pmc_w gpurun_out/prof_r01/r01_pmc_traffic.json 1000000 1920 1080 &&
find gpurun_out/prof_r01 -name "*kernel_stats.csv" | head
