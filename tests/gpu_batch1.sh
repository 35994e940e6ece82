# round-3 dev aid: new tests, observed errors, baseline bench (run on the GPU box through gpurun)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize_chain.py tests/test_gpu_dropin.py tests/test_gpu_api_paths.py tests/test_gpu_cubemap.py -m gpu -q --durations=15 > gpurun_out/r3_tests1.log 2>&1
rc=$?
tail -25 gpurun_out/r3_tests1.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python tests/observed_errors.py > gpurun_out/r3_obs.log 2>&1 || exit $?
cat gpurun_out/r3_obs.log
timeout -k 10 600 python bench.py > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err || exit $?
tail -c 1500 gpurun_out/r3_bench0.json
