# Development aid: bench.py stage times for several values of the "dev" option (GSR_DEV) with the current library.
#   bash tests/ab_dev.sh 0 4 1 5   -> gpurun_out/ab_dev.txt
mkdir -p gpurun_out
: > gpurun_out/ab_dev.txt
for dev in "$@"; do
  echo "== GSR_DEV=$dev" >> gpurun_out/ab_dev.txt
  GSR_DEV=$dev timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-c5 2>>gpurun_out/ab_dev.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms/step %.4f fwd_ms %.4f' % (d['ms_per_step'], d['forward_ms']), {k:v for k,v in d['stage_ms_per_view'].items() if 'render' in k or 'preprocess' in k or 'sort' in k or 'refl' in k})
" >> gpurun_out/ab_dev.txt || echo "FAILED" >> gpurun_out/ab_dev.txt
done
cat gpurun_out/ab_dev.txt
