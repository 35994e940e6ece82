# Development aid (GPU box): bench.py with and without one environment switch, interleaved.   bash tests/ab_env.sh GSR_DEV 4 [bench args]
#   -> gpurun_out/ab_env_<VAR>.txt   (lines: "<VAR>=<value|unset> ms/step fwd c5")
var=$1; val=$2; shift 2
out=gpurun_out/ab_env_$var.txt
: > $out
for rep in 1 2 3; do
  for mode in unset set; do
    if [ $mode = set ]; then export $var=$val; else unset $var; fi
    timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-c4 --no-heavy --no-dropin "$@" 2>>gpurun_out/ab_env.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
c5=d.get('c5',{})
print('$var=%s  ms/step %.4f fwd %.4f  c5 %s (kernels %s)' % ('$val' if '$mode'=='set' else 'unset', d['ms_per_step'], d['forward_ms'], c5.get('ms_per_step'), c5.get('kernel_sum_ms')))
" >> $out || { echo FAILED >> $out; exit 1; }
  done
done
unset $var
cat $out
