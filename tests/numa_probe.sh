# Development aid (GPU box): does the host's NUMA placement change the step time?  Prints the topology, then times the headline loop with the
# process pinned to each NUMA node's CPUs in turn (taskset), unpinned in between.  -> gpurun_out/numa_probe.txt
out=gpurun_out/numa_probe.txt
: > $out
{
echo "== nodes"; for n in /sys/devices/system/node/node*; do echo "$n $(cat $n/cpulist)"; done
echo "== gpu numa nodes"; for d in /sys/class/drm/card*/device; do echo "$d $(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null)"; done
echo "== this shell may use: $(taskset -cp $$ 2>/dev/null)"; nproc
} >> $out 2>&1
run() {
  "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-c5 --no-c4 --no-heavy --no-dropin 2>>gpurun_out/numa_probe.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ms/step %.4f  stage sum %.4f  fwd %.4f' % (d['ms_per_step'], sum(v for k,v in d['stage_ms_per_view'].items() if k!='refl_bwd_tail'), d['forward_ms']))"
}
for rep in 1 2; do
  echo "-- unpinned" >> $out; run >> $out
  for n in /sys/devices/system/node/node*; do
    cpus=$(cat $n/cpulist)
    echo "-- taskset -c $cpus ($(basename $n))" >> $out; run taskset -c $cpus >> $out || echo "(failed)" >> $out
  done
done
cat $out
