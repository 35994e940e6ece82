# Development aid: time bench.py stages for every variant library built by tests/ab_build.py (run on the GPU box).
#   [REPS=n] bash tests/ab_run.sh [extra bench args]  -> gpurun_out/ab.txt
# REPS > 1 interleaves the variants (A B C A B C ...): the boxes drift by several per cent between runs, compare minima.
mkdir -p gpurun_out
: > gpurun_out/ab.txt
for rep in $(seq 1 ${REPS:-1}); do
for lib in gaussian-splatting-reflection_amd/libgsr_hip.so gaussian-splatting-reflection_amd/csrc/_ab/lib_*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> gpurun_out/ab.txt
  GSR_BINDING=ctypes GSR_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step "$@" 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms/step %.4f fwd_ms %.4f' % (d['ms_per_step'], d['forward_ms']), {k:v for k,v in d['stage_ms_per_view'].items() if 'render' in k or 'preprocess' in k or 'sort' in k or 'refl' in k or 'emit' in k or 'scan' in k or 'ranges' in k})
if 'c5' in d: print('   c5 %.4f' % d['c5']['ms_per_step'], d['c5']['stage_ms_per_step'])
" >> gpurun_out/ab.txt || echo "FAILED" >> gpurun_out/ab.txt
done
done
python - <<'PY' >> gpurun_out/ab.txt
import re
best = {}
name = None
for line in open("gpurun_out/ab.txt"):
    if line.startswith("== "):
        name = line[3:].strip().split("/")[-1]
    m = re.match(r"ms/step ([0-9.]+) fwd_ms ([0-9.]+)", line)
    if m and name:
        b = best.setdefault(name, [9e9, 9e9, 0])
        b[0], b[1], b[2] = min(b[0], float(m.group(1))), min(b[1], float(m.group(2))), b[2] + 1
print("-- minima over runs")
for k, v in best.items():
    print("%-28s ms/step %.4f fwd %.4f (%d runs)" % (k, v[0], v[1], v[2]))
PY
cat gpurun_out/ab.txt
