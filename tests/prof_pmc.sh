# PMC collection for the bench step (run on the GPU box): bash tests/prof_pmc.sh LABEL "SET1" "SET2" ...   (each SET = counters of one pass)
# Output: gpurun_out/pmc_LABEL/pN/... and a per-kernel summary gpurun_out/pmc_LABEL.txt (tile kernels + per-Gaussian kernels).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LABEL=$1; shift
mkdir -p $R/gpurun_out/pmc_$LABEL
cd $R
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_$LABEL/p$i -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-step > gpurun_out/pmc_$LABEL/p$i.log 2>&1 || echo "set $i failed"
done
python - "$LABEL" > gpurun_out/pmc_$LABEL.txt <<'PY'
import csv, glob, collections, sys
label = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmc_%s/p*/**/*counter_collection.csv' % label, recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'gsr::' not in n: continue
        n = n[n.index('gsr::'):].split('(')[0]
        if not any(k in n for k in ('render_', 'preprocess')): continue
        a = acc[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, v in sorted(acc.items()):
    print(k)
    for c, (s, n) in sorted(v.items()):
        print('   %-28s %16.0f' % (c, s / n))
PY
cat gpurun_out/pmc_$LABEL.txt
