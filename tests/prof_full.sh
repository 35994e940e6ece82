cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_full
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_full/kt -o kt --output-format csv -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_full/kt.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_full/kt/kt_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:25]:
    print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2),'ms', round(float(r['AverageNs'])/1e3,1),'us')
PY
