cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_tr
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tr/kt -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > gpurun_out/prof_tr/kt.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_tr/kt/kt_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'deferred_refl_bwd' in r['Kernel_Name']]
a=idx[-3]-2
t0=int(rows[a]['Start_Timestamp']); prev=t0
for r in rows[a:a+30]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f}us gap {(s-prev)/1e3:7.1f} dur {(e-s)/1e3:8.1f}  {r['Kernel_Name'][:90]}")
    prev=e
    if 'surfel_render_bwd' in r['Kernel_Name']: break
PY
