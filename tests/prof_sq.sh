cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_sq
cd $R
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/prof_sq/p$i -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-step > gpurun_out/prof_sq/p$i.log 2>&1 || echo "set $i failed"
done
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob('gpurun_out/prof_sq/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'render_' not in n: continue
        n=n[n.index('gsr::'):].split('(')[0]
        a=acc[n][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k,v in acc.items():
    print(k)
    for c,(s,n) in sorted(v.items()):
        print('   %-24s %16.0f' % (c, s/n))
PY
