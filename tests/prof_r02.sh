# Round-2 profile set for the headline command (run on the GPU box): kernel-trace stats + PMC passes, summaries copied to profiles/ by hand.
#   bash tests/prof_r02.sh TAG        -> gpurun_out/prof_r02_TAG/{bench.json, kt/, pmc*/, r02_pmc_summary.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
D=gpurun_out/prof_r02_$TAG
mkdir -p $R/$D
cd $R
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-c5"
python bench.py --steps 50 --warmup 10 > $D/bench.json 2> $D/bench.err && tail -c 600 $D/bench.json &&
rocprofv3 --kernel-trace --stats -d $D/kt -o kt --output-format csv -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-c5 > $D/kt.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $D/pmc$i -o p --output-format csv -- python bench.py $ARGS > $D/pmc$i.log 2>&1 || echo "pmc set $i failed"
done
python tests/pmc_summary.py $D/r02_pmc_summary.json 1000000 1920 1080 "python bench.py $ARGS" $D/pmc1 $D/pmc2 $D/pmc3 $D/pmc4 > $D/pmc_summary.txt
find $D -name "*kernel_stats.csv" | head -3
