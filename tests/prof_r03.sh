# Round-3 profile set (run on the GPU box): kernel-trace stats + PMC passes of the headline command (C3) and of the C5 object.
#   GSR_COMMIT=<short hash> bash tests/prof_r03.sh TAG   -> gpurun_out/prof_r03_TAG/{bench.json, kt*/, pmc*/, r03_pmc_summary.json, c5_*}
# Summaries are copied into profiles/ by hand (profiles/INDEX.md).  rocprofv3 wraps `python3 bench.py ...` directly (no env / bash -c hop).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
D=gpurun_out/prof_r03_$TAG
mkdir -p $R/$D
cd $R
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-c5 --no-c4 --no-dropin"
python3 bench.py --steps 50 --warmup 10 > $D/bench.json 2> $D/bench.err && tail -c 400 $D/bench.json || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --views 8 --steps 10 --warmup 3 --no-cpu-baseline --no-c5 --no-dropin > $D/bench_views8.json 2> $D/bench_views8.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D/kt -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-c5 --no-c4 --no-dropin > $D/kt.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D/kt_sync -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-c5 --no-c4 --no-dropin --no-full-step --sync-reflection-tail > $D/kt_sync.log 2>&1 || exit 1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $D/pmc$i -o p --output-format csv -- python3 bench.py $ARGS > $D/pmc$i.log 2>&1 || { echo "pmc set $i failed"; exit 1; }
done
python3 tests/pmc_summary.py $D/r03_pmc_summary.json 1000000 1920 1080 "python bench.py $ARGS" $D/pmc1 $D/pmc2 $D/pmc3 > $D/pmc_summary.txt
# ---- C5: 5e6 Gaussians, variant G, anti-aliasing + inverse-depth backward
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D/c5_kt -o kt --output-format csv -- python3 bench.py --only-c5 --steps 10 > $D/c5_kt.log 2>&1 || exit 1
j=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES"; do
  j=$((j+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $D/c5_pmc$j -o p --output-format csv -- python3 bench.py --only-c5 --steps 5 > $D/c5_pmc$j.log 2>&1 || { echo "c5 pmc set $j failed"; exit 1; }
done
python3 tests/pmc_summary.py $D/r03_c5_pmc_summary.json 5000000 1920 1080 "python bench.py --only-c5 --steps 5" $D/c5_pmc1 $D/c5_pmc2 $D/c5_pmc3 > $D/c5_pmc_summary.txt
find $D -name "*kernel_stats.csv" | head -5
