# Round-4 profile set (run on the GPU box): kernel-trace stats + PMC passes of the headline command (C3, fused rasterize + reflect node) and of the
# C5 object, plus one memory-side pass (L2 request counts) for the streaming kernels.
#   GSR_COMMIT=<short hash> bash tests/prof_r04.sh TAG   -> gpurun_out/prof_r04_TAG/{bench.json, kt*/, pmc*/, r04_pmc_summary.json, c5_*}
# Summaries are copied into profiles/ by hand (profiles/INDEX.md).  rocprofv3 wraps `python3 bench.py ...` directly (no env / bash -c hop);
# counters are collected in their own runs with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
D=gpurun_out/prof_r04_$TAG
mkdir -p $R/$D
cd $R
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-c5 --no-c4 --no-heavy --no-dropin"
python3 bench.py --steps 50 --warmup 10 > $D/bench.json 2> $D/bench.err && tail -c 300 $D/bench.json || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D/kt -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-c5 --no-c4 --no-heavy --no-dropin > $D/kt.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D/kt_sync -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-c5 --no-c4 --no-heavy --no-dropin --no-full-step --sync-reflection-tail > $D/kt_sync.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D/kt_unfused -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-c5 --no-c4 --no-heavy --no-dropin --no-full-step --unfused > $D/kt_unfused.log 2>&1 || exit 1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $D/pmc$i -o p --output-format csv -- python3 bench.py $ARGS > $D/pmc$i.log 2>&1 || { echo "pmc set $i failed"; tail -3 $D/pmc$i.log; }
done
python3 tests/pmc_summary.py $D/r04_pmc_summary.json 1000000 1920 1080 "python bench.py $ARGS" $D/pmc1 $D/pmc2 $D/pmc3 $D/pmc4 > $D/pmc_summary.txt
# ---- C5: 5e6 Gaussians, variant G, anti-aliasing + inverse-depth backward, gradients through sinks
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D/c5_kt -o kt --output-format csv -- python3 bench.py --only-c5 --steps 10 > $D/c5_kt.log 2>&1 || exit 1
j=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES"; do
  j=$((j+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $D/c5_pmc$j -o p --output-format csv -- python3 bench.py --only-c5 --steps 5 > $D/c5_pmc$j.log 2>&1 || { echo "c5 pmc set $j failed"; }
done
python3 tests/pmc_summary.py $D/r04_c5_pmc_summary.json 5000000 1920 1080 "python bench.py --only-c5 --steps 5" $D/c5_pmc1 $D/c5_pmc2 $D/c5_pmc3 > $D/c5_pmc_summary.txt
find $D -name "*kernel_stats.csv" | head -6
