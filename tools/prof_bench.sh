# rocprofv3 kernel trace of the default bench.py run (the figures bench.py's roofline block must agree with)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/prof_bench}
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python bench.py --no-cpu-baseline --no-configs > $OUT/bench_under_rocprof.json 2> $OUT/bench.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv | cut -c1-150
