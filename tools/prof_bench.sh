# rocprofv3 kernel trace of the default bench.py run (the figures bench.py's roofline block must agree with).  The copies meant for
# profiles/ carry the identity of the binary: csrc_sha (rgcn_csrc_sha() of the library that ran) and the git head of the snapshot.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/prof_bench}
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python bench.py --no-cpu-baseline --no-configs > $OUT/bench_under_rocprof.json 2> $OUT/bench.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
SHA=$(python -c "import sys; sys.path.insert(0, 'torch-rgcn_amd'); from torch_rgcn import _native; print(_native.csrc_sha())")
HEAD=$(cat .git_head_for_profiles 2>/dev/null || echo unknown)
{ echo "# csrc_sha=$SHA git_head=$HEAD command: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-configs"; cat "$f"; } > $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv | cut -c1-150
