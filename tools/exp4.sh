set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp4
for CFG in "128 4" "64 4"; do set -- $CFG
  RGCN_TILE_ROWS=$1 RGCN_BWD_D=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp4/t$1 -o p -- python tools/kbench.py --what bwd,spmm,wtiled --iters 10 > gpurun_out/exp4/t$1.log 2>&1
  f=$(find gpurun_out/exp4/t$1 -name "*kernel_stats.csv" | head -1)
  echo "== tile $1 D $2"; head -12 "$f" | cut -c1-200
done
