set -u
mkdir -p gpurun_out/exp3
L=gpurun_out/exp3/log5.txt
: > $L
for CFG in "64 4 0" "64 4 1" "128 4 0" "128 4 1" "96 2 1"; do set -- $CFG; echo "TILE=$1 D=$2 NT=$3" >> $L; RGCN_TILE_ROWS=$1 RGCN_BWD_D=$2 RGCN_BWD_NT=$3 python tools/kbench.py --what bwd --iters 10 2>&1 | grep "bwd_fused " >> $L; done
cat $L
