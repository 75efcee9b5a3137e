set -u
mkdir -p gpurun_out/exp1
L=gpurun_out/exp1/log.txt
for T in 1 2 4 8; do RGCN_WGRAD_TILES=$T python tools/kbench.py --what wtiled --iters 10 2>&1 | grep -v setup >> $L; done
for TR in 128 256 512; do for U in 4 8; do RGCN_TILE_ROWS=$TR RGCN_SPMM_U=$U python tools/kbench.py --what spmm --iters 10 2>&1 | grep -v setup >> $L; done; done
cat $L
