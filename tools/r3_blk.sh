#!/bin/bash
# block-tile backward kernel against the lean window kernel (same box), with ablations
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_blk}; mkdir -p "$OUT"
run() { name=$1; shift; env "$@" timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/$name.log" 2>&1; echo "$name $(grep -h 'bwd_fused atomic' "$OUT/$name.log" | sed 's/.*tile=/tile=/')"; grep -h "relu-masked\|rror" "$OUT/$name.log" | head -3; }
run blk RGCN_BWD_KERNEL=blk
run lean RGCN_BWD_KERNEL=lean
for A in ${ABLS:-2 8 16}; do run blk_abl$A RGCN_BWD_KERNEL=blk RGCN_BWD_ABL=$A; done
for T in ${ROWS:-128 192}; do run blk_rows$T RGCN_BWD_KERNEL=blk RGCN_BWD_TILE_ROWS=$T; done
