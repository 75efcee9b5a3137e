#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_occ}; mkdir -p "$OUT"
run() { name=$1; shift; env "$@" timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/$name.log" 2>&1; echo "$name $(grep -h 'bwd_fused atomic' "$OUT/$name.log" | sed 's/.*relerr/relerr/')"; }
run win2_nw16 RGCN_BWD_KERNEL=win
run win2_nw8_2wg RGCN_BWD_KERNEL=win RGCN_BWD_NW=8
run win2_nw8_1wg RGCN_BWD_KERNEL=win RGCN_BWD_NW=8 RGCN_BWD_LDS_PAD=8192
run win2_nw8_1wg_abl4 RGCN_BWD_KERNEL=win RGCN_BWD_NW=8 RGCN_BWD_LDS_PAD=8192 RGCN_BWD_ABL=4
run win2_nw8_2wg_t32 RGCN_BWD_KERNEL=win RGCN_BWD_NW=8 RGCN_BWD_TILE_ROWS=32
run stage_t64 RGCN_BWD_KERNEL=stage
run stage_t128 RGCN_BWD_KERNEL=stage RGCN_BWD_TILE_ROWS=128
