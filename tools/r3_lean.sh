#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_lean}; mkdir -p "$OUT"
run() { name=$1; shift; env "$@" timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/$name.log" 2>&1; echo "$name $(grep -h 'bwd_fused atomic' "$OUT/$name.log" | sed 's/.*relerr/relerr/')"; grep -h "partial\|relu-masked\|reproducible\|rror" "$OUT/$name.log" | sed 's/.*relerr/    relerr/' | head -5; }
run lean RGCN_BWD_KERNEL=lean
run win2 RGCN_BWD_KERNEL=win
run stage RGCN_BWD_KERNEL=stage
for A in 1 4 5 8 16; do run lean_abl$A RGCN_BWD_KERNEL=lean RGCN_BWD_ABL=$A; done
