#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_pc}; mkdir -p "$OUT"
run() { name=$1; shift; env "$@" timeout 120 python tools/kbench.py --what bwd --iters 20 > "$OUT/$name.log" 2>&1; echo "$name rc=$? $(grep -h 'bwd_fused atomic' "$OUT/$name.log" | sed 's/.*relerr/relerr/')"; grep -h "PCPROF\|relu-masked\|rror" "$OUT/$name.log" | head -3; }
run pc RGCN_BWD_KERNEL=pc
run win2 RGCN_BWD_KERNEL=win
run lean RGCN_BWD_KERNEL=lean
for A in ${2:-}; do run pc_abl$A RGCN_BWD_KERNEL=pc RGCN_BWD_ABL=$A; done
