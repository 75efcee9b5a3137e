#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_lean2}; mkdir -p "$OUT"
run() { name=$1; shift; env "$@" timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/$name.log" 2>&1; echo "$name $(grep -h 'bwd_fused atomic' "$OUT/$name.log" | sed 's/.*relerr/relerr/')"; }
run lean16 RGCN_BWD_KERNEL=lean
run lean8 RGCN_BWD_KERNEL=lean RGCN_BWD_NW=8
run win2_16 RGCN_BWD_KERNEL=win
run win2_8 RGCN_BWD_KERNEL=win RGCN_BWD_NW=8
