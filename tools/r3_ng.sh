#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_ng}; mkdir -p "$OUT"
run() { name=$1; shift; env "$@" timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/$name.log" 2>&1; echo "$name $(grep -h 'bwd_fused atomic' "$OUT/$name.log" | sed 's/.*relerr/relerr/')"; grep -h PROF "$OUT/$name.log"; }
run w2_ng2 RGCN_BWD_KERNEL=win
run w2_ng3 RGCN_BWD_KERNEL=win RGCN_BWD_ABL=1000
run w2_ng3c RGCN_BWD_KERNEL=win RGCN_BWD_ABL=1256
run w2_abl4 RGCN_BWD_KERNEL=win RGCN_BWD_ABL=4
run stage RGCN_BWD_KERNEL=stage
