#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_sg}; mkdir -p "$OUT"
run() { name=$1; shift; env "$@" timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/$name.log" 2>&1; echo "$name $(grep -h 'bwd_fused atomic' "$OUT/$name.log" | sed 's/.*relerr/relerr/')"; grep -h PROF "$OUT/$name.log"; }
run sg16 RGCN_BWD_KERNEL=win1
run sg8 RGCN_BWD_KERNEL=win1 RGCN_BWD_SG=8
run sg4 RGCN_BWD_KERNEL=win1 RGCN_BWD_SG=4
run sg4dw3 RGCN_BWD_KERNEL=win1 RGCN_BWD_SG=43
run sg16c RGCN_BWD_KERNEL=win1 RGCN_BWD_ABL=256
run sg8c RGCN_BWD_KERNEL=win1 RGCN_BWD_SG=8 RGCN_BWD_ABL=256
run sg4c RGCN_BWD_KERNEL=win1 RGCN_BWD_SG=4 RGCN_BWD_ABL=256
run abl4 RGCN_BWD_KERNEL=win1 RGCN_BWD_ABL=4
run stage RGCN_BWD_KERNEL=stage
