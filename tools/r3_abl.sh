#!/bin/bash
# window-kernel ablations at S1 (timing only): which part of a chunk's work bounds the kernel?   usage: tools/r3_abl.sh <outdir> [kernel] [ablations...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_abl}
KERN=${2:-win}
shift; shift
ABLS=${*:-0 1 2 3 4 5 8 11 16}
mkdir -p "$OUT"
for A in $ABLS; do
  RGCN_BWD_KERNEL=$KERN RGCN_BWD_ABL=$A timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/abl_${KERN}_$A.log" 2>&1
  echo "$KERN ABL=$A $(grep -h 'bwd_fused atomic' "$OUT/abl_${KERN}_$A.log" | sed 's/.*relerr/relerr/')"
  { [ "$A" = "0" ] || [ "$A" = "128" ] || [ "$A" = "256" ]; } && grep -h "PROF\|partial\|relu-masked\|reproducible\|Error\|error" "$OUT/abl_${KERN}_$A.log" | sed 's/.*relerr/    relerr/'
done
