#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_pipe}
mkdir -p "$OUT"
for P in 0 1; do for A in 0 4 1 5 16; do
  RGCN_BWD_KERNEL=win1 RGCN_BWD_PIPE=$P RGCN_BWD_ABL=$A timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/p${P}_$A.log" 2>&1
  echo "win1 PIPE=$P ABL=$A $(grep -h 'bwd_fused atomic' "$OUT/p${P}_$A.log" | sed 's/.*relerr/relerr/')"
done; done
