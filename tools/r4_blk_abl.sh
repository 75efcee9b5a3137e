#!/bin/bash
# block-tile backward (round 4): ablations + barrier / epilogue time per wave, both forms, one box.  Needs make -C torch-rgcn_amd/csrc abl
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r4_blk_abl}; mkdir -p "$OUT"
export RGCN_HIP_LIB=$GRAFT_REPO_ROOT/torch-rgcn_amd/torch_rgcn/lib/librgcn_hip_abl.so
for P in 0 1; do
  for A in 0 2 8 10 16; do
    RGCN_BWD_BLK_PIPE=$P RGCN_BWD_ABL=$A timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/p${P}_a$A.log" 2>&1
    echo "pipe=$P abl=$A $(grep -h 'bwd_fused atomic' "$OUT/p${P}_a$A.log" | sed 's/.*relerr/relerr/') $(grep -h 'per wave' "$OUT/p${P}_a$A.log")"
  done
done
