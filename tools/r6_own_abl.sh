#!/bin/bash
# ablations of rgcn_bwd_own_f32 on the ablation library (wrong results on request): which part of the loop costs what
cd "$GRAFT_REPO_ROOT"
export RGCN_HIP_LIB=$PWD/torch-rgcn_amd/torch_rgcn/lib/librgcn_hip_abl.so
for A in 0 2 4 6 8 16; do
  echo "== RGCN_BWD_ABL=$A"; RGCN_BWD_ABL=$A timeout 120 python tools/own_probe.py --rows ${ROWS:-652} --iters 6 2>&1 | grep bwd_own
done
