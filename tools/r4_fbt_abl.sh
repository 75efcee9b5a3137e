#!/bin/bash
# tile kernels of the featureless basis layer at AM size: ablations (needs make -C torch-rgcn_amd/csrc abl)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RGCN_HIP_LIB=$GRAFT_REPO_ROOT/torch-rgcn_amd/torch_rgcn/lib/librgcn_hip_abl.so
for A in ${ABLS:-0 1 2 3 4 8}; do
  RGCN_BWD_ABL=$A timeout 300 python tools/fbt_bench.py 2>&1 | grep "^abl"
done
