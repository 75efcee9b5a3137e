#!/bin/bash
# block-tile backward at S1: tile height sweep (quads per wave and tile = the granularity of the tile-end barrier)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for T in ${ROWS:-186 196 206 212 218 224}; do
  RGCN_BWD_TILE_ROWS=$T timeout 300 python tools/kbench.py --what bwd --iters 30 2>&1 | grep "bwd_fused atomic" | sed 's/.*bwd_fused atomic/rows/'
done
