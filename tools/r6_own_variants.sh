#!/bin/bash
# rgcn_bwd_own_f32: tile heights on the shipped shape (16 waves x 8 units x 3 chunks per trip); other shapes are built with
#   hipcc ... -DRGCN_OWN_NW=.. -DRGCN_OWN_K=.. -DRGCN_OWN_U=.. -c rgcn_bwd_own.hip   and linked into a library of their own (RGCN_HIP_LIB)
cd "$GRAFT_REPO_ROOT"
for C in 767 600 500; do
  echo "== RGCN_OWN_ROWS_CAP=$C"
  RGCN_OWN_ROWS_CAP=$C timeout 300 python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['roofline']['forward']['avg_launch_ms'], d['roofline']['backward']['avg_launch_ms'])"
done
