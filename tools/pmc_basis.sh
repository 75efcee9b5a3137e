#!/bin/bash
# matrix-core utilisation of the dense step of the basis path (tools/basis_bench.py) -- separate rocprofv3 --pmc passes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc_basis}
mkdir -p "$OUT"
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/k$i" -o p -- python tools/basis_bench.py > "$OUT/k$i.log" 2>&1
done
python tools/pmc_summary.py $OUT/k1 $OUT/k2 $OUT/k3 > $OUT/summary.json
