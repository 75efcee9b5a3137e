set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/exp5
mkdir -p $OUT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  RGCN_TILE_ROWS=64 RGCN_BWD_D=4 timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/k$i" -o p -- python tools/kbench.py --what bwd,spmm --iters 3 > $OUT/k$i.log 2>&1
done
python tools/pmc_summary.py $OUT/k1 $OUT/k2 $OUT/k3 $OUT/k4 $OUT/k5 $OUT/k6 > $OUT/summary.json 2>$OUT/summary.err || true
ls $OUT; head -c 600 $OUT/summary.err
