cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_eval
mkdir -p $OUT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/k$i" -o p -- python tools/eval_bench.py --no-cpu > /dev/null 2>&1 </dev/null
done
python tools/pmc_summary.py $OUT/k1 $OUT/k2 $OUT/k3 > $OUT/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_eval/summary.json"))
for k,v in d.items():
    if "score_all" in k:
        print(k[:60]); print({c:round(x["mean"]) for c,x in v.items()})
PY
