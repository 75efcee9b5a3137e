#!/bin/bash
# Collect PMC counters in separate rocprofv3 passes (never combined with sys/hip traces) for the S1 kernels (forward spmm,
# fused backward, two-pass backward for comparison) and write profiles-style summaries.
# usage: tools/pmc_passes.sh <outdir>      -> <outdir>/pmc_detail.json, <outdir>/pmc_kernels.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  # the shipped S1 route (round 6: rgcn_spmm_blk_f32 on the soft-window plan, rgcn_bwd_own_f32) and, for comparison, round 5's
  # (wave-owned forward, 218-row block-tile backward: RGCN_SOFTWIN=0)
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/k$i" -o p -- python tools/s1_kernels.py --iters 3 > "$OUT/k$i.log" 2>&1
  RGCN_SOFTWIN=0 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/kl$i" -o p -- python tools/s1_kernels.py --iters 3 > "$OUT/kl$i.log" 2>&1
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/k*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
keys = {"spmm": "spmm_d16_kernel", "spmm_blk": "spmm_blk_d16_kernel", "bwd_own": "bwd_own_d16_kernel", "bwd_blk": "bwd_blk_d16_kernel<true, false, 1>"}
import os, subprocess
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "torch-rgcn_amd"))
from torch_rgcn import _native
head = open(".git_head_for_profiles").read().strip() if os.path.exists(".git_head_for_profiles") else "unknown"
meta = {"csrc_sha": _native.csrc_sha(), "git_head": head, "command": "tools/pmc_passes.sh (rocprofv3 --pmc, separate passes, tools/s1_kernels.py: the S1 forward and backward launches of the shipped route, and with RGCN_SOFTWIN=0)"}
detail = {"_meta": meta, "_how": "tools/pmc_passes.sh: separate rocprofv3 --pmc passes over tools/s1_kernels.py (S1 launches); means per launch. "
                  "GRBM_GUI_ACTIVE is summed over the 8 XCDs: MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)"}
kernels = {"_meta": meta}
for name, c in per.items():
    means = {k: sum(v) / len(v) for k, v in c.items()}
    for key, sub in keys.items():
        if sub in name:
            detail[key] = {k: v for k, v in means.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")}
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        kernels[name[:90]] = {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in c.items()
                              if k in ("FETCH_SIZE", "WRITE_SIZE")}
json.dump(detail, open(out + "/pmc_detail.json", "w"), indent=1, sort_keys=True)
json.dump(kernels, open(out + "/pmc_kernels.json", "w"), indent=1, sort_keys=True)
for key in keys:
    d = detail.get(key)
    if d and d.get("GRBM_GUI_ACTIVE"):
        print(key, "GUI", round(d["GRBM_GUI_ACTIVE"]), "RDREQ", round(d.get("TCC_EA0_RDREQ_sum", 0)), "LDS conflict frac",
              round(d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3), "MFMA busy frac",
              round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (d["GRBM_GUI_ACTIVE"] / 8 * 1024), 3), "SALU", round(d.get("SQ_INSTS_SALU", 0)),
              "VALU", round(d.get("SQ_INSTS_VALU", 0)))
PY
[ "${KEEP_RAW:-0}" = "1" ] || rm -rf "$OUT"/k[0-9]* "$OUT"/kl[0-9]*          # raw counter dumps: tens of MB (gpurun copies back 64 MiB at most)
