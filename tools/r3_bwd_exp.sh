#!/bin/bash
# round 3, fused-backward experiments at S1: round 2's staging kernel against the window kernel (16 / 8 waves per workgroup),
# HIP-event times + separate PMC passes (fabric requests, FETCH / WRITE sizes).   usage: tools/r3_bwd_exp.sh <outdir>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_bwd}
mkdir -p "$OUT"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python tools/kbench.py --what bwd --iters 20 > "$OUT/kb_$name.log" 2>&1
  grep -h "bwd_fused\|relu-masked" "$OUT/kb_$name.log" | sed "s/^/[$name] /"
}
run stage RGCN_BWD_KERNEL=stage
run win16 RGCN_BWD_NW=16
run win8 RGCN_BWD_NW=8
run win16bp RGCN_BWD_NW=16 RGCN_BWD_BPERM=1
# PMC: one pass per counter set, staging kernel and window kernel in separate processes
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for K in stage win; do
    RGCN_BWD_KERNEL=$K timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_${K}_$i" -o p -- python tools/kbench.py --what bwd --iters 2 > "$OUT/pmc_${K}_$i.log" 2>&1
  done
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = {}
for K in ("stage", "win"):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_{K}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bwd_fused_d16_kernel<4, true" in r["Kernel_Name"] or "bwd_win_d16_kernel<4, 16, 8, true, false" in r["Kernel_Name"] \
                    or "bwd_win_d16_kernel<4, 8, 4, true, false" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[K] = {k: sum(v) / len(v) for k, v in per.items()}
    d = res[K]
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["_traffic_GB_2xFETCH_plus_WRITE"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / 1e9
res["_how"] = "tools/r3_bwd_exp.sh: separate rocprofv3 --pmc passes over tools/kbench.py --what bwd at S1; means per launch of the atomic-flush kernel"
json.dump(res, open(out + "/pmc_bwd_stage_vs_win.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
