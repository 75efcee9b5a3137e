#!/bin/bash
# SQ counters of backward-kernel variants.  usage: CHUNKS=1.52e6 tools/r3_pmc2.sh <outdir> <tag> ENV=.. ENV=..
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; TAG=$2; shift; shift
mkdir -p "$OUT"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_IFETCH" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/${TAG}_$i" -o p -- python tools/kbench.py --what bwd --iters 2 > "$OUT/${TAG}_$i.log" 2>&1
done
python - "$OUT" "$TAG" "${CHUNKS:-1.863e6}" <<'PY'
import collections, csv, glob, json, sys
out, tag = sys.argv[1:3]
nchunks = float(sys.argv[3])
per = collections.defaultdict(list)
for f in glob.glob(f"{out}/{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bwd_blk" in n or (("bwd_win" in n or "bwd_lean_d16" in n or "bwd_fused_d16" in n) and "true" in n.split("<")[1][:30]):
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = {k: sum(v) / len(v) for k, v in per.items()}
json.dump(d, open(f"{out}/{tag}.json", "w"), indent=1, sort_keys=True)
g = d.get("GRBM_GUI_ACTIVE", 0) / 8
wc = d.get("SQ_WAVE_CYCLES", 1)
print(tag, "cycles/XCD %.0f" % g, "wave-cyc(quad) %.3g" % wc, "| of wave time: active %.2f wait_any %.2f wait_inst %.2f wait_lds %.3f" % (
    d.get("SQ_ACTIVE_INST_ANY", 0) / wc, d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_WAIT_INST_LDS", 0) / wc),
    "| per SIMD busy: VALU %.2f MFMA %.2f LDS(perCU) %.2f SALU(perCU) %.2f" % (
    d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / max(g, 1), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(g, 1),
    d.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / max(g, 1), d.get("SQ_INSTS_SALU", 0) / 256 / max(g, 1)),
    "| insts/chunk VALU %.0f SALU %.0f LDS %.1f MFMA %.1f VMEM %.1f" % tuple(d.get(k, 0) / nchunks for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD")))
PY
