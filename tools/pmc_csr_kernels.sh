#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) and request mix of the CSR kernels of the diagonal and
# block-diagonal layers on the AM-shaped graph, next to their kernel-trace durations.
# usage: tools/pmc_csr_kernels.sh <outdir>   -> <outdir>/pmc_csr_kernels.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc_csr}
mkdir -p "$OUT"
cat > "$OUT/drive.py" <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch
import diag_bench, block_probe
diag_bench.case("AM-shaped e-rgcn layer 1 (emb 32)", 1_666_764, 133, 5_988_321, 32)
os.environ["RGCN_BLOCK_PATH"] = "2"
block_probe.ROUTES = {"block_kernels": ("2", "1")}
block_probe.nc_case("AM-shaped, d=16, 4 blocks of 4x4", 1_666_764, 133, 5_988_321, 16, 4)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o p -- python "$OUT/drive.py" > "$OUT/trace.log" 2>&1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/k$i" -o p -- python "$OUT/drive.py" > "$OUT/k$i.log" 2>&1
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/k*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(out + "/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
res = {"_how": "tools/pmc_csr_kernels.sh: AM-shaped graph (N = 1,666,764, 13,643,406 messages); counters = mean per launch over separate "
               "rocprofv3 --pmc passes; HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 when the reads are 128-byte requests "
               "(MI355X_MICROARCH.md, HBM section); durations from a --kernel-trace --stats run of the same driver"}
for name, c in per.items():
    if not any(k in name for k in ("diag_csr_kernel", "diag_wgrad_kernel", "block_csr", "block_wgrad_kernel")):
        continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    calls, us = dur.get(name, (0, 0.0))
    e = {"launches_traced": calls, "avg_us": round(us, 1), **{k: round(v) for k, v in m.items()}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        share128 = m.get("TCC_EA0_RDREQ_128B_sum", 0) / max(m.get("TCC_EA0_RDREQ_sum", 1), 1)
        e["share_of_128B_read_requests"] = round(share128, 3)
        e["hbm_bytes_per_launch"] = round(((1 + share128) * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
        if us:
            e["hbm_GBs"] = round(e["hbm_bytes_per_launch"] / us / 1e3, 1)
    res[name[:100]] = e
json.dump(res, open(out + "/pmc_csr_kernels.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1)[:3000])
PY
