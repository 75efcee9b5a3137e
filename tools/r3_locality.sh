#!/bin/bash
# profiles/r03_locality.json: kernel times + fabric read requests per message with / without the locality relabelling
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r3_loc}; mkdir -p "$OUT"
for GR in s1 zipf; do
  ORD="none,degree,rcm"
  timeout 1500 python tools/locality_bench.py --graph $GR --orders $ORD > "$OUT/times_$GR.jsonl" 2> "$OUT/times_$GR.err"
  for O in none degree rcm; do
    timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_${GR}_$O" -o p -- \
      python tools/locality_bench.py --graph $GR --orders $O --iters 2 > "$OUT/pmc_${GR}_$O.log" 2>&1
  done
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = {"_how": "tools/r3_locality.sh: tools/locality_bench.py (HIP-event medians) + rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum per order; "
               "requests per message = fabric read requests of the kernel / messages of the graph"}
for gr in ("s1", "zipf"):
    rows = [json.loads(l) for l in open(f"{out}/times_{gr}.jsonl") if l.startswith("{")]
    for r in rows:
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{out}/pmc_{gr}_{r['order']}/**/*counter_collection.csv", recursive=True):
            for c in csv.DictReader(open(f)):
                per[c["Kernel_Name"]][c["Counter_Name"]].append(float(c["Counter_Value"]))
        k = {}
        for name, cs in per.items():
            for key, sub in (("spmm", "spmm_d16_kernel"), ("bwd", "bwd_win2_d16_kernel"), ("spmm_scatter", "spmm_scatter_d16"), ("segment_gather_sum", "segment_gather_sum"),
                             ("bwd_scatter_dw", "bwd_scatter_dw_d16")):
                if sub in name and "TCC_EA0_RDREQ_sum" in cs:
                    rd = sum(cs["TCC_EA0_RDREQ_sum"]) / len(cs["TCC_EA0_RDREQ_sum"])
                    hit = sum(cs["TCC_HIT_sum"]) / len(cs["TCC_HIT_sum"]); miss = sum(cs["TCC_MISS_sum"]) / len(cs["TCC_MISS_sum"])
                    k[key] = {"fabric_read_requests": round(rd), "requests_per_message": round(rd / r["messages"], 3), "l2_hit_rate": round(hit / max(hit + miss, 1), 3)}
        r["pmc"] = k
    res[gr] = rows
json.dump(res, open(out + "/locality.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
