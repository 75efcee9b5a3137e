#!/bin/bash
# tools/micro/gather_window.hip on the GPU box: the sweep (ms per pass per window size, U = 4 / 8, 4 / 8 workgroups per CU), then the L2 hit
# rate and the fabric read requests per window size (rocprofv3 --pmc, separate passes) -> gpurun_out/gather_window/summary.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gather_window; mkdir -p $OUT
B=tools/micro/gather_window.bin
{
for U in 4 8; do for PC in 4 8; do $B 0 $U $PC; done; done
echo "# AM shape: 1,666,764 rows (107 MB), 13.6 M reads"
$B 0 4 8 1666764 13643406
} > $OUT/summary.txt 2>&1
for W in 1000000 65536 32768 16384; do
  i=0
  for SET in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "FETCH_SIZE"; do
    i=$((i+1))
    rm -rf $OUT/p$W.$i
    timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$W.$i -o p -- $B $W 4 8 > /dev/null 2>&1
  done
done
python - $OUT >> $OUT/summary.txt <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
for W in (1000000, 65536, 32768, 16384):
    c = collections.defaultdict(list)
    for f in glob.glob(f"{out}/p{W}.*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gather_kernel" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in c.items()}
    hit = m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))
    print(f"pmc window {W}: L2 hit {hit:.3f}  RDREQ {m.get('TCC_EA0_RDREQ_sum', 0)/1e6:.2f} M (128B {m.get('TCC_EA0_RDREQ_128B_sum', 0)/1e6:.2f} M)  FETCH_SIZE {m.get('FETCH_SIZE', 0)/1e6:.3f} GB(KB units/1e6)")
PY
cat $OUT/summary.txt
