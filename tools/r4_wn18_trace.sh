#!/bin/bash
# kernel stats of the WN18 line's step (rocprofv3 --kernel-trace --stats over tools/config_bench.py --lines wn18)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/wn18_trace; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python tools/config_bench.py --lines wn18 > "$OUT/line.json" 2> "$OUT/err.txt"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:30]:
    print("%-100s calls %5s avg %8.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -c 900 "$OUT/line.json"
