#!/bin/bash
# rocprofv3 kernel stats of the AM-as-shipped line (tools/config_bench.py --lines amshipped)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/am_prof}
mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o am -- python tools/config_bench.py --lines amshipped > "$OUT/line.json" 2> "$OUT/err.log"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-90s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -c 600 "$OUT/line.json"
