#!/bin/bash
# kernel stats of the AM block-diagonal line's step (tools/nc_step_trace.py am under rocprofv3)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/am_block_trace; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python tools/nc_step_trace.py am > "$OUT/log.txt" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-100s calls %5s avg %8.1f us  per step %7.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 55e3))
PY
