#!/bin/bash
# kernel sequence of one eager step of the AIFB / MUTAG lines (rocprofv3 --kernel-trace): names in launch order with durations
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for W in aifb mutag; do
  OUT=gpurun_out/nc_trace_$W; mkdir -p "$OUT"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python tools/nc_step_trace.py $W > "$OUT/log.txt" 2>&1
  python - "$OUT" "$W" <<'PY'
import csv, glob, sys
out, w = sys.argv[1:3]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# one step ends with the optimiser's kernels (a _foreach step counter + the fused Adam): boundaries = an Adam kernel followed by another kernel
idx = [i for i, n in enumerate(names) if "multi_tensor_apply" in n]
ends = [i for i in idx if i + 1 >= len(names) or "multi_tensor_apply" not in names[i + 1]]
a, b = ends[-2] + 1, ends[-1] + 1
us = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(w, "launches per step:", b - a, " kernel time %.1f us" % sum(us(r) for r in rows[a:b]))
for r in rows[a:b]:
    print("  %7.1f us  %s" % (us(r), r["Kernel_Name"][:110]))
PY
done
