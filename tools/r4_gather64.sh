#!/bin/bash
# tools/micro/gather64: time per load flavour, then the L2's fabric request sizes per flavour (rocprofv3 --pmc, kernel trace only)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -x tools/micro/gather64.bin ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o tools/micro/gather64.bin tools/micro/gather64.hip
tools/micro/gather64.bin
for V in 0 1 2 3 5; do
  OUT=gpurun_out/gather64_v$V; rm -rf "$OUT"; mkdir -p "$OUT"
  timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$OUT" -o p -- tools/micro/gather64.bin $V > /dev/null 2>&1
  python - "$OUT" $V <<'PY'
import csv, glob, sys, collections
out, v = sys.argv[1:3]
per = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gather_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant", v, {k: round(sum(x) / len(x)) for k, x in sorted(per.items())})
PY
done
