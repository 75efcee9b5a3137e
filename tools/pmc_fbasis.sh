#!/bin/bash
# HBM-side bytes of the source-major featureless-basis kernels and the score-all kernel: FETCH_SIZE / WRITE_SIZE in their
# own rocprofv3 --pmc passes (never combined with traces), summarised by tools/pmc_summary.py.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_fb
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/fb_$C -o p -- python tools/fbasis_bench.py > /dev/null 2>&1 </dev/null
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/ev_$C -o p -- python tools/eval_bench.py --no-cpu > /dev/null 2>&1 </dev/null
done
python tools/pmc_summary.py $OUT/fb_FETCH_SIZE $OUT/fb_WRITE_SIZE $OUT/ev_FETCH_SIZE $OUT/ev_WRITE_SIZE > $OUT/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_fb/summary.json"))
for k,v in d.items():
    if any(t in k for t in ("fbasis","gather_rows","score_all","basis_aggregate","basis_dcomps")):
        print(k[:70], {c: round(x["mean"]/1024/1024,1) for c,x in v.items()}, "GiB-ish units: KB->GB" )
PY
