#!/bin/bash
# ONE profiling wrapper for the GPU box (run through gpurun).  Kernel traces and PMC counters are always separate rocprofv3 runs
# (--pmc only ever with --kernel-trace); every summary carries the identity of the binary (rgcn_csrc_sha) and the git head.
#
#   tools/prof.sh lines  <outdir> <line> [<line> ...]   kernel stats + PMC per config line of tools/config_bench.py --lines
#                                                       (amshipped am s2 wn18 aifb mutag s1ii s1iii) ->
#                                                       <outdir>/<line>_kernel_stats.csv, <outdir>/<line>_pmc.json
#   tools/prof.sh bench  <outdir>                       kernel stats of the default bench.py run -> <outdir>/kernel_stats.csv
#   tools/prof.sh s1pmc  <outdir>                       PMC passes over tools/kbench.py (S1 launches) -> pmc_detail.json, pmc_kernels.json
#   tools/prof.sh trace  <outdir> -- <command ...>      kernel stats of any command (top kernels printed)
#   tools/prof.sh pmc    <outdir> -- <command ...>      the PMC passes over any command -> <outdir>/pmc.json
#   tools/prof.sh seq    <outdir> -- <command ...>      kernels of the command in launch order with durations (last 120)
#   tools/prof.sh env    <outdir> <VAR> "<v1 v2 ..>" -- <command ...>   the command once per value of VAR (sweeps, ablation bits);
#                                                       ABL=1 points RGCN_HIP_LIB at the ablation library
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MODE=${1:?mode}; OUT=${2:?outdir}; shift 2
mkdir -p "$OUT"
SHA=$(python -c "import sys; sys.path.insert(0, 'torch-rgcn_amd'); from torch_rgcn import _native; print(_native.csrc_sha())")
HEAD=$(cat .git_head_for_profiles 2>/dev/null || echo unknown)
PMC_SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
          "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"
          "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
          "GRBM_GUI_ACTIVE GRBM_TA_BUSY")

trace_run() {   # <dir> <command...>: rocprofv3 --kernel-trace --stats
  local d=$1; shift
  rm -rf "$d"; mkdir -p "$d"
  timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- "$@" > "$d/stdout.txt" 2> "$d/stderr.txt"
}
pmc_runs() {    # <dir> <command...>: one rocprofv3 --pmc pass per counter set
  local d=$1; shift
  local i=0
  for SET in "${PMC_SETS[@]}"; do
    i=$((i+1))
    rm -rf "$d/k$i"; mkdir -p "$d/k$i"
    timeout ${PROF_TIMEOUT:-900} rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$d/k$i" -o p -- "$@" > "$d/k$i/stdout.txt" 2> "$d/k$i/stderr.txt"
  done
}
summarise() {   # <trace dir or ""> <pmc dir or ""> <stats csv out or ""> <pmc json out or ""> <command string>
  python - "$1" "$2" "$3" "$4" "$SHA" "$HEAD" "$5" <<'PY'
import collections, csv, glob, json, sys
trace, pmc, stats_out, pmc_out, sha, head, cmd = sys.argv[1:8]
dur = {}
if trace:
    fs = glob.glob(trace + "/**/*kernel_stats.csv", recursive=True)
    if fs:
        rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: -float(r["TotalDurationNs"]))
        tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
        for r in rows:
            dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
        if stats_out:
            with open(stats_out, "w") as f:
                f.write(f"# csrc_sha={sha} git_head={head} command: rocprofv3 --kernel-trace --stats -- {cmd}\n")
                f.write(open(fs[0]).read())
        for r in rows[:24]:
            print("%-96s calls %6s avg %9.1f us %5.1f%%" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
if pmc and pmc_out:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pmc + "/k*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"_meta": {"csrc_sha": sha, "git_head": head, "command": cmd,
                     "how": "tools/prof.sh: separate rocprofv3 --pmc passes (one counter set each, --kernel-trace only); means per launch; "
                            "hbm_bytes_per_launch = (FETCH_SIZE x (1 + share of 128-byte read requests) + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE "
                            "tallies a 128-byte request at 64 bytes, MI355X_MICROARCH.md HBM section); avg_us from the kernel-trace run of the same command"}}
    for name, c in per.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        calls, us = dur.get(name, (0, 0.0))
        if us and us < 3.0 and m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0) < 256:     # (fills, scans, ... of a few microseconds)
            continue
        e = {"launches_traced": calls, "avg_us": round(us, 2), **{k: round(v, 1) for k, v in sorted(m.items())}}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            share = m.get("TCC_EA0_RDREQ_128B_sum", 0.0) / max(m.get("TCC_EA0_RDREQ_sum", 0.0), 1.0)
            e["share_of_128B_read_requests"] = round(share, 3)
            e["hbm_bytes_per_launch"] = round(((1 + share) * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
            if us:
                e["hbm_GBs"] = round(e["hbm_bytes_per_launch"] / us / 1e3, 1)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 3)
        if m.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
        res[name[:110]] = e
    json.dump(res, open(pmc_out, "w"), indent=1, sort_keys=True)
    big = sorted(((k, v) for k, v in res.items() if k != "_meta"), key=lambda kv: -kv[1].get("avg_us", 0) * max(kv[1].get("launches_traced", 1), 1))[:12]
    for k, v in big:
        print("PMC %-70s %8.1f us  hbm %8.3f GB  %7s GB/s  L2 hit %s  lds-conflict %s" % (k[:70], v.get("avg_us", 0), v.get("hbm_bytes_per_launch", 0) / 1e9,
              v.get("hbm_GBs"), v.get("l2_hit_rate"), v.get("lds_bank_conflict_frac")))
PY
}

case "$MODE" in
  lines)
    for L in "$@"; do
      CMD="python tools/config_bench.py --quick --lines $L"
      echo "=== line $L: kernel trace"
      trace_run "$OUT/$L/trace" $CMD
      tail -c 400 "$OUT/$L/trace/stdout.txt" | head -c 400; echo
      if [ "${NO_PMC:-0}" != "1" ]; then echo "=== line $L: PMC passes"; pmc_runs "$OUT/$L" $CMD; fi
      summarise "$OUT/$L/trace" "$OUT/$L" "$OUT/${L}_kernel_stats.csv" "$OUT/${L}_pmc.json" "$CMD"
      tail -c 3000 "$OUT/$L/trace/stdout.txt" > "$OUT/${L}_line_tail.txt" 2>/dev/null
      [ "${KEEP_RAW:-0}" = "1" ] || rm -rf "$OUT/$L"          # the raw traces run to tens of MB per line: gpurun copies back 64 MiB at most
    done ;;
  bench)
    CMD="python bench.py --no-cpu-baseline --no-configs"
    trace_run "$OUT/trace" $CMD
    cp "$OUT/trace/stdout.txt" "$OUT/bench_under_rocprof.json"
    summarise "$OUT/trace" "" "$OUT/kernel_stats.csv" "" "$CMD"
    [ "${KEEP_RAW:-0}" = "1" ] || rm -rf "$OUT/trace" ;;
  s1pmc)
    exec bash tools/pmc_passes.sh "$OUT" ;;
  trace)
    [ "${1:-}" = "--" ] && shift
    trace_run "$OUT/trace" "$@"
    summarise "$OUT/trace" "" "$OUT/kernel_stats.csv" "" "$*"
    tail -c 1500 "$OUT/trace/stdout.txt" ;;
  pmc)
    [ "${1:-}" = "--" ] && shift
    trace_run "$OUT/trace" "$@"
    pmc_runs "$OUT" "$@"
    summarise "$OUT/trace" "$OUT" "$OUT/kernel_stats.csv" "$OUT/pmc.json" "$*"
    [ "${KEEP_RAW:-0}" = "1" ] || rm -rf "$OUT/trace" "$OUT"/k[0-9]* ;;
  seq)
    [ "${1:-}" = "--" ] && shift
    rm -rf "$OUT/seq"; mkdir -p "$OUT/seq"
    timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --output-format csv -d "$OUT/seq" -o t -- "$@" > "$OUT/seq/stdout.txt" 2> "$OUT/seq/stderr.txt"
    python - "$OUT/seq" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-120:]:
    print("  %8.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:120]))
PY
    ;;
  env)
    VAR=${1:?variable}; VALS=${2:?values}; shift 2
    [ "${1:-}" = "--" ] && shift
    [ "${ABL:-0}" = "1" ] && export RGCN_HIP_LIB=$GRAFT_REPO_ROOT/torch-rgcn_amd/torch_rgcn/lib/librgcn_hip_abl.so
    for V in $VALS; do
      echo "=== $VAR=$V"
      env "$VAR=$V" timeout ${PROF_TIMEOUT:-600} "$@" 2>&1 | tail -n ${TAIL:-6}
    done | tee "$OUT/env_$VAR.txt" ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
