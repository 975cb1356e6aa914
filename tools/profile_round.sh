#!/bin/bash
# rocprofv3 evidence for one round: kernel trace + three PMC passes over the headline bench, summaries into
# gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/).  Usage (GPU box): bash tools/profile_round.sh r02
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
BENCH1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-peaked-leg"
run() {  # name, then the rocprofv3 arguments
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" -d "$OUT/$name" -- ${CMD} > "$OUT/$name.log" 2>&1 || { echo "$name FAILED"; tail -5 "$OUT/$name.log"; return 1; }
  echo "$name ok"
}
CMD=$BENCH  run trace --kernel-trace --stats || exit 1
DB=$(find "$OUT/trace" -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py "$DB" "$OUT/${TAG}_bench_kernel_stats.csv" > /dev/null && head -12 "$OUT/${TAG}_bench_kernel_stats.csv"
grep -h '^{' "$OUT/trace.log" | tail -1 > "$OUT/${TAG}_bench_under_trace.json"
{
  echo "# $TAG: rocprofv3 --pmc, three separate passes over: $BENCH1"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    name=pmc_$(echo $pass | cut -d' ' -f1)
    CMD=$BENCH1 run $name --pmc $pass || exit 1
    DBP=$(find "$OUT/$name" -name "*.db" | head -1)
    echo "# ---- pass $pass"
    python3 $R/tools/pmc_query.py "$DBP" --min-grid 100000 --full-size
  done
} > "$OUT/${TAG}_pmc.txt" 2> "$OUT/pmc.err"
tail -3 "$OUT/pmc.err"
grep -c . "$OUT/${TAG}_pmc.txt"
# the sidecar bench.py reads for roofline.traffic: per-launch memory-side bytes of the dominant kernel from the passes above
python3 - "$OUT/${TAG}_pmc.txt" "$OUT/roofline_traffic.json" "$TAG" <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
vals = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    sec = txt.split("# ---- pass " + counter)[1].split("# ---- pass")[0]
    m = re.search(r"gemm_bf16_s256<2, 0, 10(?:, \d+)?>.*?\n\s+dispatches.*?\n\s+" + counter + r"\s+(\d+)", sec, re.S)
    vals[counter] = int(m.group(1))
kib = 2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]   # FETCH_SIZE reads half the bytes of wide streaming reads on gfx950
json.dump({"kernel": "gemm_bf16_s256<2, 0, 10, 10>", "dtype": "bf16", "batch": 1024, "fetch_size_kib": vals["FETCH_SIZE"],
           "write_size_kib": vals["WRITE_SIZE"], "traffic_bytes_per_launch": kib * 1024,
           "source": "profiles/%s_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; 2 x FETCH + WRITE)" % sys.argv[3]},
          open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
find "$OUT" -name "*.db" -delete   # databases are large; the summaries stay
