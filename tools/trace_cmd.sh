#!/bin/bash
# rocprofv3 kernel trace of one python command, summarised to CSV.  Usage (GPU box): bash tools/trace_cmd.sh TAG script.py [args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --stats -d "$OUT/t" -- python3 "$R/$1" "${@:2}" > "$OUT/log.txt" 2>&1 || { echo FAILED; tail -5 "$OUT/log.txt"; exit 1; }
DB=$(find "$OUT/t" -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py "$DB" "$R/gpurun_out/${TAG}_kernel_stats.csv" > /dev/null
head -14 "$R/gpurun_out/${TAG}_kernel_stats.csv"
tail -2 "$OUT/log.txt"
find "$OUT" -name "*.db" -delete
