#!/bin/bash
# In-situ A/B of GEMM variants: alternating bench.py runs inside one gpurun call (same box).
# Usage (GPU box): bash tools/ab_insitu.sh TAG "0 1008" [rounds]
tag=${1:-ab}; vars=${2:-"0 5"}; rounds=${3:-2}
for r in $(seq 1 $rounds); do for v in $vars; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-peaked-leg --gemm-variant $v > gpurun_out/${tag}_v${v}_$r.json 2>gpurun_out/${tag}_err.log || exit 1
  python - <<PY
import json; d=json.loads(open("gpurun_out/${tag}_v${v}_$r.json").read().strip().splitlines()[-1]); print("variant", $v, "seq/s", d["value"], "ms/step", d["ms_per_step"], "dominant launch ms", d["roofline"]["avg_launch_ms"], flush=True)
PY
done; done
