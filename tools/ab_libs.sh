#!/bin/bash
# In-situ A/B of two library builds: alternating bench.py runs inside one gpurun call (same box).
# Usage (GPU box): bash tools/ab_libs.sh TAG "dark_amd/libdark_hip_old.so dark_amd/libdark_hip.so" [rounds] [extra bench args]
tag=${1:-ab}; libs=${2}; rounds=${3:-2}; extra=${4:-}
for r in $(seq 1 $rounds); do for l in $libs; do
  n=$(basename $l .so)
  DARK_HIP_LIB=$PWD/$l timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-peaked-leg $extra > gpurun_out/${tag}_${n}_$r.json 2>gpurun_out/${tag}_err.log || exit 1
  python - <<PY
import json; d=json.loads(open("gpurun_out/${tag}_${n}_$r.json").read().strip().splitlines()[-1]); print("$n", "seq/s", d["value"], "ms/step", d["ms_per_step"], "dominant launch ms", d["roofline"]["avg_launch_ms"], flush=True)
PY
done; done
