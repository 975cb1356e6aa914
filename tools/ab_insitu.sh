for r in 1 2; do for v in 5 0; do timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-variant $v > gpurun_out/r02q_v${v}_$r.json 2>gpurun_out/r02q_err.log || exit 1; python - <<PY
import json; d=json.loads(open("gpurun_out/r02q_v${v}_$r.json").read().strip().splitlines()[-1]); print($v, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
done; done
