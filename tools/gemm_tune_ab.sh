#!/bin/bash
# A/B of gemm_bf16_s256's TUNE variants (dark_dbg_set_gemm_variant(1000 + TUNE)) on the three shapes the kernel serves.
# Needs the probe build (make -C dark_amd/csrc probes).
# Usage (GPU box): bash tools/gemm_tune_ab.sh OUT.txt "0,1001,1004,1005"
export DARK_HIP_LIB=${DARK_HIP_LIB:-dark_amd/libdark_hip_probes.so}
out=${1:-gpurun_out/tune_ab.txt}; vars=${2:-0,1001,1004,1005}
: > "$out"
for shape in "gate/up swiglu" "qkv" "down resid bf16"; do
  timeout -k 10 300 python -u tools/gemm_ab.py --variants "$vars" --rounds 4 --iters 30 --only "$shape" >> "$out" 2>&1 || exit 1
done
grep -E "variant" "$out" | grep -v "M=  777\|M=  300\|M= 2049"
