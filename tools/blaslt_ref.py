#!/usr/bin/env python3
"""Reference point for the GEMM roofline: what the vendor library (hipBLASLt through torch.matmul) sustains on the
hot path's GEMM shapes, next to this repo's kernel (dark_dbg_gemm, plain fp32-out and bf16 flavours).  Sustained:
--iters back-to-back launches per shape."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dark_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=1000)
a = ap.parse_args()
lib = _hip.load()
dev = torch.device("cuda", 0)
s = _hip.stream_ptr(dev)
shapes = [("gate_up", 23552, 6144, 768), ("qkv", 23552, 2304, 768), ("o_proj", 23552, 768, 768),
          ("down", 23552, 768, 3072), ("square8k", 8192, 8192, 8192)]
res = {}
for name, M, N, K in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    def lt():
        torch.matmul(A, W.t(), out=out)
    def ours():
        _hip.check(lib.dark_dbg_gemm(1, 0, _hip.ptr(A), _hip.ptr(W), _hip.ptr(bias), _hip.ptr(out), M, N, K, N, s))
    row = {}
    for tag, fn in (("hipblaslt", lt), ("dark", ours)):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = a.iters if M * N * K < 1e12 else max(a.iters // 10, 20)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        row[tag + "_us"] = round(ms * 1e3, 1)
        row[tag + "_tflops"] = round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1)
    res[name] = row
    print(name, row, flush=True)
print(json.dumps(res))
