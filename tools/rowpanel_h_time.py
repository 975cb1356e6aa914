#!/usr/bin/env python3
"""Row-panel GEMM on the bf16 residual stream (dark_dbg_gemm_resid_norm_h: o_proj / down_proj + residual + RMSNorm in
one launch) against the two-launch form (dark_dbg_gemm epi 8 + dark_dbg_rmsnorm), interleaved in one process.
Usage (GPU box): python tools/rowpanel_h_time.py [--rows 23552]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dark_amd import _hip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=23552)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    M, D, EPS = a.rows, 768, 1e-6
    for K in (768, 3072):
        A = (torch.randn(M, K, device=dev) * 0.01).bfloat16()
        W = (torch.randn(D, K, device=dev) * 0.02).bfloat16()
        g = torch.ones(D, device=dev)
        h = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
        x = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        P = _hip.ptr

        def fused():
            lib.dark_dbg_gemm_resid_norm_h(P(A), P(W), P(h), P(g), P(x), M, K, EPS, st)

        def two():
            lib.dark_dbg_gemm(1, 8, P(A), P(W), None, P(h), M, D, K, D, st)
            lib.dark_dbg_rmsnorm(1, 1, P(h), P(g), P(x), None, M, M, D, EPS, st)

        def gemm_only():
            lib.dark_dbg_gemm(1, 8, P(A), P(W), None, P(h), M, D, K, D, st)

        forms = (("row-panel (GEMM + residual + norm)", fused), ("two launches", two), ("residual GEMM alone", gemm_only))
        times = {n: [] for n, _ in forms}
        for r in range(a.rounds + 1):
            for n, f in forms:
                h.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[n].append(e0.elapsed_time(e1) / a.iters * 1e3)
        for n, _ in forms:
            t = sorted(times[n])
            print(f"M={M} K={K:5d} {n:36s}: median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
