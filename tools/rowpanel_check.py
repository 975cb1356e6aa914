#!/usr/bin/env python3
"""Row-panel GEMM (o_proj / down_proj + residual + RMSNorm in one launch, gemm_rowpanel.hip) against the two-launch
form (dark_dbg_gemm EPI_RESID + dark_dbg_rmsnorm) in ONE process: the fp32 residual must be bit-identical, the bf16
normalised rows may differ by the rounding of a differently associated sum of squares; both are compared with an fp64
norm of the same residual.  Then interleaved timing.  Usage (GPU box): python tools/rowpanel_check.py [--rounds 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dark_amd import _hip

EPI_RESID, BF16, EPS, D = 1, 1, 1e-6, 768


def both(lib, st, A, W, h0, g):
    M, K = A.shape
    h1, h2 = h0.clone(), h0.clone()
    x1 = torch.full((M, D), float("nan"), device=A.device, dtype=torch.bfloat16)
    x2 = torch.full_like(x1, float("nan"))
    _hip.check(lib.dark_dbg_gemm_resid_norm(_hip.ptr(A), _hip.ptr(W), _hip.ptr(h1), _hip.ptr(g), _hip.ptr(x1), M, K, EPS, st))
    _hip.check(lib.dark_dbg_gemm(BF16, EPI_RESID, _hip.ptr(A), _hip.ptr(W), None, _hip.ptr(h2), M, D, K, D, st))
    _hip.check(lib.dark_dbg_rmsnorm(BF16, 0, _hip.ptr(h2), _hip.ptr(g), _hip.ptr(x2), None, M, M, D, EPS, st))
    torch.cuda.synchronize()
    return h1, x1, h2, x2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=23552)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    gen = torch.Generator().manual_seed(0)
    for M, K in ((1, 192), (95, 192), (96, 384), (97, 768), (777, 768), (2049, 3072), (5632, 768), (args.rows, 768),
                 (args.rows, 3072)):
        A = torch.randn(M, K, generator=gen).bfloat16().to(dev)
        W = (torch.randn(D, K, generator=gen) * 0.05).bfloat16().to(dev)
        h0 = torch.randn(M, D, generator=gen).to(dev)
        g = (1.0 + 0.1 * torch.randn(D, generator=gen)).to(dev)
        h1, x1, h2, x2 = both(lib, st, A, W, h0, g)
        assert torch.equal(h1, h2), f"M={M} K={K}: residual differs, max {float((h1 - h2).abs().max()):.3e}"
        hd = h1.double()
        ref = g.double() * hd * torch.rsqrt((hd * hd).mean(-1, keepdim=True) + EPS)
        e1 = (x1.double() - ref).abs().max().item()
        e2 = (x2.double() - ref).abs().max().item()
        ndiff = int((x1 != x2).sum())
        ulp = ((x1.view(torch.int16).int() - x2.view(torch.int16).int()).abs().max().item())
        assert not torch.isnan(x1.float()).any() and ulp <= 1, f"M={M} K={K}: xn off by {ulp} bf16 ulp"
        assert e1 <= 1.02 * e2 + 1e-3, (e1, e2)
        print(f"M={M:6d} K={K:5d}: residual bit-identical; xn differs in {ndiff} of {M * D} elements (<= {ulp} ulp); "
              f"max|xn - fp64| fused {e1:.3e} two-launch {e2:.3e}", flush=True)
        # the no-norm flavour (last layer's down_proj)
        h3 = h0.clone()
        _hip.check(lib.dark_dbg_gemm_resid_norm(_hip.ptr(A), _hip.ptr(W), _hip.ptr(h3), None, None, M, K, EPS, st))
        torch.cuda.synchronize()
        assert torch.equal(h3, h2), f"M={M} K={K}: residual (no norm) differs"
    # ---- timing, interleaved
    M = args.rows
    for K in (768, 3072):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(D, K, device=dev) * 0.02).bfloat16()
        g = torch.ones(D, device=dev)
        h = torch.zeros(M, D, device=dev)
        x = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)  # evicts the Infinity Cache between forms

        def fused():
            lib.dark_dbg_gemm_resid_norm(_hip.ptr(A), _hip.ptr(W), _hip.ptr(h), _hip.ptr(g), _hip.ptr(x), M, K, EPS, st)

        def fused_nonorm():
            lib.dark_dbg_gemm_resid_norm(_hip.ptr(A), _hip.ptr(W), _hip.ptr(h), None, None, M, K, EPS, st)

        def two():
            lib.dark_dbg_gemm(BF16, EPI_RESID, _hip.ptr(A), _hip.ptr(W), None, _hip.ptr(h), M, D, K, D, st)
            lib.dark_dbg_rmsnorm(BF16, 0, _hip.ptr(h), _hip.ptr(g), _hip.ptr(x), None, M, M, D, EPS, st)

        def gemm_only():
            lib.dark_dbg_gemm(BF16, EPI_RESID, _hip.ptr(A), _hip.ptr(W), None, _hip.ptr(h), M, D, K, D, st)

        forms = (("fused", fused), ("two-launch", two), ("fused, no norm", fused_nonorm), ("resid gemm only", gemm_only))
        times = {n: [] for n, _ in forms}
        for r in range(args.rounds + 1):
            for n, f in forms:
                big.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[n].append(e0.elapsed_time(e1) / args.iters * 1e3)
        fl = 2.0 * M * D * K
        for n, _ in forms:
            t = sorted(times[n])[len(times[n]) // 2]
            print(f"K={K:5d} {n:16s}: {t:7.1f} us  ({fl / t / 1e6:6.1f} TFLOP/s)  rounds {['%.1f' % v for v in times[n]]}",
                  flush=True)


if __name__ == "__main__":
    main()
