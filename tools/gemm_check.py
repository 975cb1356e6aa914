#!/usr/bin/env python3
"""Correctness + timing of the bf16 GEMM flavours through the C ABI (dark_dbg_gemm) at the denoise forward's
shapes.  Usage (GPU box): python tools/gemm_check.py [--iters 20] [--rows 23552]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dark_amd import _hip
from dark_amd.modeling_dream_length import interleave_gate_up


def pad_rows(w, mult=128):
    r = (-w.shape[0]) % mult
    return w if r == 0 else torch.cat([w, w.new_zeros((r,) + tuple(w.shape[1:]))], 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rows", type=int, default=23552)
    ap.add_argument("--check-rows", type=int, default=777)
    ap.add_argument("--only", default="", help="substring of the shape name")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--shape", default="", help="extra shape epi,N,K (e.g. 3,8192,8192)")
    args = ap.parse_args()
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    g = torch.Generator().manual_seed(0)
    shapes = [("qkv+bias", 0, 2304, 768), ("o+resid", 1, 768, 768), ("gate/up swiglu", 2, 6144, 768),
              ("down+resid", 1, 768, 3072), ("lm_head f32", 3, 15080, 768)]
    if args.shape:
        e, n, k = (int(v) for v in args.shape.split(","))
        shapes = [(f"custom epi{e}", e, n, k)]
    for name, epi, N, K in shapes:
        if args.only and args.only not in name:
            continue
        # ---- correctness on a ragged M
        M = 8 if args.no_check else args.check_rows
        A = torch.randn(M, K, generator=g).bfloat16()
        W = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
        bias = torch.randn(N, generator=g)
        Ad, Wd, bd = A.to(dev), pad_rows(W).to(dev), pad_rows(bias).to(dev)
        ref = A.double() @ W.double().t()
        if epi == 0:
            out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            ref = ref + bias.double()
            ldo = N
        elif epi == 1:
            res = torch.randn(M, N, generator=g)
            out = res.clone().to(dev)
            ref = ref + res.double()
            ldo = N
        elif epi == 2:
            Wg, Wu = W[: N // 2], W[N // 2:]
            Wd = pad_rows(interleave_gate_up(Wg, Wu, 16)).to(dev)
            gg, uu = A.double() @ Wg.double().t(), A.double() @ Wu.double().t()
            ref = torch.nn.functional.silu(gg) * uu
            out = torch.zeros(M, N // 2, device=dev, dtype=torch.bfloat16)
            ldo = N // 2
        else:
            out = torch.zeros(M, N, device=dev)
            ldo = N
        _hip.check(lib.dark_dbg_gemm(1, epi, _hip.ptr(Ad), _hip.ptr(Wd), _hip.ptr(bd) if epi == 0 else None,
                                     _hip.ptr(out), M, N, K, ldo, st))
        torch.cuda.synchronize()
        err = (out.cpu().double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        tol = 2e-2 * scale if out.dtype == torch.bfloat16 else 2e-3 * scale
        status = "ok" if (err <= tol or args.no_check) else "FAIL"
        # ---- timing at the full row count
        M = args.rows
        A = torch.randn(M, K, device=dev).bfloat16()
        Wt = (torch.randn(Wd.shape[0], K, device=dev) * 0.05).bfloat16()
        if epi in (0, 2):
            o = torch.zeros(M, ldo, device=dev, dtype=torch.bfloat16)
        else:
            o = torch.zeros(M, ldo, device=dev)
        bptr = _hip.ptr(bd) if epi == 0 else None
        for _ in range(3):
            lib.dark_dbg_gemm(1, epi, _hip.ptr(A), _hip.ptr(Wt), bptr, _hip.ptr(o), M, N, K, ldo, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            lib.dark_dbg_gemm(1, epi, _hip.ptr(A), _hip.ptr(Wt), bptr, _hip.ptr(o), M, N, K, ldo, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        print(f"{name:16s} N={N:5d} K={K:4d}  check M={args.check_rows}: max|err|={err:.3e} ({status})   "
              f"M={M}: {ms:.4f} ms  {tf:7.1f} TFLOP/s", flush=True)
        if status != "ok":
            sys.exit(1)


if __name__ == "__main__":
    main()
