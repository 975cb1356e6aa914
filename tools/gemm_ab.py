#!/usr/bin/env python3
"""A/B of the bf16 GEMM kernels in ONE process (interleaved rounds, same device, random operands): variant 1 = the
128-row kernels of gemm.hip, 2 = the persistent 256 x 256 kernel of gemm256.hip, 4 = its staggered form with the
automatic tile height, 8 / 9 / 10 = the staggered form with 256 / 288 / 320-row tiles (a variant serves the shapes it supports and
falls back to variant 1 elsewhere).  Checks variant 2 against an
fp64 reference on ragged row counts first.  Usage (GPU box): python tools/gemm_ab.py [--rows 23552] [--rounds 5]"""
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
    ap.add_argument("--rows", type=int, default=23552)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--variants", default="1,2")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    g = torch.Generator().manual_seed(0)
    shapes = [("gate/up swiglu", 2, 6144, 768), ("qkv+bias", 0, 2304, 768), ("lm_head f32", 3, 15080, 768),
              ("gate/up K3072", 2, 6144, 3072), ("o_proj resid", 1, 768, 768), ("down resid", 1, 768, 3072),
              ("o_proj resid bf16", 8, 768, 768), ("down resid bf16", 8, 768, 3072),   # (bf16 residual stream)
              ("n768 k768 bias", 0, 768, 768), ("n768 k3072 bias", 0, 768, 3072),      # (one-round probes for s256)
              ("probe bias N6144", 0, 6144, 768), ("probe swiglu N2304", 2, 2304, 768)]  # (flavour vs round count)
    for name, epi, N, K in shapes:
        if args.only and args.only not in name:
            continue
        # ---- correctness of every variant on ragged row counts
        for M in (777, 300, 2049):
            A = torch.randn(M, K, generator=g).bfloat16()
            W = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
            bias = torch.randn(N, generator=g)
            Ad, Wd, bd = A.to(dev), pad_rows(W).to(dev), pad_rows(bias).to(dev)
            ref = A.double() @ W.double().t()
            if epi == 0:
                ref, ldo, odt = ref + bias.double(), N, torch.bfloat16
            elif epi == 2:
                Wg, Wu = W[: N // 2], W[N // 2:]
                Wd = pad_rows(interleave_gate_up(Wg, Wu, 16)).to(dev)
                ref = torch.nn.functional.silu(A.double() @ Wg.double().t()) * (A.double() @ Wu.double().t())
                ldo, odt = N // 2, torch.bfloat16
            else:
                ldo, odt = N, torch.float32
            base = torch.randn(M, ldo, generator=g) if epi in (1, 8) else None   # the residual stream the GEMM adds to
            if epi == 8:
                base, odt = base.bfloat16(), torch.bfloat16
            if base is not None:
                ref = ref + base.double()
            outs = []
            for v in variants:
                _hip.check(lib.dark_dbg_set_gemm_variant(v))
                out = base.to(dev).clone() if base is not None else torch.full((M, ldo), float("nan"), device=dev, dtype=odt)
                _hip.check(lib.dark_dbg_gemm(1, epi, _hip.ptr(Ad), _hip.ptr(Wd), _hip.ptr(bd) if epi == 0 else None,
                                             _hip.ptr(out), M, N, K, ldo, st))
                torch.cuda.synchronize()
                err = (out.cpu().double() - ref).abs().max().item()
                tol = (2e-2 if odt == torch.bfloat16 else 2e-3) * ref.abs().max().item()
                assert err <= tol or v == 3,  f"{name} variant {v} M={M}: max|err| {err:.3e} > {tol:.3e}"
                outs.append(out)
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            print(f"{name:16s} M={M:5d}: all variants within tolerance; bit-identical to each other: {same}", flush=True)
        # ---- timing
        M = args.rows
        A = torch.randn(M, K, device=dev).bfloat16()
        Wt = (torch.randn(pad_rows(torch.empty(N, 1)).shape[0], K, device=dev) * 0.05).bfloat16()
        bd = torch.randn(Wt.shape[0], device=dev)
        ldo = N // 2 if epi == 2 else N
        o = torch.zeros(M, ldo, device=dev, dtype=torch.float32 if epi in (1, 3) else torch.bfloat16)
        if epi == 8:
            A = A * 0.01   # (the in-place sums of thousands of timing launches must stay finite in bf16)
        bptr = _hip.ptr(bd) if epi == 0 else None
        times = {v: [] for v in variants}
        for r in range(args.rounds + 1):
            for v in variants:
                _hip.check(lib.dark_dbg_set_gemm_variant(v))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    lib.dark_dbg_gemm(1, epi, _hip.ptr(A), _hip.ptr(Wt), bptr, _hip.ptr(o), M, N, K, ldo, st)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[v].append(e0.elapsed_time(e1) / args.iters)
        # ---- the variants' outputs at the timed size, bit for bit against the first one
        base_o = (torch.randn(M, ldo, device=dev) * 0.1).to(o.dtype)
        first = None
        for v in variants:
            _hip.check(lib.dark_dbg_set_gemm_variant(v))
            oo = base_o.clone()
            _hip.check(lib.dark_dbg_gemm(1, epi, _hip.ptr(A), _hip.ptr(Wt), bptr, _hip.ptr(oo), M, N, K, ldo, st))
            torch.cuda.synchronize()
            if first is None:
                first = oo
            elif v != 3 and not (16 <= v < 1000):
                print(f"{name:16s} M={M}: variant {v} bit-identical to variant {variants[0]}: {torch.equal(first, oo)}", flush=True)
        for v in variants:
            ts = sorted(times[v])
            med = ts[len(ts) // 2]
            print(f"{name:16s} N={N:5d} K={K:4d} M={M} variant {v}: median {med*1e3:7.1f} us  min {ts[0]*1e3:7.1f} us  "
                  f"{2.0*M*N*K/(med*1e-3)/1e12:7.1f} TFLOP/s", flush=True)
    lib.dark_dbg_set_gemm_variant(0)


if __name__ == "__main__":
    main()
