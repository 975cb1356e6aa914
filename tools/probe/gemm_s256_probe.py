#!/usr/bin/env python3
"""Where an interval of gemm_bf16_s256 goes: times the gate/up launch (M = 23552, N = 6144) with parts of the kernel
removed (timing-only builds of the 256-row tile, WRONG results; dark_dbg_set_gemm_variant(16 + p)): p bit 0 = no LDS-DMA
issue in the steady state, bit 1 = no fragment reads after the first K-tile, bit 2 = no epilogue, bit 7 = no MFMAs; then
the per-phase s_memtime sums of both tile heights (correct results).  One process, interleaved rounds.
Needs the probe build of the library (make -C dark_amd/csrc probes).
Usage (GPU box): DARK_HIP_LIB=dark_amd/libdark_hip_probes.so python tools/probe/gemm_s256_probe.py [--iters 40] [--rounds 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dark_amd import _hip

NAMES = {8: "full kernel, 256-row tile", 10: "full kernel, 320-row tile", 17: "no DMA issue",
         18: "no fragment reads", 19: "no DMA, no fragment reads", 20: "no epilogue", 23: "MFMAs + barriers only",
         144: "no MFMAs (data movement only)", 148: "no MFMAs, no epilogue",
         32: "full kernel, whole-line DMA", 160: "no MFMAs, whole-line DMA", 164: "no MFMAs/epilogue, whole-line"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--phases-only", action="store_true")
    args = ap.parse_args()
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    M, N = 23552, 6144
    for K in (768, 3072):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        o = torch.zeros(M, N // 2, device=dev, dtype=torch.bfloat16)
        times = {v: [] for v in NAMES}
        for r in range(0 if args.phases_only else args.rounds + 1):
            for v in NAMES:
                _hip.check(lib.dark_dbg_set_gemm_variant(v))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    lib.dark_dbg_gemm(1, 2, _hip.ptr(A), _hip.ptr(W), None, _hip.ptr(o), M, N, K, N // 2, st)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[v].append(e0.elapsed_time(e1) / args.iters)
        ideal = 2.0 * M * N * K / 2.5e15 * 1e6
        for v, name in ({} if args.phases_only else NAMES).items():
            ts = sorted(times[v])
            med = ts[len(ts) // 2] * 1e3
            tiles = 7 if v == 10 else 9  # 2208 tiles of 256 x 256 (1776 of 320 x 256) over 256 CUs: 9 (7) rounds
            print(f"K={K:4d} {name:28s}: {med:7.1f} us  ({med / tiles / (K // 32) * 1e3:6.0f} ns per 32-deep K-tile; "
                  f"MFMA-bound at 2.4 GHz: {ideal:6.1f} us)", flush=True)
        # per-phase cycle sums (probe bit 3; results stay correct): int64 [256 workgroups][8 waves][6]
        for pv, label in ((24, "256-row tile"), (25, "320-row tile"), (104, "320-row tile, TUNE 1"), (105, "320-row tile, TUNE 5"), (106, "320-row tile, TUNE 8")):
            _hip.check(lib.dark_dbg_set_gemm_variant(pv))
            buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
            for _ in range(3):
                buf.zero_()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _hip.check(lib.dark_dbg_gemm(1, 2, _hip.ptr(A), _hip.ptr(W), buf.data_ptr(), _hip.ptr(o), M, N, K, N // 2, st))
                e1.record()
                torch.cuda.synchronize()
            t = buf.view(256, 8, 8).double().cpu()
            print(f"K={K:4d} instrumented launch ({label}): {e0.elapsed_time(e1) * 1e3:.1f} us; s_memtime ticks per 32-deep "
                  f"K-tile and wave, mean over 256 workgroups:")
            for g, sl in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
                x = t[:, sl, :].reshape(-1, 8)
                per = (x[:, :5] / x[:, 5:6]).mean(0)
                sub = (x[:, 6:8] / x[:, 5:6]).mean(0)
                print(f"   {g}: LOAD work {per[0]:6.0f} | DMA wait {per[1]:5.0f} | barrier after LOAD {per[2]:5.0f} | "
                      f"MFMA issue {per[3]:6.0f} | barrier after MFMA {per[4]:5.0f} | sum {per.sum():6.0f} "
                      f"|| inside LOAD work: fragment-read issue {sub[0]:5.0f}, DMA issue {sub[1]:5.0f}", flush=True)
    lib.dark_dbg_set_gemm_variant(0)


if __name__ == "__main__":
    main()
