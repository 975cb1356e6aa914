#!/usr/bin/env python3
"""Does the operand row pitch matter (L2 channel / bank camping)?  Times the gate/up flavour of the persistent GEMM
(M = 23552, N = 6144) for K = 768 ... 1152 in steps of 64 (row pitch 1536 ... 2304 bytes: 12 ... 18 cache lines) and prints
the time per 32-deep K-tile; a pitch that camps on a subset of the L2 channels would show as a slower K-tile.
Usage (GPU box): python tools/probe/gemm_pitch_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dark_amd import _hip


def main():
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    M, N = 23552, 6144
    Ks = [768, 832, 896, 960, 1024, 1088, 1152, 1536, 2048, 3072, 3136]
    bufs = {}
    for K in Ks:
        bufs[K] = (torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) * 0.05).bfloat16())
    o = torch.zeros(M, N // 2, device=dev, dtype=torch.bfloat16)
    times = {K: [] for K in Ks}
    for r in range(5):
        for K in Ks:
            A, W = bufs[K]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.dark_dbg_gemm(1, 2, _hip.ptr(A), _hip.ptr(W), None, _hip.ptr(o), M, N, K, N // 2, st)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[K].append(e0.elapsed_time(e1) / 20)
    for K in Ks:
        t = sorted(times[K])[len(times[K]) // 2] * 1e3
        print(f"K={K:5d} pitch {2 * K:5d} B = {2 * K / 128:5.1f} lines: {t:7.1f} us  {t / 7 / (K // 32) * 1e3:6.0f} ns per K-tile "
              f"(7 rounds of 320-row tiles)  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
