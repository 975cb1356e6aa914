#!/usr/bin/env python3
"""Timing of dark_sample_rows at the bench's shapes (M masked rows x V logits, T = 0.1, top_p = 0.9, entropy)
on logits shaped like the random-init model's (N(0, sigma^2)).  Usage: sample_check.py [--rows M] [--sigma S]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dark_amd import _hip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16896)
    ap.add_argument("--vocab", type=int, default=15080)
    ap.add_argument("--sigma", type=float, default=0.55)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    M, V = a.rows, a.vocab
    logits = torch.randn(M, V, device=dev) * a.sigma
    noise = torch.empty(M, V, device=dev).exponential_()
    conf = torch.empty(M, device=dev)
    x0 = torch.empty(M, dtype=torch.long, device=dev)
    for T, top_p in ((0.1, 0.9), (0.5, 0.9), (1.0, 0.9), (0.0, 0.0)):
        def run():
            _hip.check(lib.dark_sample_rows(_hip.ptr(logits), V, M, V, T, top_p, 0, 2, _hip.ptr(noise) if T > 0 else None,
                                            V, _hip.ptr(conf), _hip.ptr(x0), st))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        gb = M * V * 4 / 1e9
        print(f"T={T} top_p={top_p} M={M} V={V}: {ms:.3f} ms  logits {gb / ms * 1e3:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
