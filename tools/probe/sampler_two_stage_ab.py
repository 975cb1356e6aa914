#!/usr/bin/env python3
"""A/B of the two-stage sampler (sample_head_kernel + sample_kernel on its list) against the single-stage kernel:
same logits, several regimes; asserts bit-identical conf / x0 and prints the times (dark_dbg_set_sampler_stages)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dark_amd import _hip

lib = _hip.load()
dev = torch.device("cuda", 0)
s = _hip.stream_ptr(dev)


NOISE = {}


def run(logits, V, T, top_p, mode, philox, two_stage, reps=5):
    M = logits.shape[0]
    if "--mem-noise" in sys.argv:   # the materialised fill instead of the lazy Philox draw
        if M not in NOISE:
            NOISE[M] = torch.empty(M, logits.stride(0), device=dev).exponential_(generator=torch.Generator(device=dev).manual_seed(7))
        return run_mem(logits, V, T, top_p, mode, NOISE[M], two_stage, reps)
    conf = torch.zeros(M, device=dev)
    x0 = torch.zeros(M, dtype=torch.long, device=dev)
    raw = lib.dark_sample_rows_philox
    _hip.check(lib.dark_dbg_set_sampler_stages(0 if two_stage else 1))

    def go():
        _hip.check(raw(_hip.ptr(logits), logits.stride(0), M, V, T, top_p, 0, mode, philox[0], philox[1], philox[2],
                       _hip.ptr(conf), _hip.ptr(x0), s))
    go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    return conf.clone(), x0.clone(), (time.perf_counter() - t0) / reps * 1e3


def run_mem(logits, V, T, top_p, mode, noise, two_stage, reps):
    M = logits.shape[0]
    conf = torch.zeros(M, device=dev)
    x0 = torch.zeros(M, dtype=torch.long, device=dev)
    _hip.check(lib.dark_dbg_set_sampler_stages(0 if two_stage else 1))

    def go():
        _hip.check(lib.dark_sample_rows(_hip.ptr(logits), logits.stride(0), M, V, T, top_p, 0, mode, _hip.ptr(noise),
                                        noise.stride(0), _hip.ptr(conf), _hip.ptr(x0), s))
    go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    return conf.clone(), x0.clone(), (time.perf_counter() - t0) / reps * 1e3


def main():
    g = torch.Generator(device="cpu").manual_seed(0)
    philox = (1234, 0, 256 * 1024)
    regimes = (("bench regime (random-init, T=0.1)", 16896, 15080, 0.554, 0.1),
               ("peaked (trained-like)", 16896, 15080, 2.0, 0.1),
               ("flat (T=0.5)", 16896, 15080, 0.554, 0.5),
               ("mixed", 16896, 15080, 0.554, 0.1),
               ("greedy T=0", 16896, 15080, 5.54, 0.0),
               ("dbpedia vocab", 8192, 25427, 0.554, 0.1))
    for name, M, V, sigma, T in regimes[:int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else len(regimes)]:
        ld = (V + 3) & ~3
        buf = torch.zeros(M, ld)
        buf[:, :V] = torch.randn(M, V, generator=g) * sigma
        if name == "mixed":
            buf[::3, :V] *= 0.2
        buf[5, 100:140] = buf[5, :V].max() + 0.3     # runs of equal logits across the nucleus boundary
        buf[6, 200:500] = buf[6, :V].max() + 0.3
        logits = buf.to(dev)
        for mode in (2, 0, 1):
            for _ in range(2):   # twice: the hint adapts
                c1, x1, t1 = run(logits, V, T, 0.9, mode, philox, True)
            c0, x0, t0 = run(logits, V, T, 0.9, mode, philox, False)
            same = bool(torch.equal(c0.view(torch.int32), c1.view(torch.int32)) and torch.equal(x0, x1))
            print(f"{name:36s} mode {mode}: single-stage {t0:7.3f} ms   two-stage {t1:7.3f} ms   bit-identical {same}", flush=True)
            assert same or "--no-assert" in sys.argv


if __name__ == "__main__":
    main()
