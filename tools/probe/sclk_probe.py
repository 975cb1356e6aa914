#!/usr/bin/env python3
"""Sustained shader clock under the gate/up GEMM for a list of dark_dbg_set_gemm_variant values (probe build: 16 + p are
the timing-only probes of gemm_bf16_s256, e.g. 144 = data movement only).  rocm-smi sampled while the launch repeats.
Usage (GPU box): DARK_HIP_LIB=dark_amd/libdark_hip_probes.so python tools/probe/sclk_probe.py 8,144,20"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dark_amd import _hip
from power_probe import Sampler, smi


def main():
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8,144").split(",")]
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    M, N, K = 23552, 6144, 768
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    o = torch.zeros(M, N // 2, device=dev, dtype=torch.bfloat16)
    print("idle:", smi(), flush=True)
    for v in variants:
        _hip.check(lib.dark_dbg_set_gemm_variant(v))
        smp = Sampler()
        smp.start()
        t0 = time.time()
        while time.time() - t0 < 4.0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                _hip.check(lib.dark_dbg_gemm(1, 2, _hip.ptr(A), _hip.ptr(W), None, _hip.ptr(o), M, N, K, N // 2, st))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
        smp.stop = True
        smp.join()
        clk = [s for _, s in smp.rows[2:]]
        print(f"variant {v:5d}: {us:6.1f} us per launch; sclk samples {clk}", flush=True)
        time.sleep(1.0)
    _hip.check(lib.dark_dbg_set_gemm_variant(0))


if __name__ == "__main__":
    main()
