#!/usr/bin/env python3
"""What the gate/up launch (320-row tile) costs when every LDS-DMA piece hits in the XCD's L2: probe 32 of gemm_bf16_s256
(every tile streams the SAME row panel and column tile: timing only, WRONG results) against the full kernel, with the
per-phase sums of both.  Needs the probe build.
Usage (GPU box): DARK_HIP_LIB=dark_amd/libdark_hip_probes.so python tools/probe/gemm_l2hit_probe.py"""
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
    for K in (768, 3072):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        o = torch.zeros(M, N // 2, device=dev, dtype=torch.bfloat16)
        names = {10: "full kernel, 320-row tile", 48: "every DMA an L2 hit"}
        times = {v: [] for v in names}
        for r in range(4):
            for v in names:
                _hip.check(lib.dark_dbg_set_gemm_variant(v))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    lib.dark_dbg_gemm(1, 2, _hip.ptr(A), _hip.ptr(W), None, _hip.ptr(o), M, N, K, N // 2, st)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[v].append(e0.elapsed_time(e1) / 30)
        for v, name in names.items():
            ts = sorted(times[v])
            print(f"K={K:4d} {name:28s}: {ts[len(ts) // 2] * 1e3:7.1f} us", flush=True)
        for pv, label in ((25, "full kernel"), (56, "every DMA an L2 hit")):
            _hip.check(lib.dark_dbg_set_gemm_variant(pv))
            buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
            for _ in range(3):
                buf.zero_()
                torch.cuda.synchronize()
                _hip.check(lib.dark_dbg_gemm(1, 2, _hip.ptr(A), _hip.ptr(W), buf.data_ptr(), _hip.ptr(o), M, N, K, N // 2, st))
                torch.cuda.synchronize()
            t = buf.view(256, 8, 8).double().cpu()
            for g, sl in (("group 0", slice(0, 4)), ("group 1", slice(4, 8))):
                x = t[:, sl, :].reshape(-1, 8)
                per = (x[:, :5] / x[:, 5:6]).mean(0)
                print(f"K={K:4d} {label:22s} {g}: LOAD work {per[0]:6.0f} | DMA wait {per[1]:5.0f} | barrier after LOAD {per[2]:5.0f} | "
                      f"MFMA issue {per[3]:6.0f} | barrier after MFMA {per[4]:5.0f} | sum {per.sum():6.0f}", flush=True)
    lib.dark_dbg_set_gemm_variant(0)


if __name__ == "__main__":
    main()
