#!/usr/bin/env python3
"""Is the sustained forward power-limited?  Runs one GEMM flavour in a loop for a few seconds while a thread samples
`rocm-smi` (socket power, sclk) twice a second, then the same for the whole denoise loop.  Usage (GPU box):
python tools/probe/power_probe.py"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dark_amd import _hip
from dark_amd.modeling_dream_length import interleave_gate_up


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                             timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        p = next((v for k, v in card.items() if "ower" in k and "W" in k), None)
        s = next((v for k, v in card.items() if k.startswith("sclk")), None)
        return p, s
    except Exception as e:  # noqa: BLE001
        return None, repr(e)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []

    def run(self):
        while not self.stop:
            self.rows.append(smi())
            time.sleep(0.4)


def main():
    lib = _hip.load()
    dev = torch.device("cuda")
    st = _hip.stream_ptr(dev)
    print("idle:", smi(), flush=True)
    M = 23552
    shapes = [("gate/up swiglu", 2, 6144, 768), ("qkv+bias", 0, 2304, 768), ("down resid bf16", 8, 768, 3072),
              ("o_proj resid bf16", 8, 768, 768)]
    for name, epi, N, K in shapes:
        A = (torch.randn(M, K, device=dev) * (0.01 if epi == 8 else 1.0)).bfloat16()
        W = (torch.randn((N + 127) // 128 * 128, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(W.shape[0], device=dev)
        ldo = N // 2 if epi == 2 else N
        o = torch.zeros(M, ldo, device=dev, dtype=torch.bfloat16)
        smp = Sampler()
        smp.start()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        first = None
        while time.time() - t0 < 4.0:
            e0.record()
            for _ in range(200):
                lib.dark_dbg_gemm(1, epi, _hip.ptr(A), _hip.ptr(W), _hip.ptr(b) if epi == 0 else None, _hip.ptr(o), M, N, K,
                                  ldo, st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            first = us if first is None else first
            n += 200
        smp.stop = True
        smp.join()
        print(f"{name:18s}: first 200 launches {first:6.1f} us each, last {us:6.1f} us "
              f"({2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s); smi samples (power, sclk): {smp.rows}", flush=True)
        time.sleep(1.0)
        # the same launch at a low duty cycle: one launch, then ~4 launch-times of a one-thread busy-wait kernel (the
        # chip idles at low power), timed by events around the GEMM alone -- what the kernel does when the power
        # management has headroom
        if epi == 2:
            for gap_cycles in (0, 400_000, 1_600_000):
                evs = []
                for _ in range(300):
                    if gap_cycles:
                        torch.cuda._sleep(gap_cycles)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    lib.dark_dbg_gemm(1, epi, _hip.ptr(A), _hip.ptr(W), None, _hip.ptr(o), M, N, K, ldo, st)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[100:])
                print(f"{name:18s}: idle gap of {gap_cycles:8d} busy-wait cycles between launches -> median {ts[len(ts) // 2]:6.1f} us, "
                      f"min {ts[0]:6.1f} us per launch", flush=True)


if __name__ == "__main__":
    main()
