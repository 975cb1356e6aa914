#!/usr/bin/env python3
"""Sustained timing of the fused attention kernel (dark_dbg_attention) at the bench shape."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dark_amd import _hip
from dark_amd.modeling_dream_length import rope_tables

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--N", type=int, default=46)
ap.add_argument("--H", type=int, default=8)
ap.add_argument("--HD", type=int, default=96)
a = ap.parse_args()
lib = _hip.load()
dev = torch.device("cuda")
s = _hip.stream_ptr(dev)
d = a.H * a.HD
qkv = (torch.randn(a.B * a.N, 3 * d, device=dev) * 0.5).bfloat16()
out = torch.empty(a.B * a.N, d, device=dev, dtype=torch.bfloat16)
cos, sin = rope_tables(a.HD, 1000.0, 128)
cos, sin = cos.to(dev), sin.to(dev)
def run():
    _hip.check(lib.dark_dbg_attention(1, _hip.ptr(qkv), _hip.ptr(cos), _hip.ptr(sin), None, None, _hip.ptr(out),
                                      a.B, a.N, a.H, a.HD, 128, s))
for _ in range(20):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / a.iters * 1e3
mb = (qkv.numel() + out.numel()) * 2 / 1e6
print(f"attention B={a.B} N={a.N} H={a.H} HD={a.HD}: {us:.1f} us  ({mb / us:.2f} TB/s of qkv+out traffic)  checksum {float(out.float().abs().mean()):.6f}")
