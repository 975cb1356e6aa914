"""Timing of the attention kernel at the bench shape with the qkv buffer evicted between launches (in situ it comes
from the qkv GEMM, not from a warm cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dark_amd import _hip
if len(sys.argv) > 1:  # an alternative build of the library
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
from dark_amd.modeling_dream_length import rope_tables
lib = _hip.load(); dev = torch.device("cuda"); st = _hip.stream_ptr(dev)
B, N, H, HD = 512, 46, 8, 96
d = H * HD
qkv = torch.randn(B, N, 3 * d, device=dev).to(torch.bfloat16)
cos, sin = rope_tables(HD, 1000.0, 128); cos, sin = cos.to(dev), sin.to(dev)
out = torch.zeros(B * N, d, device=dev, dtype=torch.bfloat16)
junk = torch.randn(128 << 20, device=dev)
def run():
    _hip.check(lib.dark_dbg_attention(_hip.DARK_BF16, _hip.ptr(qkv), _hip.ptr(cos), _hip.ptr(sin), None, None, _hip.ptr(out), B, N, H, HD, 128, st))
for mode in ("warm", "evicted"):
    tot = 0.0
    for i in range(20):
        if mode == "evicted": junk.mul_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        if i >= 4: tot += e0.elapsed_time(e1)
    print(f"{mode}: {tot / 16 * 1e3:.1f} us")
