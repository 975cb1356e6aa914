"""Timing of the attention kernel (default: the bench shape) with the qkv buffer evicted between launches (in situ it
comes from the qkv GEMM, not from a warm cache).  Usage (GPU box): python tools/attn_time.py [B N H HD] [--variants 0,12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dark_amd import _hip
from dark_amd.modeling_dream_length import rope_tables
args = [a for a in sys.argv[1:] if not a.startswith("--")]
variants = [0]
for a in sys.argv[1:]:
    if a.startswith("--variants="):
        variants = [int(v) for v in a.split("=")[1].split(",")]
lib = _hip.load(); dev = torch.device("cuda"); st = _hip.stream_ptr(dev)
B, N, H, HD = [int(v) for v in args[:4]] if len(args) >= 4 else (512, 46, 8, 96)
d = H * HD
qkv = torch.randn(B, N, 3 * d, device=dev).to(torch.bfloat16)
cos, sin = rope_tables(HD, 1000.0, 128); cos, sin = cos.to(dev), sin.to(dev)
out = torch.zeros(B * N, d, device=dev, dtype=torch.bfloat16)
junk = torch.randn(128 << 20, device=dev)
def run():
    _hip.check(lib.dark_dbg_attention(_hip.DARK_BF16, _hip.ptr(qkv), _hip.ptr(cos), _hip.ptr(sin), None, None, _hip.ptr(out), B, N, H, HD, 128, st))
ref = None
for rnd in range(2):
    for v in variants:
        _hip.check(lib.dark_dbg_set_gemm_variant(v))
        for mode in ("warm", "evicted"):
            tot = 0.0
            for i in range(20):
                if mode == "evicted": junk.mul_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                if i >= 4: tot += e0.elapsed_time(e1)
            byts = B * N * d * 2 * 4
            print(f"B={B} N={N} H={H} HD={HD} variant {v} {mode}: {tot / 16 * 1e3:.1f} us  ({byts / (tot / 16 * 1e-3) / 1e12:.2f} TB/s of qkv + output)", flush=True)
        if ref is None:
            ref = out.clone()
        else:
            print(f"   variant {v} bit-identical to variant {variants[0]}: {torch.equal(ref, out)}", flush=True)
_hip.check(lib.dark_dbg_set_gemm_variant(0))
