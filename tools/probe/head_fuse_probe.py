#!/usr/bin/env python3
"""Stage check of the fused LM head: the EPI_HEAD slice records against the logits the plain head GEMM writes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dark_amd import _hip
lib = _hip.load()
dev = torch.device("cuda"); st = _hip.stream_ptr(dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16896
V, K, T = 15080, 768, 0.1
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).bfloat16().to(dev)
W = (torch.randn(V, K, generator=g) * 0.05).bfloat16()
Wp = torch.cat([W, W.new_zeros(((-V) % 128, K))], 0).to(dev)
P = 4 * ((V + 255) // 256)
logits = torch.empty(M, V, device=dev)
rec = torch.full((M, P, 4), -7.0, device=dev)
_hip.check(lib.dark_dbg_gemm(1, 3, _hip.ptr(A), _hip.ptr(Wp), None, _hip.ptr(logits), M, V, K, V, st))
_hip.check(lib.dark_dbg_gemm_head(_hip.ptr(A), _hip.ptr(Wp), _hip.ptr(rec), M, V, K, T, None, st))
torch.cuda.synchronize()
Mc = min(M, 2048)
lg = torch.nn.functional.pad(logits[:Mc], (0, P * 64 - V), value=float("-inf")).view(Mc, P, 64)
smax, col = lg.max(-1)
# lowest column among equal maxima
eq = lg == smax.unsqueeze(-1)
col_low = torch.where(eq, torch.arange(64, device=dev).expand_as(lg), torch.full_like(lg, 64, dtype=torch.long)).min(-1).values
r = rec[:Mc]
c2 = np.float32(1.4426950408889634) / np.float32(T)
rng = np.float32(6.0) * np.float32(T) * np.float32(1.001)
thr = smax - (float(rng) + 2e-6 * smax.abs())
nn_ref = (lg > thr.unsqueeze(-1)).sum(-1)
ts_ref = torch.exp2((lg - smax.unsqueeze(-1)).double() * float(c2)).sum(-1) - 1.0
bits = r[..., 2].contiguous().view(torch.int32)
nn_got, col_got = bits & 0xFFFF, bits >> 16
ok_max = torch.equal(r[..., 0], smax)
ok_col = torch.equal(col_got.long(), col_low)
nn_bad = int((nn_got.long() != nn_ref).sum())
ts_err = ((r[..., 1].double() - ts_ref).abs() / (ts_ref.abs() + 1e-3)).max().item()
print(f"records: max exact {ok_max}, column exact {ok_col}, near-count mismatches {nn_bad} (threshold roundings only), "
      f"tsum max rel err {ts_err:.2e}; hot slices with >= 2 near columns: "
      f"{float(((nn_got >= 2) & (r[..., 0] > smax.max(-1, keepdim=True).values - float(rng))).any(-1).float().mean()):.4f} of rows")
assert ok_max and ok_col and ts_err < 1e-3 and nn_bad <= 4
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_f32 = timeit(lambda: lib.dark_dbg_gemm(1, 3, _hip.ptr(A), _hip.ptr(Wp), None, _hip.ptr(logits), M, V, K, V, st))
t_head = timeit(lambda: lib.dark_dbg_gemm_head(_hip.ptr(A), _hip.ptr(Wp), _hip.ptr(rec), M, V, K, T, None, st))
print(f"M={M}: head GEMM fp32 logits {t_f32:.1f} us, slice records {t_head:.1f} us")
