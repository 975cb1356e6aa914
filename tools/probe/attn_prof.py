"""Per-phase timestamps of attn_kernel at the bench shape (needs the instrumented build of make_attn_prof.py); phases: staging, kbias + barrier, S = K Q^T, softmax, P V + stores."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dark_amd import _hip
from dark_amd.modeling_dream_length import rope_tables
lib = _hip.load(); dev = torch.device("cuda"); st = _hip.stream_ptr(dev)
B, N, H, HD = 512, 46, 8, 96
d = H * HD
qkv = torch.randn(B, N, 3 * d, device=dev).to(torch.bfloat16)
cos, sin = rope_tables(HD, 1000.0, 128); cos, sin = cos.to(dev), sin.to(dev)
out = torch.zeros(B * N, d, device=dev, dtype=torch.bfloat16)
junk = torch.randn(64 << 20, device=dev)
def run():
    _hip.check(lib.dark_dbg_attention(_hip.DARK_BF16, _hip.ptr(qkv), _hip.ptr(cos), _hip.ptr(sin), None, None, _hip.ptr(out), B, N, H, HD, 128, st))
for _ in range(3): run()
junk.mul_(1.0)  # evict
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
raw = ctypes.CDLL(_hip.LIB_PATH)
nb = B * H
o = np.zeros(nb * 8, dtype=np.uint64)
assert raw.dark_dbg_attn_ts(o.ctypes.data_as(ctypes.c_void_p), nb * 8) == 0
ts = o.reshape(nb, 8).astype(np.int64)
dd = np.diff(ts[:, :6], axis=1) / 100.0
print(f"kernel {e0.elapsed_time(e1)*1e3:.1f} us; per-workgroup phases (wave 0), us:")
for k, n in enumerate(["stage Q/K/V (loads, RoPE, LDS stores)", "kbias + barrier", "S = K Q^T", "softmax", "P V + output stores"]):
    print(f"  {n:38s} {dd[:, k].mean():6.2f}  p10 {np.percentile(dd[:, k],10):6.2f} p90 {np.percentile(dd[:, k],90):6.2f}")
life = (ts[:, 5] - ts[:, 0]) / 100.0
span = (ts[:, 5].max() - ts[:, 0].min()) / 100.0
print(f"  lifetime {life.mean():.2f} us, span {span:.1f} us, concurrency {life.sum()/span:.0f} workgroups")
