import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dark_amd import _hip
if len(sys.argv) > 1:  # an alternative build of the library (e.g. one with an older attn.hip linked in)
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
from dark_amd.modeling_dream_length import rope_tables
lib = _hip.load(); dev = torch.device("cuda"); st = _hip.stream_ptr(dev)
B, N, H, HD = 512, 46, 8, 96
d = H * HD
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, N, 3 * d, device=dev, generator=g).to(torch.bfloat16)
cos, sin = rope_tables(HD, 1000.0, 128); cos, sin = cos.to(dev), sin.to(dev)
outs = []
for i in range(6):
    out = torch.full((B * N, d), 7.0, device=dev, dtype=torch.bfloat16)
    _hip.check(lib.dark_dbg_attention(_hip.DARK_BF16, _hip.ptr(qkv), _hip.ptr(cos), _hip.ptr(sin), None, None, _hip.ptr(out), B, N, H, HD, 128, st))
    torch.cuda.synchronize()
    outs.append(out)
for i in range(1, 6):
    df = (outs[0].float() - outs[i].float()).abs()
    print(i, "identical" if torch.equal(outs[0].view(torch.int16), outs[i].view(torch.int16)) else f"DIFF max {float(df.max()):.4f} n {int((df>0).sum())} rows {sorted(set((df>0).nonzero()[:,0].tolist()))[:8]} cols {sorted(set((df>0).nonzero()[:,1].tolist()))[:12]}")
