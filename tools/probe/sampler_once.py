#!/usr/bin/env python3
"""Ten sampler calls at the bench's regime (16 896 rows x 15 080 random-init logits, T = 0.1, top_p = 0.9, entropy, Philox
noise) for rocprofv3 --pmc passes over sample_head_kernel / sample_kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dark_amd import _hip

lib = _hip.load()
dev = torch.device("cuda", 0)
M, V = 16896, 15080
logits = (torch.randn(M, V, generator=torch.Generator().manual_seed(0)) * 0.554).to(dev)
conf = torch.zeros(M, device=dev)
x0 = torch.zeros(M, dtype=torch.long, device=dev)
for _ in range(10):
    _hip.check(lib.dark_sample_rows_philox(_hip.ptr(logits), V, M, V, 0.1, 0.9, 0, 2, 1234, 0, 256 * 1024, _hip.ptr(conf),
                                           _hip.ptr(x0), _hip.stream_ptr(dev)))
torch.cuda.synchronize()
print("done", int(x0.min()), int(x0.max()))
