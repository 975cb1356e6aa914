#!/usr/bin/env python3
"""Timing probe: the one-workgroup-per-row sampler against the streaming shard kernels (S = 1) on [M, V] logits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dark_amd import _hip
lib = _hip.load()
dev = torch.device("cuda"); st = _hip.stream_ptr(dev)
M, V, T, top_p = 16896, 15080, 0.1, 0.9
torch.manual_seed(0)
logits = torch.randn(M, V, device=dev) * 2.0
conf = torch.empty(M, device=dev); x0 = torch.empty(M, dtype=torch.int64, device=dev)
conf2 = torch.empty(M, device=dev); x02 = torch.empty(M, dtype=torch.int64, device=dev)
mx = torch.empty(M, device=dev); parts = torch.zeros(M, _hip.SHARD_PART_BYTES, dtype=torch.uint8, device=dev)
flags = torch.zeros(M, dtype=torch.uint8, device=dev)
seed, off, nth = 1234, 0, 1 << 20
def full():
    _hip.check(lib.dark_sample_rows_philox(_hip.ptr(logits), V, M, V, T, top_p, 0, 2, seed, off, nth, _hip.ptr(conf), _hip.ptr(x0), st))
def p_max():
    _hip.check(lib.dark_shard_row_max(_hip.ptr(logits), V, M, V, T, _hip.ptr(mx), st))
def p_parts():
    _hip.check(lib.dark_shard_partials(_hip.ptr(logits), V, M, V, 0, T, _hip.ptr(mx), _hip.ptr(parts), st))
def p_merge():
    _hip.check(lib.dark_sample_from_shards(_hip.ptr(parts), M, 1, M, V, T, top_p, 0, 2, _hip.ptr(mx), None, 0, seed, off, nth,
                                           _hip.ptr(conf2), _hip.ptr(x02), _hip.ptr(flags), st))
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, f in (("full sampler", full), ("row max", p_max), ("partials", p_parts), ("merge", p_merge)):
    print(f"{name:14s} {timeit(f):8.1f} us", flush=True)
nf = int(flags.sum())
ok = (flags == 0)
print("flagged rows", nf, "of", M, "; unflagged rows equal:", bool(torch.equal(conf[ok], conf2[ok]) and torch.equal(x0[ok], x02[ok])))
