"""A/B of two builds of the sampler: bit-equality of (conf, x0) and timing.  usage: sample_ab.py old.so new.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dark_amd import _hip
dev = torch.device("cuda")
st = _hip.stream_ptr(dev)
libs = []
for p in sys.argv[1:3]:
    l = C.CDLL(os.path.abspath(p))
    f = l.dark_sample_rows_philox
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                  C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    libs.append(f)
M, V = 16896, 15080
g = torch.Generator(device=dev).manual_seed(0)
base = torch.randn(M, V, device=dev, generator=g)
bad = 0
for sigma in (0.55, 0.2, 1.5):
    logits = base * sigma
    logits[::7, 5] = logits[::7, 9]          # exact ties
    logits[::11, :64] = logits[::11, :1]     # a run of equal logits (may straddle the nucleus boundary)
    for T, top_p, mode in ((0.1, 0.9, 2), (0.1, 0.9, 0), (0.1, 0.9, 1), (0.3, 0.5, 2), (1.0, 0.9, 2), (0.0, 0.9, 2), (0.1, 0.999, 2)):
        res, tm = [], []
        for f in libs:
            conf = torch.full((M,), -7.0, device=dev); x0 = torch.full((M,), -7, dtype=torch.long, device=dev)
            def run():
                rc = f(logits.data_ptr(), V, M, V, T, top_p, 0, mode, 4321, 8, 256 * 2048, conf.data_ptr(), x0.data_ptr(), st)
                assert rc == 0
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            tm.append(e0.elapsed_time(e1) / 10)
            res.append((conf.clone(), x0.clone()))
        same = torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32)) and torch.equal(res[0][1], res[1][1])
        bad += not same
        print(f"sigma={sigma} T={T} top_p={top_p} conf_mode={mode}: old {tm[0]:.3f} ms  new {tm[1]:.3f} ms  bit-equal={same}", flush=True)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
