"""Per-phase clock profile of sample_kernel (needs the profiling build of the library: see DESIGN.md section 5)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dark_amd import _hip
lib = _hip.load()
dev = torch.device("cuda")
st = _hip.stream_ptr(dev)
M, V = 16896, 15080
logits = torch.randn(M, V, device=dev) * 0.55
noise = torch.empty(M, V, device=dev).exponential_()
conf = torch.empty(M, device=dev); x0 = torch.empty(M, dtype=torch.long, device=dev)
raw = ctypes.CDLL(_hip.LIB_PATH)
names = ["load", "block_max", "nucleus_crossing", "filter/list", "softmax Z", "p/entropy", "draw", "argmax+write"]
xnames = ["zero / -", "mass (+hist) pass", "block sum", "- / hist pass", "scan", "bin gather", "resolve"]
for T, top_p, lazy, stages in ((0.1, 0.9, True, 1), (0.5, 0.9, True, 1), (0.5, 0.9, False, 1), (0.5, 0.9, False, 0), (1.0, 0.9, True, 1),
                               (0.0, 0.0, False, 1)):
    # stages = 1: the single-stage kernel; 0: two stages (flat rows reach the same kernel with the first stage's hint)
    _hip.check(lib.dark_dbg_set_sampler_stages(stages))
    def run():
        if lazy:
            _hip.check(lib.dark_sample_rows_philox(_hip.ptr(logits), V, M, V, T, top_p, 0, 2, 1234, 0, 256 * 2048, _hip.ptr(conf),
                                                   _hip.ptr(x0), st))
        else:
            _hip.check(lib.dark_sample_rows(_hip.ptr(logits), V, M, V, T, top_p, 0, 2, _hip.ptr(noise) if T > 0 else None,
                                            V, _hip.ptr(conf), _hip.ptr(x0), st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out = np.zeros(M * 10, dtype=np.uint64)
    assert raw.dark_dbg_sample_ts(out.ctypes.data_as(ctypes.c_void_p), M * 10) == 0
    ts = out.reshape(M, 10).astype(np.int64)
    if top_p == 0.0:
        ts[:, 3] = ts[:, 2]
    d = np.diff(ts[:, :9], axis=1) / 100.0
    print(f"T={T} top_p={top_p} lazy={lazy} stages={'two' if stages == 0 else 'one'}: {ms:.3f} ms/launch; mean per-block phase time (us):")
    for k, n in enumerate(names):
        print(f"   {n:18s} {d[:, k].mean():7.2f}  (p10 {np.percentile(d[:, k], 10):6.2f}  p90 {np.percentile(d[:, k], 90):6.2f})")
    life = (ts[:, 8] - ts[:, 0]) / 100.0
    span = (ts[:, 8].max() - ts[:, 0].min()) / 100.0
    print(f"   block lifetime mean {life.mean():.2f} us; kernel span {span:.1f} us; mean concurrency {life.sum() / span:.0f} blocks")
    if T == 0.5 and not lazy:
        outx = np.zeros(M * 8, dtype=np.uint64)
        assert raw.dark_dbg_sample_tx(outx.ctypes.data_as(ctypes.c_void_p), M * 8) == 0
        dx = np.diff(outx.reshape(M, 8).astype(np.int64)[:, :7], axis=1) / 100.0
        if stages == 0:   # flat path: slots 0..3 are entry, zeroed, fused pass, block sum
            cols = [("zero hist", 0), ("mass + hist pass", 1), ("block sum", 2), ("scan", 3), ("bin gather", 4), ("resolve", 5)]
        else:
            cols = [("mass pass", 0), ("block sum", 1), ("hist pass", 2), ("scan", 3), ("bin gather", 4), ("resolve", 5)]
        print("   inside nucleus_crossing:")
        for n, k in cols:
            print(f"      {n:18s} {dx[:, k].mean():7.2f}")
