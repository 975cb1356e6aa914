import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda")
M, V = 16896, 15080
x = torch.randn(M, V, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = M * V * 4 / 1e9
for name, fn in (("amax(dim=1)", lambda: x.amax(dim=1)), ("sum()", lambda: x.sum()), ("mul_(1.0) rw", lambda: x.mul_(1.0)),
                 ("clone", lambda: x.clone())):
    ms = t(fn)
    print(f"{name}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s (read side)")
