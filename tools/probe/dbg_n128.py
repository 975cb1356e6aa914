"""Step-by-step comparison of the N = 128 edge case (tests/test_hip_edges.py) with the oracle under alg='entropy': shows that the
free runs part at an fp32 near-tie of the confidences around the k-th place (x0 equal, |dconf| < 2e-6).  GPU box: python tools/probe/dbg_n128.py"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_hip_edges as T
from dark_amd import DreamConfig, DreamModel_length, synth
from oracle import dream_oracle as O
cfg = DreamConfig(**T.KW); w = synth.make_weights(cfg, seed=3, rich=True)
m = DreamModel_length(cfg); m.load_state_dict(w, strict=True); m = m.to("cuda")
ocfg = O.OracleConfig(**{k: v for k, v in T.KW.items() if k in O.OracleConfig.__dataclass_fields__})
ids = T._ids(4, 64, n_in=127, seed=9)
g = torch.Generator().manual_seed(9); noises = {}
def nf(step, M, V):
    if step not in noises: noises[step] = torch.empty(M, V).exponential_(1, generator=g)
    return noises[step]
tr = {}
kw = dict(steps=8, temperature=0.5, top_p=0.9, alg="entropy", alg_temp=0)
out = m.diffusion_generate(inputs=ids.cuda(), generation_config=T._gc(128), return_dict_in_generate=True, output_history=True, _noise_fn=nf, _trace=tr, **kw)
otr = O.SampleTrace()
seq, hist = O.sample_loop(w, ocfg, ids, max_length=128, mask_token_id=synth.MASK, noise_fn=nf, tie_break="index", trace=otr, **kw)
print("k", tr["k"], otr.k)
for i in range(8):
    same = torch.equal(out.history[i].cpu(), hist[i])
    c, oc = tr["confidence"][i].cpu(), otr.confidence[i]
    x0, ox0 = tr["x0"][i].cpu(), otr.x0[i]
    if c.shape == oc.shape:
        print(i, same, "x0 equal", torch.equal(x0, ox0), "max|dconf|", float((c - oc).abs().max()), "M", c.numel())
    else:
        print(i, same, "shapes", c.shape, oc.shape); break
    if not same:
        d = (out.history[i].cpu() != hist[i]).nonzero()
        print("differs at", d.tolist())
        prev = hist[i-1]
        mi = prev == synth.MASK
        off = torch.cumsum(mi.sum(1), 0) - mi.sum(1)
        for b in set(d[:, 0].tolist()):
            pos = mi[b].nonzero().flatten()
            cb, ocb = c[off[b]:off[b]+pos.numel()], oc[off[b]:off[b]+pos.numel()]
            k = tr["k"][i]
            print("row", b, "k", k, "hip top", torch.sort(cb, descending=True).values[:k+2].tolist(), "oracle top", torch.sort(ocb, descending=True).values[:k+2].tolist())
        break
