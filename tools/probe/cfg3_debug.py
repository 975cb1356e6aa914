"""Where does the V = 25427 loop leave the reference golden?  Replays cfg3_abd_t01 with tracing against the oracle."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, synth
from oracle import dream_oracle as O
H = os.path.join(ROOT, "tests", "golden")
meta = json.load(open(os.path.join(H, "reference_goldens_real.json")))
g = np.load(os.path.join(H, "reference_goldens_real.npz"))
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_abd_t01"
c = [c for c in meta["cases"] if c["name"] == name][0]
kwm = meta["cfg3"] if name.startswith("cfg3") else meta["real"]
cfg = DreamConfig(**kwm)
w = synth.make_weights(cfg, seed=0, rich=False)
m = DreamModel_length(cfg); m.load_state_dict(w, strict=True); m = m.to("cuda")
ids = torch.from_numpy(g[name + "_ids"])
gc = DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, max_length=46)
noises = {}
torch.manual_seed(c["seed"])
def rec(step, M, V):
    noises[step] = torch.empty(M, V).exponential_(1); return noises[step]
otr = O.SampleTrace()
ocfg = O.OracleConfig(**{k: v for k, v in kwm.items() if k in O.OracleConfig.__dataclass_fields__})
kw = c["kw"]
_, oh = O.sample_loop(w, ocfg, ids, max_length=46, steps=kw["steps"], mask_token_id=synth.MASK, alg=kw["alg"], alg_temp=0,
                      temperature=kw["temperature"], top_p=kw["top_p"], noise_fn=rec, trace=otr, tie_break="torch")
oh = torch.stack(oh)
print("oracle == golden:", np.array_equal(oh.numpy(), g[name + "_hist"]))
tr = {}
forced = []
out = m.diffusion_generate(inputs=ids.cuda(), generation_config=gc, return_dict_in_generate=True, output_history=True,
                           _noise_fn=lambda s, M, V: noises[s], _trace=tr,
                           _state_hook=lambda i, x: (forced.append(x.cpu().clone()), oh[i])[1], **kw)
for i in range(kw["steps"]):
    x0 = tr["x0"][i].cpu(); cf = tr["confidence"][i].cpu()
    dx = (x0 != otr.x0[i]).nonzero().flatten().tolist()
    dc = (cf - otr.confidence[i]).abs().max().item()
    dh = (forced[i] != oh[i]).nonzero().tolist()
    print(f"step {i}: k={tr['k'][i]}/{otr.k[i]} M={tr['M'][i]} x0 diffs {dx[:8]} max|dconf| {dc:.3e} hist diffs {dh[:6]}")
    for r in dx[:3]:
        print("   row", r, "hip", int(x0[r]), "oracle", int(otr.x0[i][r]), "conf", float(cf[r]), float(otr.confidence[i][r]))
    for b, j in dh[:3]:
        prev = oh[i-1] if i else torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
        mi = prev == synth.MASK
        off = int(mi[:b].sum())
        pos = mi[b].nonzero().flatten().tolist()
        print("   seq", b, "pos", j, "hip", int(forced[i][b, j]), "oracle", int(oh[i][b, j]), "confs", [round(float(otr.confidence[i][off + q]), 7) for q in range(len(pos))],
              "hipconfs", [round(float(cf[off + q]), 7) for q in range(len(pos))])
