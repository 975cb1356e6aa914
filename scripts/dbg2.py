import sys, json, torch, numpy as np
sys.path.insert(0, '.')
from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, synth
from oracle import dream_oracle as O
meta = json.load(open('tests/golden/reference_goldens.json')); G = np.load('tests/golden/reference_goldens.npz')
name = sys.argv[1] if len(sys.argv) > 1 else "abd_t05_s32"
case = [c for c in meta["cases"]["loops"] if c["name"] == name][0]; kw = dict(case["kw"])
cfg = DreamConfig(**meta["mini"]); m = DreamModel_length(cfg); w = synth.make_weights(cfg, 0, True); m.load_state_dict(w); m = m.to('cuda')
ocfg = O.OracleConfig(**{k: v for k, v in meta["mini"].items() if k in O.OracleConfig.__dataclass_fields__})
ids = torch.from_numpy(G["loop_abd_ids" if name.startswith("abd") else "loop_ded_ids"])
noises = {}
torch.manual_seed(case["seed"])
def noise_fn(step, M, V):
    noises[step] = torch.empty(M, V).exponential_(1); return noises[step]
gc = DreamGenerationConfig(mask_token_id=6, pad_token_id=2, max_length=46)
trace = {}
out = m.diffusion_generate(inputs=ids.cuda(), generation_config=gc, return_dict_in_generate=True, output_history=True, _noise_fn=noise_fn, _trace=trace, **kw)
otr = O.SampleTrace()
seq, hist = O.sample_loop(w, ocfg, ids, max_length=46, mask_token_id=6, steps=kw["steps"], alg=kw["alg"], alg_temp=kw.get("alg_temp"), temperature=kw["temperature"], top_p=kw.get("top_p"), top_k=kw.get("top_k"), noise_fn=lambda s, M, V: noises[s], trace=otr, keep_logits=True, tie_break="index")
T = kw["temperature"]
for i in range(kw["steps"]):
    hx, ox = trace["x0"][i].cpu(), otr.x0[i]
    hc, oc = trace["confidence"][i].cpu(), otr.confidence[i]
    bad = (hx != ox).nonzero().view(-1).tolist()
    badc = ((hc - oc).abs() > 1e-4 * oc.abs() + 1e-6).nonzero().view(-1).tolist()
    if bad or badc or not torch.equal(out.history[i].cpu(), hist[i]):
        print("step", i, "x0 mismatch rows", bad, "conf mismatch rows", badc)
        for r in (bad + badc)[:3]:
            lg = otr.mask_logits[i][r] / T
            srt, idx = torch.sort(lg, descending=True)
            p = torch.softmax(srt, -1); cum = torch.cumsum(p, -1)
            nk = int((cum <= kw["top_p"]).sum()) + 1
            print(" row", r, "hip x0", int(hx[r]), "ora x0", int(ox[r]), "hip conf", float(hc[r]), "ora conf", float(oc[r]))
            print("  kept", nk, "cum around boundary", cum[max(0,nk-3):nk+2].tolist(), "dbl", torch.cumsum(torch.softmax(srt.double(),-1),-1)[max(0,nk-3):nk+2].tolist())
            print("  top logits", srt[:nk+2].tolist())
        h = out.history[i].cpu()
        rows = (h != hist[i]).any(1).nonzero().view(-1).tolist()
        prev = hist[i-1] if i else torch.nn.functional.pad(ids, (0,1), value=6)
        mi = prev == 6
        for b in rows[:2]:
            off = int(mi[:b].sum()); n = int(mi[b].sum())
            print(" seq", b, "k", trace["k"][i], "masked pos", mi[b].nonzero().view(-1).tolist())
            print("  hip conf", [repr(float(v)) for v in hc[off:off+n]])
            print("  ora conf", [repr(float(v)) for v in oc[off:off+n]])
            print("  hip row", h[b].tolist()); print("  ora row", hist[i][b].tolist())
        break
else:
    print("all equal")
