import sys, json, torch, numpy as np
sys.path.insert(0, '.')
from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, synth
from oracle import dream_oracle as O
meta = json.load(open('tests/golden/reference_goldens.json'))
kw = dict(meta["real"], num_hidden_layers=2)
cfg = DreamConfig(**kw); m = DreamModel_length(cfg); w = synth.make_weights(cfg, 0, False); m.load_state_dict(w); m = m.to('cuda')
ocfg = O.OracleConfig(**{k: v for k, v in kw.items() if k in O.OracleConfig.__dataclass_fields__})
ids = synth.make_inputs(64, "abd", seed=1)
g = torch.Generator().manual_seed(99); noises = {}
def noise_fn(step, M, V):
    if step not in noises: noises[step] = torch.empty(M, V).exponential_(1, generator=g)
    return noises[step]
gc = DreamGenerationConfig(mask_token_id=6, pad_token_id=2, max_length=46)
trace = {}
out = m.diffusion_generate(inputs=ids.cuda(), generation_config=gc, return_dict_in_generate=True, output_history=True, steps=16, temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9, _noise_fn=noise_fn, _trace=trace)
otr = O.SampleTrace()
seq, hist = O.sample_loop(w, ocfg, ids, max_length=46, steps=16, mask_token_id=6, alg="entropy", alg_temp=0, temperature=0.1, top_p=0.9, noise_fn=lambda s, M, V: noises[s], trace=otr, keep_logits=True)
for i in range(16):
    h = out.history[i].cpu()
    if not torch.equal(h, hist[i]):
        rows = (h != hist[i]).any(1).nonzero().view(-1)
        print("step", i, "k", trace["k"][i], "rows differing", rows.tolist())
        prev = hist[i-1] if i else torch.nn.functional.pad(ids, (0,1), value=6)
        mi = (prev == 6)
        b = int(rows[0])
        off = int(mi[:b].sum()); n = int(mi[b].sum())
        print("masked pos", mi[b].nonzero().view(-1).tolist())
        print("hip conf", trace["confidence"][i][off:off+n].cpu().tolist())
        print("ora conf", otr.confidence[i][off:off+n].tolist())
        print("hip x0", trace["x0"][i][off:off+n].cpu().tolist())
        print("ora x0", otr.x0[i][off:off+n].tolist())
        print("hip row", h[b].tolist()); print("ora row", hist[i][b].tolist()); print("prev   ", prev[b].tolist())
        break
