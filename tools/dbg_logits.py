import sys; sys.path.insert(0, '.')
import torch
from dark_amd import DreamConfig, DreamModel_length, synth
cfg = DreamConfig(vocab_size=15080, mask_token_id=6, pad_token_id=2)
m = DreamModel_length(cfg); m.load_state_dict(synth.make_weights(cfg, 0)); m = m.to('cuda').to(torch.bfloat16)
for kind in ("abd", "ded"):
    ids = synth.make_inputs(64, kind, seed=1).cuda()
    x = torch.nn.functional.pad(ids, (0, 1), value=6)
    lg = m(x).logits.float()
    lg = torch.cat([lg[:, :1], lg[:, :-1]], 1)[x == 6] / 0.1
    mx = lg.max(-1, keepdim=True).values
    d = mx - lg
    p = torch.softmax(lg, -1)
    head = d < 10
    print(kind, "rows", lg.shape[0], "std", lg.std(-1).mean().item(), "n within 10:", head.sum(-1).float().mean().item(), head.sum(-1).max().item(),
          "head mass", (p * head).sum(-1).mean().item(), (p * head).sum(-1).min().item(),
          "n within 28:", (d < 27.8).sum(-1).float().mean().item())
    sp = p.sort(-1, descending=True).values.cumsum(-1)
    print("   kept set size", ((sp <= 0.9).sum(-1) + 1).float().mean().item(), "max p", p.max(-1).values.mean().item())
