"""Same seed, same inputs, N runs of the full-size bf16 denoise loop: are the trajectories identical?  Also checks the
forward alone (hidden states) and the sampler alone (fixed logits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, _hip, synth
dev = torch.device("cuda")
cfg = DreamConfig(vocab_size=synth.vocab_size(**synth.FB15K237), mask_token_id=synth.MASK, pad_token_id=synth.END)
m = DreamModel_length(cfg); m.load_state_dict(synth.make_weights(cfg, seed=0), strict=True)
m = m.to(dev).to(torch.bfloat16)
ids = synth.make_inputs(512, "ded", seed=21).to(dev)
gc = DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS, eos_token_id=synth.END,
                           sep_token_id=synth.SEP, max_length=46)
import warnings; warnings.filterwarnings("ignore")
x = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
lg = [m(x).logits.float().clone() for _ in range(4)]
print("forward logits identical over 4 runs:", all(torch.equal(lg[0], l) for l in lg[1:]),
      "max diff", max(float((lg[0] - l).abs().max()) for l in lg[1:]))
runs = []
for r in range(4):
    torch.manual_seed(77)
    out = m.diffusion_generate(inputs=ids, generation_config=gc, return_dict_in_generate=True, output_history=True, steps=32,
                               temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9, skip_noop_steps=False)
    runs.append(torch.stack(out.history))
for r in range(1, 4):
    d = (runs[0] != runs[r])
    first = int(d.flatten(1).any(1).nonzero()[0]) if d.any() else -1
    print(f"run {r} vs 0: identical={not bool(d.any())} first differing step {first} rows differing {int(d.any(0).any(1).sum())}")
