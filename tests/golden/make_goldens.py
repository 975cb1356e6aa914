"""Generate golden vectors by importing the reference's own code (build container only).

    python tests/golden/make_goldens.py

Imports ``custom_mydream.{configuration_dream,generation_utils,modeling_dream_length}``
from the read-only mount ``/root/reference`` (two harness shims for the newer
transformers in this image), runs the reference on seeded synthetic
weights/inputs (``dark_amd.synth``), and writes small ``.npz`` fixtures next to
this file.  Nothing of the reference is copied: fixtures hold inputs, settings
and expected outputs only.  The reference never travels to the GPU box.
"""
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/akgr/abduction_model/model")

import numpy as np
import torch

# shim 1: transformers>=5 dropped ROPE_INIT_FUNCTIONS['default'] (4.50 definition restated)
from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS


def _default_rope(config=None, device=None, seq_len=None, **kw):
    hd = config.hidden_size // config.num_attention_heads
    inv = 1.0 / (config.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    return inv, 1.0


ROPE_INIT_FUNCTIONS.setdefault("default", _default_rope)

from custom_mydream.configuration_dream import DreamConfig  # noqa: E402
from custom_mydream import generation_utils as ref_gen  # noqa: E402
from custom_mydream.modeling_dream_length import DreamModel_length  # noqa: E402

# shim 2: GenerationConfig.update() now calls validate(user_set_attributes=...)
ref_gen.DreamGenerationConfig.validate = lambda self, *a, **k: None

from dark_amd import synth  # noqa: E402

MINI = dict(vocab_size=512, hidden_size=192, intermediate_size=384, num_hidden_layers=2,
            num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=64,
            mask_token_id=synth.MASK, pad_token_id=synth.END)
REAL = dict(vocab_size=15080, hidden_size=768, intermediate_size=3072, num_hidden_layers=10,
            num_attention_heads=8, num_key_value_heads=8, max_position_embeddings=64,
            mask_token_id=synth.MASK, pad_token_id=synth.END)
MINI_ENT = dict(nentity=300, nrelation=111)  # 100 + 300 + 111 + 1 = 512


def build(cfg_kw, seed, rich):
    cfg = DreamConfig(**cfg_kw)
    model = DreamModel_length(cfg).eval()
    w = synth.make_weights(cfg, seed=seed, rich=rich)
    missing, unexpected = model.load_state_dict(w, strict=True)
    assert not missing and not unexpected
    return cfg, model


def gen(model, ids, seed, **kw):
    # same construction as the live caller, akgr/abduction_model/main.py:355-363, :393-404
    gc = ref_gen.DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                                       eos_token_id=synth.END, expand_token_id=synth.EXPAND,
                                       delete_token_id=synth.END, sep_token_id=synth.SEP,
                                       max_length=ids.shape[1] + 1)
    torch.manual_seed(seed)
    out = model.diffusion_generate(inputs=ids, generation_config=gc, return_dict_in_generate=True,
                                   output_history=True, number_transfer_tokens=1, **kw)
    return torch.stack(out.history).numpy(), out.sequences.numpy()


def main():
    torch.set_num_threads(8)
    out = {}
    meta = {"torch": torch.__version__, "mini": MINI, "real": REAL, "mini_ent": MINI_ENT, "cases": {}}
    cfg, model = build(MINI, seed=0, rich=True)

    # weight checksums: detect RNG drift between generator and consumer
    w = synth.make_weights(cfg, seed=0, rich=True)
    out["mini_w_checksum"] = np.array([float(w[k].double().sum()) for k in sorted(w)], dtype=np.float64)

    # ---- (i) forward: hidden after embed / each layer / final norm, logits
    abd = synth.make_inputs(2, "abd", seed=1, **MINI_ENT)
    ded = synth.make_inputs(2, "ded", seed=2, **MINI_ENT)
    ids = torch.nn.functional.pad(torch.cat([abd, ded]), (0, 1), value=synth.MASK)
    B, N = ids.shape
    full = torch.ones(B, 1, N, N, dtype=torch.bool)
    pos = torch.arange(N).unsqueeze(0).expand(B, -1)
    with torch.no_grad():
        o = model(ids, full, pos, output_hidden_states=True)
    out["fwd_ids"] = ids.numpy()
    out["fwd_hidden"] = torch.stack(o.hidden_states).numpy()  # [L+1 (+final norm replaces last?)]
    out["fwd_logits"] = o.logits.numpy()
    # padded attention-mask forward (reference builds outer-AND mask and tok_idx, generation_utils.py:462-472)
    am = torch.ones(B, N, dtype=torch.long)
    am[0, 20:40] = 0
    am[2, 3:9] = 0
    tok = am.cumsum(-1) - 1
    tok.masked_fill_(am == 0, 1)
    m4 = torch.logical_and(am.unsqueeze(1).unsqueeze(-2), am.unsqueeze(1).unsqueeze(-1))
    with torch.no_grad():
        o2 = model(ids, m4, tok)
    out["fwdpad_mask"] = am.numpy()
    out["fwdpad_logits"] = o2.logits.numpy()

    # ---- (ii) sample_tokens on fixed logits
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(24, 512, generator=g) * 0.6
    logits[3] = logits[3] * 8          # peaked row
    logits[5] = 0.0                    # flat row (all ties)
    logits[7, :200] = logits[7, 200:400]  # many exact duplicates
    out["st_logits"] = logits.numpy()
    st_cases = []
    for ci, (T, tp, tk, margin, negent) in enumerate([
            (0.0, None, None, False, False), (0.0, 0.9, None, False, True), (0.1, 0.9, None, False, True),
            (0.5, 0.9, None, False, True), (1.0, None, None, False, True), (1.0, 0.95, None, True, False),
            (0.5, 0.9, None, False, False), (0.7, None, 20, False, True), (0.3, 0.8, 5, False, True),
            (0.1, 0.5, None, False, True)]):
        torch.manual_seed(100 + ci)
        conf, x0 = ref_gen.sample_tokens(logits, temperature=T, top_p=tp, top_k=tk,
                                         margin_confidence=margin, neg_entropy=negent)
        out[f"st{ci}_conf"] = conf.numpy()
        out[f"st{ci}_x0"] = x0.numpy()
        st_cases.append(dict(T=T, top_p=tp, top_k=tk, margin=margin, neg_entropy=negent, seed=100 + ci))
    meta["cases"]["sample_tokens"] = st_cases

    # ---- (iii) loops on the mini model
    abd8 = synth.make_inputs(8, "abd", seed=11, **MINI_ENT)
    ded8 = synth.make_inputs(8, "ded", seed=12, **MINI_ENT)
    out["loop_abd_ids"] = abd8.numpy()
    out["loop_ded_ids"] = ded8.numpy()
    loops = []

    def add_loop(name, ids, seed, **kw):
        h, s = gen(model, ids, seed, **kw)
        out[name + "_hist"] = h
        loops.append(dict(name=name, seed=seed, kw=kw))

    add_loop("abd_greedy", abd8, 0, steps=16, temperature=0.0, alg="entropy", alg_temp=0, top_p=0.9)
    add_loop("abd_t01", abd8, 2, steps=16, temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9)
    add_loop("abd_t05_s32", abd8, 3, steps=32, temperature=0.5, alg="entropy", alg_temp=0, top_p=0.9)
    add_loop("ded_t01", ded8, 4, steps=32, temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9)
    add_loop("ded_t05_s8", ded8, 5, steps=8, temperature=0.5, alg="entropy", alg_temp=0, top_p=0.9)
    add_loop("abd_maskgit", abd8, 6, steps=12, temperature=0.4, alg="maskgit_plus", top_p=0.95)
    add_loop("abd_margin", abd8, 7, steps=12, temperature=0.4, alg="topk_margin", top_p=0.95)
    add_loop("abd_origin", abd8, 8, steps=12, temperature=0.4, alg="origin", top_p=0.95)
    add_loop("ded_algtemp", ded8, 9, steps=10, temperature=0.3, alg="entropy", alg_temp=0.5, top_p=0.9)
    add_loop("ded_topk", ded8, 10, steps=10, temperature=0.8, alg="entropy", alg_temp=0, top_k=12)

    # (iv) padded attention_mask through diffusion_generate
    am_in = torch.ones_like(abd8)
    am_in[1, 30:44] = 0
    am_in[5, 14:20] = 0
    out["loop_pad_mask"] = am_in.numpy()
    add_loop("abd_padmask", abd8, 13, steps=16, temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9,
             attention_mask=am_in)
    loops[-1]["kw"] = {k: v for k, v in loops[-1]["kw"].items() if k != "attention_mask"}
    loops[-1]["attention_mask"] = "loop_pad_mask"

    # (v) ragged mask counts: exercises the batch-average k and the re-mask quirk (generation_utils.py:520-534)
    rag = abd8.clone()
    rag[0, 1:8] = torch.arange(101, 108)       # 3+1 masks left
    rag[3, 1:10] = torch.arange(111, 120)      # 1+1 masks left
    rag[6, 20:30] = synth.MASK                 # 21 masks
    out["loop_rag_ids"] = rag.numpy()
    add_loop("rag_t0", rag, 14, steps=6, temperature=0.0, alg="entropy", alg_temp=0, top_p=0.9)
    loops[-1]["ids"] = "loop_rag_ids"
    meta["cases"]["loops"] = loops

    # ---- real shape, a handful of rows
    del model
    cfg_r, model_r = build(REAL, seed=0, rich=False)
    wr = synth.make_weights(cfg_r, seed=0, rich=False)
    out["real_w_checksum"] = np.array([float(wr["lm_head.weight"].double().sum()),
                                       float(wr["model.layers.9.mlp.down_proj.weight"].double().sum())])
    abd4 = synth.make_inputs(4, "abd", seed=1)
    out["real_abd_ids"] = abd4.numpy()
    x = torch.nn.functional.pad(abd4, (0, 1), value=synth.MASK)
    with torch.no_grad():
        lg = model_r(x, torch.ones(4, 1, 46, 46, dtype=torch.bool), torch.arange(46).unsqueeze(0).expand(4, -1)).logits
    out["real_logits_row0_pos0_3"] = lg[0, :3].numpy()   # hidden rows 0..2 -> predict positions 1..3 after the shift
    h, s = gen(model_r, abd4, 2, steps=16, temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9)
    out["real_abd_t01_hist"] = h
    h, s = gen(model_r, abd4, 2, steps=8, temperature=0.0, alg="entropy", alg_temp=0, top_p=0.9)
    out["real_abd_greedy_hist"] = h

    np.savez_compressed(os.path.join(HERE, "reference_goldens.npz"), **out)
    with open(os.path.join(HERE, "reference_goldens.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(out), "arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    main()
