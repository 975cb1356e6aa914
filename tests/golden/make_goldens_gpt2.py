"""Golden vectors of the GPT-2 backbone, generated in the build container only.

    python tests/golden/make_goldens_gpt2.py

Two sources, both on seeded synthetic weights (``dark_amd.synth.make_gpt2_weights``):

A. ``transformers.GPT2LMHeadModel`` (the class the reference subclasses; the installed build, 5.15.0): logits for the
   reference's call convention ``model(input_ids, attention_mask [B, N], position_ids)`` -- causal in this build --
   and for a 4-D additive mask that holds only the key padding -- bidirectional, the semantics the reference's
   ``attn.bias.fill_(True)`` (custom_gpt.py:34-36) states as its intent.
B. the reference's own ``DiffusionModel`` (custom_gpt.py:25-444).  The module cannot be imported (``akgr/__init__.py``
   imports ``pykeen``, absent here), so the class definition is compiled straight from the reference's source file with
   ``ast`` (as make_goldens_grpo.py does) in a namespace that holds what the module's own imports would have put there;
   ``akgr/utils/generate_utils.py`` is loaded by path.  Harness shims: (1) ``GPT2Attention`` of transformers 5.x no
   longer registers the ``bias`` buffer the class fills (:35-36) -- it is re-registered (and, as in 5.x, read by
   nothing); (2) ``DreamGenerationConfig.validate`` takes the keyword transformers 5.x passes (see make_goldens.py).
   ``diffusion_generate`` is then run the way ``main.test`` runs the Dream variant (akgr/abduction_model/main.py:236-274).

Writes ``reference_goldens_gpt2.{npz,json}``.  Nothing of the reference is copied and it never travels to the GPU box.
"""
import ast
import copy
import importlib.util
import json
import os
import sys
import warnings

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/akgr"

import numpy as np
import torch
import torch.nn.functional as F
from transformers import GPT2Config, GPT2LMHeadModel
from transformers.models.gpt2 import modeling_gpt2
from transformers.modeling_outputs import CausalLMOutput
from transformers.utils import ModelOutput, is_torchdynamo_compiling, logging

from dark_amd import synth  # noqa: E402

MINI = dict(vocab_size=512, n_positions=64, n_embd=128, n_layer=2, n_head=2)
WIDE = dict(vocab_size=1000, n_positions=64, n_embd=768, n_layer=1, n_head=12)   # the stock gpt2 width / head count
MINI_ENT = dict(nentity=300, nrelation=111)
BOOST = dict(expand_gain=2.5, delete_gain=2.0)
SPECIAL = dict(bos_token_id=synth.BOS, eos_token_id=synth.END, pad_token_id=synth.END)


def hf_model(shape, weights):
    cfg = GPT2Config(**shape, **SPECIAL)
    m = GPT2LMHeadModel(cfg).eval()
    missing, unexpected = m.load_state_dict(weights, strict=False)
    assert not unexpected and all(k.endswith(".attn.bias") or k.endswith("masked_bias") for k in missing), (missing, unexpected)
    return m


def forward_cases(out, meta):
    for tag, shape, B, N in (("mini", MINI, 3, 20), ("wide", WIDE, 2, 24)):
        w = synth.make_gpt2_weights(GPT2Config(**shape), seed=0, rich=True)
        m = hf_model(shape, w)
        g = torch.Generator().manual_seed(1)
        ids = torch.randint(0, shape["vocab_size"], (B, N), generator=g)
        am = torch.ones(B, N, dtype=torch.long)
        am[1, N - 6:] = 0                                   # a shorter window, as the sampler passes (custom_gpt.py:344-346)
        pos = am.cumsum(-1) - 1
        pos.masked_fill_(am == 0, 1)                         # :352-353
        with torch.no_grad():
            causal = m(input_ids=ids, attention_mask=am, position_ids=pos).logits
            m4 = torch.zeros(B, 1, N, N).masked_fill_(~am.bool()[:, None, None, :].expand(B, 1, N, N),
                                                      torch.finfo(torch.float32).min)
            bidir = m(input_ids=ids, attention_mask=m4, position_ids=pos).logits
            nomask = m(input_ids=ids).logits                 # attention_mask=None -> ones (:82-83), arange positions
        # sanity: the two semantics differ, and "causal" really is (a later token does not reach an earlier logit)
        ids2 = ids.clone()
        ids2[0, N - 1] = (ids2[0, N - 1] + 1) % shape["vocab_size"]
        with torch.no_grad():
            assert torch.equal(m(input_ids=ids2, attention_mask=am, position_ids=pos).logits[0, :N - 1], causal[0, :N - 1])
            assert not torch.equal(m(input_ids=ids2, attention_mask=m4, position_ids=pos).logits[0, :N - 1], bidir[0, :N - 1])
        out[f"fwd_{tag}_ids"], out[f"fwd_{tag}_am"], out[f"fwd_{tag}_pos"] = ids.numpy(), am.numpy(), pos.numpy()
        out[f"fwd_{tag}_causal"], out[f"fwd_{tag}_bidir"] = causal.numpy(), bidir.numpy()
        out[f"fwd_{tag}_nomask_causal"] = nomask.numpy()
        meta["forward"].append(dict(tag=tag, shape=shape, B=B, N=N))
        print(f"forward {tag}: logits {tuple(causal.shape)}, |causal - bidir| max {float((causal - bidir).abs().max()):.3f}")


def load_reference_class():
    """class DiffusionModel, compiled from the reference's source text (see the module docstring)."""
    spec = importlib.util.spec_from_file_location("ref_generate_utils", os.path.join(REF, "utils", "generate_utils.py"))
    gu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gu)
    gu.DreamGenerationConfig.validate = lambda self, *a, **k: None                        # shim 2
    with open(os.path.join(REF, "abduction_model", "model", "custom_gpt.py")) as f:
        tree = ast.parse(f.read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DiffusionModel"]
    assert len(cls) == 1
    from typing import Any, Dict, Optional, Tuple, Union
    ns = dict(GPT2LMHeadModel=GPT2LMHeadModel, GPT2Config=GPT2Config, CausalLMOutput=CausalLMOutput, torch=torch, F=F,
              DreamGenerationConfig=gu.DreamGenerationConfig, DreamModelOutput=gu.DreamModelOutput,
              sample_tokens=gu.sample_tokens, copy=copy, warnings=warnings, ModelOutput=ModelOutput,
              is_torchdynamo_compiling=is_torchdynamo_compiling, logging=logging,
              logger=logging.get_logger("custom_gpt"), Any=Any, Dict=Dict, Optional=Optional, Tuple=Tuple, Union=Union)
    exec(compile(ast.Module(body=cls, type_ignores=[]), "custom_gpt.py", "exec"), ns)

    orig_init = modeling_gpt2.GPT2Attention.__init__

    def init_with_bias(self, config, *a, **k):                                           # shim 1
        orig_init(self, config, *a, **k)
        n = config.max_position_embeddings
        self.register_buffer("bias", torch.tril(torch.ones((n, n), dtype=torch.bool)).view(1, 1, n, n), persistent=False)

    modeling_gpt2.GPT2Attention.__init__ = init_with_bias
    return ns["DiffusionModel"], gu


def loop_cases(out, meta):
    DiffusionModel, gu = load_reference_class()
    cfg = GPT2Config(**MINI, **SPECIAL)
    model = DiffusionModel(cfg).eval()
    w = synth.boost_gpt2_edit_tokens(synth.make_gpt2_weights(cfg, seed=0, rich=True), **BOOST)
    missing, unexpected = model.load_state_dict(w, strict=False)
    assert not unexpected and all(".attn.bias" in k or "masked_bias" in k for k in missing), (missing, unexpected)
    assert all(bool(b.attn.bias.all()) for b in model.transformer.h)        # the reference's fill_(True) ran

    abd = synth.make_var_inputs(3, "abd", 8, seed=3, **MINI_ENT)
    ded = synth.make_var_inputs(2, "ded", 6, seed=4, **MINI_ENT)
    inputs = {"abd0": abd[0], "abd1": abd[1], "abd2": abd[2], "ded0": ded[0], "ded1": ded[1]}
    for k, v in inputs.items():
        out["in_" + k] = v.numpy()

    def add(name, inp, seed, max_length=50, **kw):
        gc = gu.DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                                      eos_token_id=synth.END, expand_token_id=synth.EXPAND, delete_token_id=synth.END,
                                      max_length=max_length)
        torch.manual_seed(seed)
        with torch.no_grad():
            res = model.diffusion_generate(inputs=inputs[inp].clone(), generation_config=gc,
                                           return_dict_in_generate=True, output_history=True, **kw)
        hist = [h.numpy() for h in res.history]
        out[name + "_seq"] = res.sequences.numpy()
        out[name + "_hist_len"] = np.array([len(h) for h in hist], dtype=np.int64)
        out[name + "_hist"] = np.concatenate(hist) if hist else np.zeros(0, dtype=np.int64)
        lens = [len(h) for h in hist]
        n_del = sum(1 for a, b in zip(lens, lens[1:]) if b < a)
        n_exp = sum(1 for a, b in zip(lens, lens[1:]) if b > a)
        meta["loops"].append(dict(name=name, input=inp, seed=seed, max_length=max_length, kw=kw, snapshots=len(hist),
                                  shrinks=n_del, grows=n_exp))
        print(f"{name:20s} snapshots {len(hist):3d}  shrinks {n_del:2d}  grows {n_exp:2d}  final window {lens[-1]}")

    base = dict(alg="entropy", alg_temp=0, top_p=0.9, number_transfer_tokens=1)
    add("abd0_greedy", "abd0", 10, temperature=0.0, **base)
    add("abd1_t05", "abd1", 21, temperature=0.5, **base)
    add("abd2_t01", "abd2", 22, temperature=0.1, **base)
    add("ded0_t05", "ded0", 40, temperature=0.5, **base)
    add("ded1_padright", "ded1", 51, temperature=0.5, pad_delete_to_right=True, **base)
    add("abd2_budget2", "abd2", 52, temperature=0.5, expand_budget=2, **base)
    add("abd0_margin_k2", "abd0", 56, temperature=0.5, alg="topk_margin", alg_temp=0, top_p=0.95, number_transfer_tokens=2)
    add("ded0_maskgit_topk", "ded0", 58, temperature=0.8, alg="maskgit_plus", alg_temp=0, top_k=12, number_transfer_tokens=1)


def main():
    torch.set_num_threads(8)
    out, meta = {}, dict(torch=torch.__version__, mini=MINI, wide=WIDE, mini_ent=MINI_ENT, boost=BOOST, forward=[], loops=[])
    import transformers
    meta["transformers"] = transformers.__version__
    forward_cases(out, meta)
    loop_cases(out, meta)
    np.savez_compressed(os.path.join(HERE, "reference_goldens_gpt2.npz"), **out)
    with open(os.path.join(HERE, "reference_goldens_gpt2.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(out), "arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    main()
