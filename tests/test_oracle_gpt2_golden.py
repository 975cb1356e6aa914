"""Pins oracle/gpt2_oracle.py against tests/golden/reference_goldens_gpt2.npz: logits of transformers'
own GPT2LMHeadModel for both mask semantics, and the histories the reference's `DiffusionModel.diffusion_generate`
source produced on that transformers build (tests/golden/make_goldens_gpt2.py)."""
import json
import os

import numpy as np
import pytest
import torch

from dark_amd import synth
from oracle import dream_oracle as O
from oracle import gpt2_oracle as G

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLDEN_DIR, "reference_goldens_gpt2.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLDEN_DIR, "reference_goldens_gpt2.npz"))


def oracle_cfg(shape):
    return G.Gpt2OracleConfig(**{k: v for k, v in shape.items() if k in G.Gpt2OracleConfig.__dataclass_fields__})


def split_history(g, name):
    lens, flat = g[name + "_hist_len"], g[name + "_hist"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    return [flat[offs[i]:offs[i + 1]] for i in range(len(lens))]


def run_loop(cfg, w, g, c, causal=True, **over):
    kw = dict(c["kw"])
    ids = torch.from_numpy(g["in_" + c["input"]]).clone()
    torch.manual_seed(c["seed"])
    return O.expand_loop(None, None, ids, max_length=c["max_length"], mask_token_id=synth.MASK, eos_token_id=synth.END,
                         delete_token_id=synth.END, expand_token_id=synth.EXPAND,
                         expand_budget=kw.get("expand_budget"), pad_delete_to_right=kw.get("pad_delete_to_right", False),
                         number_transfer_tokens=kw["number_transfer_tokens"], alg=kw["alg"], alg_temp=kw["alg_temp"],
                         temperature=kw["temperature"], top_p=kw.get("top_p"), top_k=kw.get("top_k"),
                         forward_fn=G.expand_forward_fn(w, cfg, causal), **over)


def test_forward_matches_transformers(gold):
    meta, g = gold
    for f in meta["forward"]:
        tag, cfg = f["tag"], oracle_cfg(f["shape"])
        w = synth.make_gpt2_weights(cfg, seed=0, rich=True)
        ids, am, pos = (torch.from_numpy(g[f"fwd_{tag}_{k}"]) for k in ("ids", "am", "pos"))
        live = am.bool()
        for name, causal in (("causal", True), ("bidir", False)):
            got = G.forward_logits(w, cfg, ids, am, pos, causal=causal)
            want = torch.from_numpy(g[f"fwd_{tag}_{name}"])
            # rows outside the window (attention_mask == 0) are never read by the sampler (custom_gpt.py:348)
            err = float((got - want)[live].abs().max())
            assert err < 2e-5 * max(1.0, float(want.abs().max())), (tag, name, err)
        got = G.forward_logits(w, cfg, ids, None, None, causal=True)
        assert float((got - torch.from_numpy(g[f"fwd_{tag}_nomask_causal"])).abs().max()) < 2e-5 * max(1.0, float(got.abs().max()))


def test_loop_matches_reference_class(gold):
    """custom_gpt.py:303-444 is generation_expand_utils.py:358-503 with a [B, N] mask: dream_oracle.expand_loop fed the
    GPT-2 forward must reproduce every snapshot the reference's DiffusionModel recorded (causal in that build)."""
    meta, g = gold
    cfg = oracle_cfg(meta["mini"])
    w = synth.boost_gpt2_edit_tokens(synth.make_gpt2_weights(cfg, seed=0, rich=True), **meta["boost"])
    for c in meta["loops"]:
        x, hist = run_loop(cfg, w, g, c)
        want = split_history(g, c["name"])
        assert len(hist) == len(want), c["name"]
        for i, (a, b) in enumerate(zip(hist, want)):
            assert np.array_equal(a.numpy(), b), f"{c['name']} snapshot {i}"
        assert np.array_equal(x.numpy(), g[c["name"] + "_seq"]), c["name"]
    assert sum(c["shrinks"] for c in meta["loops"]) >= 10 and sum(c["grows"] for c in meta["loops"]) >= 10


def test_mask_semantics_differ(gold):
    meta, g = gold
    cfg = oracle_cfg(meta["mini"])
    w = synth.make_gpt2_weights(cfg, seed=0, rich=True)
    ids = torch.from_numpy(g["fwd_mini_ids"])
    a, b = G.forward_logits(w, cfg, ids, causal=True), G.forward_logits(w, cfg, ids, causal=False)
    assert not torch.allclose(a[:, 0], b[:, 0], atol=1e-3)    # position 0 sees only itself under the causal mask
