"""Deterministic synthetic weights and query batches (no dataset, no tokenizer, no checkpoint).

Weights follow the reference's ``_init_weights`` distribution
(modeling_dream_length.py:538-547: normal(0, 0.02), zero biases, zero pad row,
unit norm weights); ``rich=True`` additionally randomises biases and norm
weights so parity tests exercise them.

Inputs follow the fixed layouts the reference builds on the host
(akgr/tokenizer.py:287-338): ``[BOS, query x10, SEP, obs x32, EOS]``
with the query (abduction) or the observation (deduction) side masked.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

# special ids, akgr/configs/config-dataloader.yml:1-15
BOS, END, UNK, SEP, EXPAND, MASK = 0, 2, 3, 4, 5, 6
OPS = (15, 16, 17)
OFFSET = 100
QRY_LEN, ANS_LEN = 10, 32
FB15K237 = dict(nentity=14505, nrelation=474)   # V = 100 + 14505 + 474 + 1 = 15080
DBPEDIA50 = dict(nentity=24624, nrelation=702)  # V = 25427
WN18RR = dict(nentity=40943, nrelation=22)      # V = 41066 (11 relations, doubled for inverse edges, load_kg_util.py:116-118)


def vocab_size(nentity: int, nrelation: int) -> int:
    """ntoken + 1 (akgr/tokenizer.py:19-27, akgr/abduction_model/main.py:528)."""
    return OFFSET + nentity + nrelation + 1


def state_dict_layout(vocab: int, d: int, ff: int, layers: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Names/shapes of ``DreamModel_length.state_dict()`` in order."""
    names: List[Tuple[str, Tuple[int, ...]]] = [("model.embed_tokens.weight", (vocab, d))]
    for i in range(layers):
        p = f"model.layers.{i}."
        names += [
            (p + "self_attn.q_proj.weight", (d, d)), (p + "self_attn.q_proj.bias", (d,)),
            (p + "self_attn.k_proj.weight", (d, d)), (p + "self_attn.k_proj.bias", (d,)),
            (p + "self_attn.v_proj.weight", (d, d)), (p + "self_attn.v_proj.bias", (d,)),
            (p + "self_attn.o_proj.weight", (d, d)),
            (p + "mlp.gate_proj.weight", (ff, d)), (p + "mlp.up_proj.weight", (ff, d)),
            (p + "mlp.down_proj.weight", (d, ff)),
            (p + "input_layernorm.weight", (d,)), (p + "post_attention_layernorm.weight", (d,)),
        ]
    names += [("model.norm.weight", (d,)), ("lm_head.weight", (vocab, d))]
    return names


def make_weights(config, seed: int = 0, rich: bool = False, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """fp32 CPU state_dict for ``config`` (anything with the DreamConfig shape attributes)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in state_dict_layout(config.vocab_size, config.hidden_size, config.intermediate_size,
                                         config.num_hidden_layers):
        if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
            t = torch.ones(shape)
            if rich:
                t = t + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * std if rich else torch.zeros(shape)
        else:
            t = torch.randn(shape, generator=g) * std
        out[name] = t
    pad = getattr(config, "pad_token_id", None)
    if pad is not None and 0 <= pad < config.vocab_size:
        out["model.embed_tokens.weight"][pad].zero_()
    return out


def make_inputs(batch: int, kind: str = "abd", nentity: int = FB15K237["nentity"],
                nrelation: int = FB15K237["nrelation"], seed: int = 1) -> torch.Tensor:
    """int64 ``[batch, 45]`` ids in the reference's fixed layout; ``kind`` in {"abd", "ded"}."""
    g = torch.Generator().manual_seed(seed)
    n_in = 1 + QRY_LEN + 1 + ANS_LEN + 1
    x = torch.full((batch, n_in), END, dtype=torch.long)
    x[:, 0] = BOS
    x[:, 1 + QRY_LEN] = SEP
    ent_lo, rel_lo = OFFSET, OFFSET + nentity
    if kind == "abd":
        x[:, 1:1 + QRY_LEN] = MASK
        n_ans = torch.randint(1, ANS_LEN + 1, (batch,), generator=g)
        # sorted, distinct entity ids: sorted uniforms over a range shrunk by ANS_LEN, plus arange
        r = torch.rand(batch, ANS_LEN, generator=g).sort(dim=1).values
        ents = (r * (nentity - ANS_LEN)).long() + torch.arange(ANS_LEN) + ent_lo
        keep = torch.arange(ANS_LEN)[None, :] < n_ans[:, None]
        x[:, 2 + QRY_LEN:2 + QRY_LEN + ANS_LEN] = torch.where(keep, ents, torch.full_like(ents, END))
    elif kind == "ded":
        x[:, 2 + QRY_LEN:2 + QRY_LEN + ANS_LEN] = MASK
        n_act = torch.randint(2, QRY_LEN + 1, (batch,), generator=g)
        kinds = torch.randint(0, 3, (batch, QRY_LEN), generator=g)
        ops = torch.tensor(OPS)[torch.randint(0, 3, (batch, QRY_LEN), generator=g)]
        rels = torch.randint(rel_lo, rel_lo + nrelation, (batch, QRY_LEN), generator=g)
        ents = torch.randint(ent_lo, ent_lo + nentity, (batch, QRY_LEN), generator=g)
        acts = torch.where(kinds == 0, ops, torch.where(kinds == 1, rels, ents))
        keep = torch.arange(QRY_LEN)[None, :] < n_act[:, None]
        x[:, 1:1 + QRY_LEN] = torch.where(keep, acts, torch.full_like(acts, END))
    else:
        raise ValueError(f"unknown layout kind: {kind}")
    return x


def make_var_inputs(count: int, kind: str = "abd", n_masks: int = 8, nentity: int = FB15K237["nentity"],
                    nrelation: int = FB15K237["nrelation"], seed: int = 1) -> List[torch.Tensor]:
    """`count` un-padded int64 ``[1, L]`` inputs of the variable-length path (akgr/tokenizer.py:154-184):
    abd ``[BOS, MASK x n_masks, SEP, observation, EOS]``, ded ``[BOS, query, SEP, MASK x n_masks, EOS]``."""
    fixed = make_inputs(count, kind, nentity, nrelation, seed)  # supplies the unmasked side
    out = []
    for b in range(count):
        row = fixed[b]
        if kind == "abd":
            obs = row[2 + QRY_LEN:2 + QRY_LEN + ANS_LEN]
            obs = obs[obs != END]
            ids = [BOS] + [MASK] * n_masks + [SEP] + obs.tolist() + [END]
        elif kind == "ded":
            qry = row[1:1 + QRY_LEN]
            qry = qry[qry != END]
            ids = [BOS] + qry.tolist() + [SEP] + [MASK] * n_masks + [END]
        else:
            raise ValueError(f"unknown layout kind: {kind}")
        out.append(torch.tensor([ids], dtype=torch.long))
    return out


def boost_edit_tokens(weights: Dict[str, torch.Tensor], expand_gain: float = 0.0, delete_gain: float = 0.0,
                      mask_id: int = MASK, expand_id: int = EXPAND, delete_id: int = END) -> Dict[str, torch.Tensor]:
    """A random-init model almost never emits EXPAND / DELETE.  Point their lm_head rows along the MASK
    embedding (the residual stream of a masked position keeps a large component of it) so that synthetic
    runs exercise the edit branches; gains are in units of the typical row norm."""
    w = dict(weights)
    head = w["lm_head.weight"].clone()
    e = w["model.embed_tokens.weight"][mask_id]
    unit = e / e.norm()
    typical = head.norm(dim=1).median()
    if expand_gain:
        head[expand_id] = unit * typical * expand_gain
    if delete_gain:
        head[delete_id] = unit * typical * delete_gain
    w["lm_head.weight"] = head
    return w


# ---- GPT-2 backbone (akgr/abduction_model/model/custom_gpt.py)
def gpt2_state_dict_layout(cfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """Names/shapes of transformers' ``GPT2LMHeadModel.state_dict()`` (Conv1D weights are [in, out]); ``lm_head.weight``
    is tied to ``transformer.wte.weight`` and not listed."""
    d, inner = cfg.n_embd, (getattr(cfg, "n_inner", None) or 4 * cfg.n_embd)
    names: List[Tuple[str, Tuple[int, ...]]] = [("transformer.wte.weight", (cfg.vocab_size, d)),
                                                ("transformer.wpe.weight", (cfg.n_positions, d))]
    for i in range(cfg.n_layer):
        p = f"transformer.h.{i}."
        names += [(p + "ln_1.weight", (d,)), (p + "ln_1.bias", (d,)),
                  (p + "attn.c_attn.weight", (d, 3 * d)), (p + "attn.c_attn.bias", (3 * d,)),
                  (p + "attn.c_proj.weight", (d, d)), (p + "attn.c_proj.bias", (d,)),
                  (p + "ln_2.weight", (d,)), (p + "ln_2.bias", (d,)),
                  (p + "mlp.c_fc.weight", (d, inner)), (p + "mlp.c_fc.bias", (inner,)),
                  (p + "mlp.c_proj.weight", (inner, d)), (p + "mlp.c_proj.bias", (d,))]
    names += [("transformer.ln_f.weight", (d,)), ("transformer.ln_f.bias", (d,))]
    return names


def make_gpt2_weights(cfg, seed: int = 0, rich: bool = False, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """fp32 CPU state_dict with GPT2PreTrainedModel._init_weights' distribution (normal(0, 0.02), zero biases, unit
    LayerNorm weights); ``rich=True`` randomises biases and LayerNorm parameters so parity tests exercise them."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in gpt2_state_dict_layout(cfg):
        if ".ln_" in name:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            if rich:
                t = t + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * std if rich else torch.zeros(shape)
        else:
            t = torch.randn(shape, generator=g) * std
        out[name] = t
    out["lm_head.weight"] = out["transformer.wte.weight"]
    return out


def boost_gpt2_edit_tokens(weights: Dict[str, torch.Tensor], expand_gain: float = 0.0, delete_gain: float = 0.0,
                           mask_id: int = MASK, expand_id: int = EXPAND, delete_id: int = END
                           ) -> Dict[str, torch.Tensor]:
    """boost_edit_tokens for the tied head: the wte rows of EXPAND / DELETE are pointed along the MASK embedding."""
    w = dict(weights)
    wte = w["transformer.wte.weight"].clone()
    unit = wte[mask_id] / wte[mask_id].norm()
    typical = wte.norm(dim=1).median()
    if expand_gain:
        wte[expand_id] = unit * typical * expand_gain
    if delete_gain:
        wte[delete_id] = unit * typical * delete_gain
    w["transformer.wte.weight"] = w["lm_head.weight"] = wte
    return w
