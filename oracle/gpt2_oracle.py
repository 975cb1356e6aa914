"""CPU oracle of the GPT-2 backbone.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in
the product path); it is the checker, never the thing measured or shipped.

Restates what ``DiffusionModel.forward`` computes (akgr/abduction_model/model/custom_gpt.py:58-114): the class adds
nothing to ``GPT2LMHeadModel`` but the call convention, so the arithmetic is that of a THIRD-PARTY dependency that is
absent from /root/reference: ``transformers==4.50.2`` (requirements.txt:177), ``models/gpt2/modeling_gpt2.py`` --
``GPT2Model.forward`` (wte + wpe, blocks, ln_f), ``GPT2Block.forward`` (pre-LN residual blocks), ``GPT2Attention``
(Conv1D c_attn -> split q, k, v -> heads -> softmax(q k^T / sqrt(head_dim) + mask) v -> c_proj), ``GPT2MLP`` (c_fc ->
NewGELUActivation -> c_proj), tied ``lm_head``.  Pinned by ``tests/golden/reference_goldens_gpt2.npz``: logits of the
transformers build installed in the build container (5.15.0) for BOTH mask semantics, and loop histories produced
by the reference's own ``DiffusionModel.diffusion_generate`` source run on that build
(``tests/golden/make_goldens_gpt2.py``; ``tests/test_oracle_gpt2_golden.py`` checks this file against them).

Mask semantics: ``causal=False`` = bidirectional with key padding (the class's stated intent, :34-36); ``causal=True`` =
what the installed transformers computes for the reference's call.  The sampling loop is ``dream_oracle.expand_loop``
(custom_gpt.py:303-444 is the loop of generation_expand_utils.py:358-503 with a [B, N] mask) fed ``forward_fn``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class Gpt2OracleConfig:
    vocab_size: int = 50257
    n_positions: int = 1024
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    layer_norm_epsilon: float = 1e-5


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    """transformers.activations.NewGELUActivation."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def conv1d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """transformers.pytorch_utils.Conv1D: weight is [in, out]; addmm(bias, x, weight)."""
    return torch.addmm(b, x.reshape(-1, x.shape[-1]), w).view(*x.shape[:-1], w.shape[1])


def forward_hidden(w: Dict[str, torch.Tensor], cfg: Gpt2OracleConfig, input_ids: torch.Tensor,
                   attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                   causal: bool = False) -> torch.Tensor:
    """Hidden states after ln_f, [B, N, d].  `attention_mask` [B, N] (1 = attend) masks KEYS only."""
    B, N = input_ids.shape
    H, d = cfg.n_head, cfg.n_embd
    hd = d // H
    if position_ids is None:
        position_ids = torch.arange(N)[None, :].expand(B, N)
    h = w["transformer.wte.weight"][input_ids] + w["transformer.wpe.weight"][position_ids]
    neg = torch.finfo(torch.float32).min
    bias = torch.zeros(B, 1, N, N)
    if attention_mask is not None:
        bias = bias.masked_fill(~attention_mask.bool()[:, None, None, :], neg)
    if causal:
        keep = torch.tril(torch.ones(N, N, dtype=torch.bool))
        bias = bias.masked_fill(~keep[None, None], neg)
    for i in range(cfg.n_layer):
        p = f"transformer.h.{i}."
        x = F.layer_norm(h, (d,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.layer_norm_epsilon)
        qkv = conv1d(x, w[p + "attn.c_attn.weight"], w[p + "attn.c_attn.bias"])
        q, k, v = (t.view(B, N, H, hd).transpose(1, 2) for t in qkv.split(d, dim=2))
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + bias, dim=-1) @ v
        att = att.transpose(1, 2).reshape(B, N, d)
        h = h + conv1d(att, w[p + "attn.c_proj.weight"], w[p + "attn.c_proj.bias"])
        x = F.layer_norm(h, (d,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.layer_norm_epsilon)
        x = gelu_new(conv1d(x, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"]))
        h = h + conv1d(x, w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])
    return F.layer_norm(h, (d,), w["transformer.ln_f.weight"], w["transformer.ln_f.bias"], cfg.layer_norm_epsilon)


def forward_logits(w, cfg, input_ids, attention_mask=None, position_ids=None, causal: bool = False) -> torch.Tensor:
    """custom_gpt.py:100-103: lm_head (tied to wte, no bias) on every position."""
    return forward_hidden(w, cfg, input_ids, attention_mask, position_ids, causal) @ w["transformer.wte.weight"].t()


def expand_forward_fn(w, cfg, causal: bool):
    """`forward_fn` for dream_oracle.expand_loop: that loop builds the Dream-style [B,1,N,N] outer-AND mask
    (generation_expand_utils.py:410-413); the GPT-2 loop passes its [B, N] source vector instead (custom_gpt.py:344-356),
    which is the diagonal of the 4-D mask."""
    def fwd(x, attn4d, tok_idx):
        am = torch.diagonal(attn4d[:, 0], dim1=-2, dim2=-1).to(torch.int64)
        return forward_logits(w, cfg, x, am, tok_idx, causal)
    return fwd
