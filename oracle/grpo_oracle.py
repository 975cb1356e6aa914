"""CPU oracle for the forward-only log-prob path of DARK's diffusion GRPO trainer.

TEST INFRASTRUCTURE ONLY (same rule as oracle/dream_oracle.py): only ``tests/``, ``__graft_entry__.smoke()`` and
benchmark baseline legs may import this module.

From-scratch restatement of ``akgr/abduction_model/custom_grpo2.py``: ``selective_log_softmax`` (:89-161),
``DiffuGRPOTrainer.forward_process`` (:350-387), ``get_logits`` (:409-418) and ``_get_per_token_logps``
(:421-498), as the trainer runs them under ``no_grad`` for the old / reference policy (:681-709).
Pinning: fixtures produced by executing those very definitions (tests/golden/make_goldens_grpo.py), checked by
tests/test_oracle_grpo_golden.py.  On CPU the reference's ``autocast('cuda', bf16)`` is inactive, so goldens and
oracle are fp32.
"""
from __future__ import annotations

import random
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .dream_oracle import OracleConfig, forward_logits


def set_seed(seed: int) -> None:
    """transformers.set_seed as called at custom_grpo2.py:351 (python, numpy and torch generators)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def selective_log_softmax(logits: torch.Tensor, index: torch.Tensor, weights: Optional[torch.Tensor] = None,
                          mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """custom_grpo2.py:89-161.  logits [I*3*B, N, V] ordered (iteration, version, row); index / mask [I*B, N];
    weights [3I].  Returns [I*B, N]: (p0 + where(mask, p1*w1, p2*w2)) / 2, or p0 without weights / mask."""
    full = logits.size(0) // 3
    iters = weights.size(0) // 3
    B = full // iters
    N, V = logits.shape[1:]
    lp = F.log_softmax(logits.view(iters, 3, B, N, V), dim=-1)
    idx = index.view(iters, 1, B, N, 1).expand(iters, 3, B, N, 1)
    lp = lp.gather(-1, idx).squeeze(-1)                                   # [I, 3, B, N]
    if weights is None or mask is None:
        return lp[:, 0].reshape(full, N)
    w = weights.view(iters, 3)
    weighted = torch.where(mask.view(iters, B, N), lp[:, 1] * w[:, 1, None, None], lp[:, 2] * w[:, 2, None, None])
    return ((lp[:, 0] + weighted) / 2).reshape(full, N)


def forward_process(batch: torch.Tensor, mask_target: torch.Tensor, mask_id: int, seed: Optional[int] = None,
                    accumulate: bool = False, gradient_accumulation_steps: int = 1,
                    rand_fn: Optional[Callable[[int, int], torch.Tensor]] = None):
    """custom_grpo2.py:350-387.  Returns ([x_all_masked, x_masked_where_rand<t, x_masked_where_rand>t],
    [1, 1/t, 1/(1-t)], completion_mask).  `rand_fn(rows, l)` injects the uniforms (GPU parity tests)."""
    set_seed(seed)
    b, l = batch.shape
    mask_ratio = random.uniform(0.2, 0.8)
    t_p = torch.ones(b) * mask_ratio
    draw = rand_fn or (lambda rows, cols: torch.rand((rows, cols)))
    if accumulate:
        r = draw(b // gradient_accumulation_steps, l)
        r = torch.cat([r] * gradient_accumulation_steps, dim=0)
    else:
        r = draw(b, l)
    query = mask_target.bool()
    noisy = [torch.where(query, mask_id, batch)]
    is_mask = (r < t_p.unsqueeze(1)) & query
    completion_mask = is_mask
    noisy.append(torch.where(is_mask, mask_id, batch))
    is_mask = (r > t_p.unsqueeze(1)) & query
    noisy.append(torch.where(is_mask, mask_id, batch))
    return noisy, [1, 1 / mask_ratio, 1 / (1 - mask_ratio)], completion_mask


def get_logits(w, cfg: OracleConfig, batch: torch.Tensor) -> torch.Tensor:
    """custom_grpo2.py:409-418: full-attention forward, logits shifted right by one position."""
    logits = forward_logits(w, cfg, batch)
    return torch.cat([logits[:, 0:1], logits[:, :-1]], dim=1)


def per_token_logps(w, cfg: OracleConfig, input_ids: torch.Tensor, mask_attention: torch.Tensor,
                    mask_seeds: Sequence[int], mask_id: int, gradient_accumulation_steps: int = 1,
                    rand_fn: Optional[Callable[[int, int, int], torch.Tensor]] = None) -> torch.Tensor:
    """custom_grpo2.py:421-498.  input_ids / mask_attention [I, B, N]; returns fp32 [B, I, N]."""
    if input_ids.dim() != 3:
        raise ValueError(f"Expected input_ids to have 3 dimensions, got {input_ids.dim()}")
    iters, B, N = input_ids.size()
    if len(mask_seeds) != iters:
        raise ValueError(f"Expected mask_seeds length to be {iters}, got {len(mask_seeds)}")
    seqs: List[torch.Tensor] = []
    weights: List[float] = []
    masks: List[torch.Tensor] = []
    for i, seed in enumerate(mask_seeds):
        noisy, tw, cm = forward_process(input_ids[i], mask_attention[i], mask_id, seed=seed, accumulate=iters > 1,
                                        gradient_accumulation_steps=gradient_accumulation_steps,
                                        rand_fn=(lambda r, c, i=i: rand_fn(i, r, c)) if rand_fn else None)
        seqs.extend(noisy)
        weights.extend(tw)
        masks.append(cm)
    logits = get_logits(w, cfg, torch.cat(seqs, dim=0))
    out = selective_log_softmax(logits, input_ids.reshape(iters * B, N), torch.tensor(weights), torch.cat(masks, 0))
    return out.view(iters, B, N).permute(1, 0, 2).to(torch.float32)
