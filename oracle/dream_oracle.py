"""CPU oracle for DARK's masked-diffusion denoise / unmask path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``dark_amd``) never does and fails loudly when its HIP library is missing.

This is a from-scratch restatement, in plain functional PyTorch on CPU fp32, of
the algorithm the reference runs for this path.  Every function cites the
reference lines it follows (paths relative to ``/root/reference/``):

* ``akgr/abduction_model/model/custom_mydream/modeling_dream_length.py``
  (RMSNorm :69-74, rotary :151-171, rotate_half/apply :175-207, MLP :221-222,
  SDPA attention :353-438, decoder layer :459-524, base model :633-733,
  LM head :771-824)
* ``akgr/abduction_model/model/custom_mydream/generation_utils.py``
  (top_p_logits :37-48, top_k_logits :50-55, sample_tokens :58-90,
  think sample :367-432, _sample :435-548)
* ``akgr/abduction_model/model/custom_mydream/generation_expand_utils.py``
  (variable-length EXPAND/DELETE sampler: diffusion_generate :294-356, _sample :358-503)

Pinning: the reference has no tests or golden vectors of its own.
The oracle is pinned by fixtures generated from the reference's own code,
imported in the build container by ``tests/golden/make_goldens.py``; the
fixtures are committed under ``tests/golden/`` and checked by
``tests/test_oracle_golden.py``.

Weights are a plain ``dict[str, Tensor]`` with the reference's state_dict names
(SURVEY.md §5 "Checkpoint / resume").
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import contextlib

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Shape contract, defaults = configuration_dream.py:29-50."""
    vocab_size: int = 151936
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 10
    num_attention_heads: int = 8
    max_position_embeddings: int = 64
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000.0
    mask_token_id: int = 151666
    pad_token_id: int = 151643

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


# --------------------------------------------------------------------------- model
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """modeling_dream_length.py:69-74."""
    in_dtype = x.dtype
    h = x.to(torch.float32)
    variance = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(variance + eps)
    return weight * h.to(in_dtype)


def rope_inv_freq(head_dim: int, theta: float) -> torch.Tensor:
    """transformers 4.50 ``ROPE_INIT_FUNCTIONS['default']`` as used at
    modeling_dream_length.py:120-122 (SURVEY.md §8c shim 1)."""
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))


def rope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """modeling_dream_length.py:151-171 (fp32, attention_scaling = 1)."""
    inv_freq = rope_inv_freq(head_dim, theta)
    inv = inv_freq[None, :, None].float().expand(position_ids.shape[0], -1, 1)
    pos = position_ids[:, None, :].float()
    freqs = (inv @ pos).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    """modeling_dream_length.py:175-179."""
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def apply_rope(q, k, cos, sin):
    """modeling_dream_length.py:183-207 (unsqueeze_dim=1)."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (_rotate_half(q) * sin), (k * cos) + (_rotate_half(k) * sin)


def attention(w: Dict[str, torch.Tensor], pfx: str, cfg: OracleConfig, h, attn_mask, cos, sin):
    """modeling_dream_length.py:379-438: q/k/v with bias, RoPE, non-causal SDPA, o_proj without bias."""
    B, N, _ = h.shape
    H, hd = cfg.num_attention_heads, cfg.head_dim
    q = F.linear(h, w[pfx + "q_proj.weight"], w[pfx + "q_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
    k = F.linear(h, w[pfx + "k_proj.weight"], w[pfx + "k_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
    v = F.linear(h, w[pfx + "v_proj.weight"], w[pfx + "v_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    o = F.scaled_dot_product_attention(
        q, k, v, attn_mask=attn_mask if isinstance(attn_mask, torch.Tensor) else None,
        dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).contiguous().view(B, N, H * hd)
    return F.linear(o, w[pfx + "o_proj.weight"])


def mlp(w, pfx, h):
    """modeling_dream_length.py:221-222 (SwiGLU, no bias)."""
    return F.linear(F.silu(F.linear(h, w[pfx + "gate_proj.weight"])) * F.linear(h, w[pfx + "up_proj.weight"]),
                    w[pfx + "down_proj.weight"])


def forward_hidden(w, cfg: OracleConfig, input_ids, attn_mask=None, position_ids=None,
                   collect: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """modeling_dream_length.py:633-733: embedding, shared RoPE table, L pre-norm blocks, final RMSNorm."""
    B, N = input_ids.shape
    x = F.embedding(input_ids, w["model.embed_tokens.weight"])
    if position_ids is None:
        position_ids = torch.arange(N).unsqueeze(0)
    cos, sin = rope_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta)
    cos, sin = cos.to(x.dtype), sin.to(x.dtype)
    if collect is not None:
        collect.append(x)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        h = rms_norm(x, w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        x = x + attention(w, p + "self_attn.", cfg, h, attn_mask, cos, sin)
        h = rms_norm(x, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        x = x + mlp(w, p + "mlp.", h)
        if collect is not None:
            collect.append(x)
    x = rms_norm(x, w["model.norm.weight"], cfg.rms_norm_eps)
    if collect is not None:
        collect.append(x)
    return x


def forward_logits(w, cfg, input_ids, attn_mask=None, position_ids=None) -> torch.Tensor:
    """modeling_dream_length.py:771-824 with num_logits_to_keep=0 (all positions)."""
    return F.linear(forward_hidden(w, cfg, input_ids, attn_mask, position_ids), w["lm_head.weight"])


# --------------------------------------------------------------------------- sampling
# exact_nucleus(): evaluate the nucleus prefix sums of top_p_logits in float64 instead of the reference's fp32.  The
# reference's kept set is `cumsum(softmax(sorted)) > top_p` in fp32; the normaliser of an fp32 softmax over V terms
# carries a relative error of a few 1e-6 at V = 41 066 (measured on this torch build: a row whose first 39 tokens hold
# 0.8999976 of the mass comes out as 0.9000022 and loses its 40th token), so which side of top_p a prefix sum within
# that distance falls on is decided by rounding, differently on a CPU and on a device.  Tests of large vocabularies
# compare the kernel (integer masses, good to 2.4e-7) with this exactly-evaluated form of the same rule.
_NUCLEUS_FP64 = False


@contextlib.contextmanager
def exact_nucleus():
    global _NUCLEUS_FP64
    prev, _NUCLEUS_FP64 = _NUCLEUS_FP64, True
    try:
        yield
    finally:
        _NUCLEUS_FP64 = prev


def top_p_logits(logits: torch.Tensor, top_p: float, stable_ties: bool = False) -> torch.Tensor:
    """generation_utils.py:37-48.  The reference's `torch.sort` leaves the order inside a run of equal logits
    unspecified, so which members of a run that straddles the nucleus boundary survive is not pinned by it;
    `stable_ties=True` fixes index order (the device's rule) for tests that plant such runs."""
    sorted_logits, sorted_idx = torch.sort(logits, descending=True, stable=stable_ties)
    if _NUCLEUS_FP64:
        cum = torch.cumsum(F.softmax(sorted_logits.double(), dim=-1), dim=-1)
        top_p = float(torch.tensor(top_p, dtype=logits.dtype))   # the threshold the fp32 comparison uses
    else:
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
    remove = cum > top_p
    remove[..., 1:] = remove[..., :-1].clone()
    remove[..., 0] = 0
    mask = torch.zeros_like(logits, dtype=torch.bool).scatter_(-1, sorted_idx, remove)
    return logits.masked_fill(mask, torch.finfo(logits.dtype).min)


def top_k_logits(logits: torch.Tensor, top_k: int) -> torch.Tensor:
    """generation_utils.py:50-55."""
    top_k = min(top_k, logits.size(-1))
    remove = logits < torch.topk(logits, top_k)[0][..., -1, None]
    return logits.masked_fill(remove, torch.finfo(logits.dtype).min)


def categorical_sample(probs: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``Categorical(probs=probs).sample()`` of generation_utils.py:70, restated.

    torch normalises ``probs`` by its row sum, validates the simplex constraint
    (raises -> the reference's bare ``except`` falls back to greedy, :72-73) and
    draws with ``multinomial(1)``, which is ``argmax(p / q)`` with
    ``q = empty_like(p).exponential_(1)`` drawn from the default generator
    (ATen ``multinomial`` fast path; verified against the reference in
    tests/golden/make_goldens.py).  ``noise`` injects ``q`` for GPU parity tests.
    """
    p = probs / probs.sum(-1, keepdim=True)
    ok = bool(((p >= 0).all(dim=-1) & ((p.sum(-1) - 1).abs() < 1e-6)).all()) if p.numel() else True
    if not ok:
        raise ValueError("probs violate the simplex constraint")
    q = torch.empty_like(p).exponential_(1) if noise is None else noise
    return torch.argmax(p / q, dim=-1)


def sample_tokens(logits, temperature=0.0, top_p=None, top_k=None, margin_confidence=False,
                  neg_entropy=False, noise: Optional[torch.Tensor] = None, stable_ties: bool = False):
    """generation_utils.py:58-90."""
    if temperature > 0:
        logits = logits / temperature
    if top_p is not None and top_p < 1:
        logits = top_p_logits(logits, top_p, stable_ties)
    if top_k is not None:
        logits = top_k_logits(logits, top_k)
    probs = torch.softmax(logits, dim=-1)
    if temperature > 0:
        try:
            x0 = categorical_sample(probs, noise)
            confidence = torch.gather(probs, -1, x0.unsqueeze(-1)).squeeze(-1)
        except ValueError:
            confidence, x0 = probs.max(dim=-1)
    else:
        confidence, x0 = probs.max(dim=-1)
    if margin_confidence:
        sorted_probs, _ = torch.sort(probs, dim=-1, descending=True)
        confidence = sorted_probs[:, 0] - sorted_probs[:, 1]
    if neg_entropy:
        confidence = torch.sum(probs * torch.log(probs + 1e-10), dim=-1)
    return confidence, x0


def transfer_schedule(masks_per_row: float, steps: int, eps: float = 1e-3) -> List[int]:
    """k per step for a uniform layout (generation_utils.py:484, :520-521), fp32 tensor arithmetic."""
    ts = torch.linspace(1, eps, steps + 1)
    n = torch.tensor(float(masks_per_row), dtype=torch.float32)
    out = []
    for i in range(steps):
        k = int(n * (1 - ts[i + 1] / ts[i])) if i < steps - 1 else int(n)
        out.append(k)
        n = n - k
    return out


# --------------------------------------------------------------------------- think sample (generation_utils.py:367-432)
def _calculate_results(fwd, x, attn4d, mask_token_id: int, temperature: float):
    """generation_utils.py:367-385: forward with arange positions, sample EVERY position, fill the masked ones."""
    mask_index = x == mask_token_id
    tok = torch.ones_like(x).long().cumsum(-1) - 1
    logits = fwd(x, attn4d, tok)
    logits = torch.cat([logits[:, :1], logits[:, :-1]], dim=1)
    probs = torch.softmax(logits, dim=-1)
    if temperature > 0:
        try:
            x0 = categorical_sample(probs.reshape(-1, probs.shape[-1])).view(probs.shape[:-1])
        except ValueError:
            x0 = probs.max(dim=-1)[1]
    else:
        x0 = probs.max(dim=-1)[1]
    x_final = x.clone()
    x_final[mask_index] = x0[mask_index].clone()
    return logits, x_final


def _compare_results(x1, x2, expand_size: int) -> List[int]:
    """generation_utils.py:387-409: per group of `expand_size` rows, the first row with the largest Jaccard
    similarity of token SETS."""
    scores = []
    for i in range(x1.shape[0]):
        a, b = set(x1[i].tolist()), set(x2[i].tolist())
        scores.append(len(a & b) / len(a | b) if len(a | b) > 0 else 0)
    out = []
    for g0 in range(0, len(scores), expand_size):
        grp = scores[g0:g0 + expand_size]
        out.append(g0 + max(range(len(grp)), key=grp.__getitem__))
    return out


def check_and_select(fwd, x, attn4d, expand_size: int, mask_token_id: int, sep_token_id: int,
                     temperature: float = 1.0) -> torch.Tensor:
    """generation_utils.py:411-432.  Returns shifted logits [B, N, V] of the selected copy of every row."""
    x_in = x.repeat_interleave(expand_size, dim=0) if expand_size != 1 else x
    am = attn4d.repeat_interleave(expand_size, dim=0) if expand_size != 1 else attn4d
    logits_select, x_select = _calculate_results(fwd, x_in, am, mask_token_id, temperature)
    sep_pos = int((x_select[0] == sep_token_id).nonzero(as_tuple=True)[0][0])
    x_mask = x_select.clone()
    x_mask[:, sep_pos + 1:-1] = mask_token_id
    _, x_final = _calculate_results(fwd, x_mask, am, mask_token_id, temperature)
    picks = _compare_results(x_in[:, sep_pos + 1:], x_final[:, sep_pos + 1:], expand_size)
    return logits_select[picks]


# --------------------------------------------------------------------------- the loop
@dataclass
class SampleTrace:
    """Per-step internals recorded for parity tests (not part of the reference's API)."""
    k: List[int] = field(default_factory=list)
    confidence: List[torch.Tensor] = field(default_factory=list)
    x0: List[torch.Tensor] = field(default_factory=list)
    mask_logits: List[torch.Tensor] = field(default_factory=list)
    states: List[tuple] = field(default_factory=list)   # expand_loop: (x, window, expand_budget) entering a step
    snaps: List[List[torch.Tensor]] = field(default_factory=list)  # expand_loop: the step's history snapshots


def sample_loop(w, cfg: OracleConfig, input_ids: torch.Tensor, *, max_length: int, steps: int,
                mask_token_id: int, attention_mask: Optional[torch.Tensor] = None, eps: float = 1e-3,
                alg: str = "origin", alg_temp: Optional[float] = None, temperature: float = 0.0,
                top_p: Optional[float] = None, top_k: Optional[int] = None,
                tokens_hook: Callable = lambda step, x, logits: x,
                logits_hook: Callable = lambda step, x, logits: logits,
                noise_fn: Optional[Callable[[int, int, int], torch.Tensor]] = None,
                trace: Optional[SampleTrace] = None, keep_logits: bool = False,
                forward_fn: Optional[Callable] = None, rand_fn: Optional[Callable[[int, int], torch.Tensor]] = None,
                tie_break: str = "torch", think_sample: bool = False, think_sample_interval: int = 2,
                sep_token_id: Optional[int] = None,
                rank_noise_fn: Optional[Callable[[int, int, int], torch.Tensor]] = None
                ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """generation_utils.py:435-548.  Returns (sequences, history) with history[i] = x after step i.

    ``noise_fn(step, M, V)`` injects the exponential noise of the categorical draw
    (default: drawn from torch's global CPU generator exactly where the reference draws it).
    ``forward_fn(x, attn4d, tok_idx) -> logits`` overrides the model (used to time stages).
    ``tie_break``: which of several EXACTLY equal confidences `torch.topk` (:526) transfers is not defined
    by the reference, it is whatever topk does on the device it runs on.  "torch" calls torch.topk here
    (CPU: an nth_element order; this is what the CPU goldens hold).  "index" takes the lowest index among
    equals, which is what ATen's CUDA/HIP topk (radix select, then an in-order gather of the values equal to
    the k-th) returns and what the HIP kernel implements.  The two differ only at exact ties on the k-th value.
    """
    x = F.pad(input_ids, (0, max_length - input_ids.shape[1]), value=mask_token_id)
    if attention_mask is not None and torch.any(attention_mask == 0.0):
        am = F.pad(attention_mask, (0, max_length - attention_mask.shape[1]), value=1.0)
    else:
        am = torch.ones_like(x)
    tok_idx = am.long().cumsum(-1) - 1
    tok_idx.masked_fill_(am == 0, 1)
    attn4d = torch.logical_and(am.unsqueeze(1).unsqueeze(-2), am.unsqueeze(1).unsqueeze(-1))
    timesteps = torch.linspace(1, eps, steps + 1)
    history: List[torch.Tensor] = []
    fwd = forward_fn or (lambda xx, m, p: forward_logits(w, cfg, xx, m, p))

    x = tokens_hook(None, x, None)
    for i in range(steps):
        mask_index = x == mask_token_id
        if think_sample and i % think_sample_interval == 0:                      # :490-491
            logits = check_and_select(fwd, x, attn4d, 4, mask_token_id, sep_token_id)
        else:
            logits = fwd(x, attn4d, tok_idx)
            logits = torch.cat([logits[:, :1], logits[:, :-1]], dim=1)
        logits = logits_hook(i, x, logits)
        mask_logits = logits[mask_index]
        t, s = timesteps[i], timesteps[i + 1]
        if alg == "origin":
            p_transfer = 1 - s / t if i < steps - 1 else 1
            x0 = torch.zeros_like(x[mask_index], dtype=torch.long) + mask_token_id
            transfer = (torch.rand(*x0.shape) if rand_fn is None else rand_fn(i, x0.shape[0])) < p_transfer
            sel = mask_logits[transfer]
            nz = noise_fn(i, sel.shape[0], sel.shape[1]) if (noise_fn and temperature > 0) else None
            _, x0[transfer] = sample_tokens(sel, temperature=temperature, top_p=top_p, top_k=top_k, noise=nz)
            x[mask_index] = x0.clone()
        else:
            nz = noise_fn(i, mask_logits.shape[0], mask_logits.shape[1]) if (noise_fn and temperature > 0) else None
            if alg == "maskgit_plus":
                confidence, x0 = sample_tokens(mask_logits, temperature=temperature, top_p=top_p, top_k=top_k, noise=nz)
            elif alg == "topk_margin":
                confidence, x0 = sample_tokens(mask_logits, temperature=temperature, top_p=top_p, top_k=top_k,
                                               margin_confidence=True, noise=nz)
            elif alg == "entropy":
                confidence, x0 = sample_tokens(mask_logits, temperature, top_p=top_p, top_k=top_k,
                                               neg_entropy=True, noise=nz)
            else:
                raise RuntimeError(f"Unknown alg: {alg}")
            num_mask_token = mask_index.sum() / mask_index.shape[0]
            k = int(num_mask_token * (1 - s / t)) if i < steps - 1 else int(num_mask_token)
            if trace is not None:
                trace.k.append(k)
                trace.confidence.append(confidence.clone())
                trace.x0.append(x0.clone())
                if keep_logits:
                    trace.mask_logits.append(mask_logits.clone())
            full_conf = torch.full_like(x, -torch.inf, dtype=logits.dtype)
            full_conf[mask_index] = confidence
            if k > 0:
                if alg_temp is None or alg_temp == 0:
                    if tie_break == "index":
                        transfer_index = torch.sort(full_conf, dim=1, descending=True, stable=True).indices[:, :k]
                    else:
                        _, transfer_index = torch.topk(full_conf, k)
                else:
                    full_conf = F.softmax(full_conf / alg_temp, dim=-1)
                    # torch.multinomial(p, k) without replacement IS topk(p / q, k) with q = one exponential_(1) fill
                    # of p's shape from the same generator (ATen Distributions.cpp, multinomial fast path); written
                    # out so that the draws can be injected (`rank_noise_fn(step, B, N)`).  Checked against the
                    # reference golden `ded_algtemp`, which the reference made with torch.multinomial itself.
                    q = (rank_noise_fn(i, *full_conf.shape) if rank_noise_fn is not None
                         else torch.empty_like(full_conf).exponential_(1))
                    transfer_index = torch.topk(full_conf / q, k).indices
                x_ = torch.zeros_like(x, dtype=torch.long) + mask_token_id
                x_[mask_index] = x0.clone()
                rows = torch.arange(x.size(0)).unsqueeze(1).expand_as(transfer_index)
                x[rows, transfer_index] = x_[rows, transfer_index]
        x = tokens_hook(i, x, logits)
        history.append(x.clone())
    return x, history


# --------------------------------------------------------------------------- variable-length loop
def expand_loop(w, cfg: OracleConfig, input_ids: torch.Tensor, *, max_length: int, mask_token_id: int,
                eos_token_id: int, delete_token_id: int, expand_token_id: int,
                expand_budget: Optional[int] = None, pad_delete_to_right: bool = False,
                number_transfer_tokens: int = 1, alg: str = "entropy", alg_temp: Optional[float] = None,
                temperature: float = 0.0, top_p: Optional[float] = None, top_k: Optional[int] = None,
                noise_fn: Optional[Callable[[int, int, int], torch.Tensor]] = None,
                forward_fn: Optional[Callable] = None, tie_break: str = "torch",
                trace: Optional[SampleTrace] = None,
                rank_noise_fn: Optional[Callable[[int, int], torch.Tensor]] = None
                ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """``generation_expand_utils.py:294-356`` (prefix / mask bookkeeping) + ``_sample`` ``:358-503``: the
    DreamOn-style variable-length sampler.  ``input_ids`` is ``[1, L]`` (the path inspects row 0 only, :445-495).
    Returns (x ``[1, max_length]``, history) where history holds the up to three snapshots per step the reference
    appends (after the commit :444-445, after deletions :474-476, after expansions :493-495).

    ``noise_fn(step, M, V)`` injects the Exp(1) draws of the categorical sample; ``tie_break`` as in sample_loop.
    """
    assert input_ids.dim() == 2 and input_ids.shape[0] == 1
    mask_pos = torch.where(input_ids == mask_token_id)[1]
    if mask_pos.numel() == 0:
        raise ValueError("No mask tokens found in the input_ids.")                          # :318-319
    start_gen_len = int(mask_pos.numel())
    prefix_len = int(mask_pos[0])
    if input_ids.shape[1] > max_length:                                                        # :177-183
        raise ValueError(f"Input length of input_ids is {input_ids.shape[1]}, but `max_length` is set to {max_length}.")
    num_generation_tokens = start_gen_len
    max_gen_len = max_length - prefix_len
    if expand_budget is None:
        expand_budget = max_gen_len * 2
    fwd = forward_fn or (lambda xx, m, p: forward_logits(w, cfg, xx, m, p))
    history: List[torch.Tensor] = []
    L_in = input_ids.shape[1]

    x = F.pad(input_ids, (0, max_length - L_in), value=eos_token_id)
    for i in range(2 * max_gen_len + 2 * expand_budget):
        cwl = L_in - start_gen_len + num_generation_tokens
        am = torch.ones(1, cwl, dtype=torch.int16)
        am = F.pad(am, (0, max_length - cwl), value=0)          # negative pad crops, as in the reference (:401)
        mask_index = (x == mask_token_id) & (am == 1)
        if torch.all(~mask_index[:, :cwl]):
            break
        tok_idx = am.long().cumsum(-1) - 1
        tok_idx.masked_fill_(am == 0, 1)
        attn4d = torch.logical_and(am.unsqueeze(1).unsqueeze(-2), am.unsqueeze(1).unsqueeze(-1))
        logits = fwd(x, attn4d, tok_idx)
        logits = torch.cat([logits[:, :1], logits[:, :-1]], dim=1)
        logits = logits[mask_index]
        if cwl == max_length or expand_budget == 0:              # :421-422
            logits[:, expand_token_id] -= 1e9
        nz = noise_fn(i, logits.shape[0], logits.shape[1]) if (noise_fn and temperature > 0) else None
        if alg == "maskgit_plus":
            confidence, x0 = sample_tokens(logits, temperature=temperature, top_p=top_p, top_k=top_k, noise=nz)
        elif alg == "topk_margin":
            confidence, x0 = sample_tokens(logits, temperature=temperature, top_p=top_p, top_k=top_k,
                                           margin_confidence=True, noise=nz)
        elif alg == "entropy":
            confidence, x0 = sample_tokens(logits, temperature=temperature, top_p=top_p, top_k=top_k,
                                           neg_entropy=True, noise=nz)
        else:
            raise RuntimeError(f"Unknown alg: {alg}")
        if trace is not None:
            trace.k.append(number_transfer_tokens)
            trace.confidence.append(confidence.clone())
            trace.x0.append(x0.clone())
            trace.states.append((x.clone(), cwl, expand_budget))
        x, d_gen, n_exp, snaps = expand_commit_edit(
            x, mask_index, confidence, x0, number_transfer_tokens=number_transfer_tokens, alg_temp=alg_temp,
            tie_break=tie_break, mask_token_id=mask_token_id, delete_token_id=delete_token_id,
            expand_token_id=expand_token_id, pad_delete_to_right=pad_delete_to_right, max_length=max_length,
            window=cwl,
            rank_noise=(rank_noise_fn(i, confidence.numel()) if (rank_noise_fn is not None and alg_temp) else None))
        num_generation_tokens += d_gen
        expand_budget -= n_exp
        history.extend(snaps)
        if trace is not None:
            trace.snaps.append(snaps)
    return x[:, :max_length], history


def expand_commit_edit(x, mask_index, confidence, x0, *, number_transfer_tokens: int, alg_temp, tie_break: str,
                       mask_token_id: int, delete_token_id: int, expand_token_id: int, pad_delete_to_right: bool,
                       max_length: int, window: int, rank_noise: Optional[torch.Tensor] = None):
    """One iteration's commit and sequence edits, generation_expand_utils.py:434-495, on `x` [1, max_length].
    Returns (x, change of num_generation_tokens, number of expansions, history snapshots)."""
    snaps: List[torch.Tensor] = []
    if alg_temp is None or alg_temp == 0:
        if tie_break == "index":
            if number_transfer_tokens > confidence.numel():
                raise RuntimeError("selected index k out of range")
            transfer_index = torch.sort(confidence, descending=True, stable=True).indices[:number_transfer_tokens]
        else:
            _, transfer_index = torch.topk(confidence, number_transfer_tokens)
    else:
        confidence = F.softmax(confidence / alg_temp, dim=-1)
        if number_transfer_tokens > confidence.numel():
            raise RuntimeError("cannot sample n_sample > prob_dist.size(-1) samples without replacement")
        # torch.multinomial(p, k) without replacement == topk(p / q, k), q = one exponential_(1) fill (see sample_loop)
        q = rank_noise if rank_noise is not None else torch.empty_like(confidence).exponential_(1)
        transfer_index = torch.topk(confidence / q, number_transfer_tokens).indices
    x0_ = torch.zeros_like(x0, dtype=torch.long) + mask_token_id
    x0_[transfer_index] = x0[transfer_index].clone()
    x[mask_index] = x0_
    snaps.append(x[0, :window].clone())
    d_gen = 0

    # ---- delete (:447-476)
    if pad_delete_to_right:
        row = x[0]
        hits = (row == delete_token_id).nonzero(as_tuple=True)[0]
        if hits.numel() > 0:
            first = int(hits[0])
            fill = (torch.arange(row.size(0)) >= first) & mask_index[0]
            row.masked_fill_(fill, delete_token_id)
            x = row.unsqueeze(0)
    del_idx = ((x[0] == delete_token_id) & mask_index[0]).nonzero(as_tuple=False).squeeze(1)
    if del_idx.numel() > 0:
        for idx in sorted(del_idx.tolist(), reverse=True):
            x = torch.cat((x[:, :idx], x[:, idx + 1:], torch.tensor([[mask_token_id]])), dim=1)
            d_gen -= 1
        snaps.append(x[0, :max(window + d_gen, 0)].clone())
    # ---- expand (:477-495)
    n_exp = 0
    exp_idx = (x[0] == expand_token_id).nonzero(as_tuple=False).squeeze(1)
    if exp_idx.numel() > 0:
        for idx in sorted(exp_idx.tolist(), reverse=True):
            x = torch.cat((x[:, :idx], torch.tensor([[mask_token_id, mask_token_id]]), x[:, idx + 1:]), dim=1)
            d_gen += 1
            n_exp += 1
            if x.shape[1] > max_length:
                x = x[:, :max_length]
        snaps.append(x[0, :max(window + d_gen, 0)].clone())
    return x, d_gen, n_exp, snaps


# --------------------------------------------------------------------------- helpers shared by tests / bench
def weight_names(cfg: OracleConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict names and shapes of ``DreamModel_length``, in state_dict order."""
    d, ff, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    names: List[Tuple[str, Tuple[int, ...]]] = [("model.embed_tokens.weight", (V, d))]
    for i in range(cfg.num_hidden_layers):
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
    names += [("model.norm.weight", (d,)), ("lm_head.weight", (V, d))]
    return names
