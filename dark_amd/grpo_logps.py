"""Forward-only log-probabilities of the diffusion GRPO trainer on the HIP path.

Mirrors, with the same names and argument meaning, the pieces of
``akgr/abduction_model/custom_grpo2.py`` that the trainer runs under ``torch.no_grad()`` to score completions with
the old / reference policy (:681-709): ``selective_log_softmax`` (:89-161) and the ``DiffuGRPOTrainer`` methods
``forward_process`` (:350-387), ``get_logits`` (:409-418), ``_get_per_token_logps`` (:421-498).
``DiffuGRPOLogpsMixin`` carries the three methods so that a trainer class can take them over for those calls;
the differentiable call inside ``compute_loss`` (:228) needs autograd and stays with the reference model.

What runs underneath: one ``dark_forward_hidden`` over the ``3 * num_iterations * batch`` perturbed sequences, then
``dark_lm_head_logprob`` -- final norm + LM head + log_softmax + gather without ever writing the ``[rows, V]``
logits for bf16 models -- and ``dark_grpo_combine``.  torch supplies memory, the stream and the random draws
(``random.uniform`` / ``torch.rand`` are consumed exactly where the reference consumes them).
"""
from __future__ import annotations

import random
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import _hip


def set_seed(seed: Optional[int]) -> None:
    """What transformers.set_seed does (custom_grpo2.py:351)."""
    if seed is None:
        return
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _u8(t: torch.Tensor) -> torch.Tensor:
    return (t != 0).to(torch.uint8).contiguous()


@_hip.device_guard(lambda batch, *a, **k: batch.device)
def forward_process(batch: torch.Tensor, mask_target: torch.Tensor, mask_id: int, seed: Optional[int] = None,
                    accumulate: bool = False, gradient_accumulation_steps: int = 1,
                    _rand_fn: Optional[Callable[[int, int], torch.Tensor]] = None):
    """custom_grpo2.py:350-387 on the device: returns ([3 x [b, l] int64], [1, 1/t, 1/(1-t)], completion_mask)."""
    if not batch.is_cuda:
        raise RuntimeError("dark_amd runs on a HIP device only")
    lib = _hip.load()
    set_seed(seed)
    b, l = batch.shape
    mask_ratio = random.uniform(0.2, 0.8)
    draw = _rand_fn or (lambda rows, cols: torch.rand((rows, cols), device=batch.device))
    if accumulate:
        r = draw(b // gradient_accumulation_steps, l).to(batch.device)
        r = torch.cat([r] * gradient_accumulation_steps, dim=0)
    else:
        r = draw(b, l).to(batch.device)
    if r.shape != (b, l):
        raise ValueError(f"random matrix {tuple(r.shape)} does not cover the batch {(b, l)}")
    r = r.to(torch.float32).contiguous()
    x = batch.to(torch.int64).contiguous()
    tgt = _u8(mask_target)
    noisy = torch.empty((3, b, l), dtype=torch.int64, device=batch.device)
    cmask = torch.empty((b, l), dtype=torch.uint8, device=batch.device)
    # the comparison runs in fp32 against float32(mask_ratio), as `random_matrix < t_p` does with t_p = ones * ratio
    _hip.check(lib.dark_grpo_perturb(_hip.ptr(x), _hip.ptr(tgt), _hip.ptr(r), float(np.float32(mask_ratio)),
                                     int(mask_id), b, l, _hip.ptr(noisy), _hip.ptr(cmask),
                                     _hip.stream_ptr(batch.device)))
    return [noisy[0], noisy[1], noisy[2]], [1, 1 / mask_ratio, 1 / (1 - mask_ratio)], cmask.bool()


def _shift_rows(model, S: int, N: int) -> torch.Tensor:
    src = torch.empty(S * N, dtype=torch.int32, device=model.device)
    _hip.check(_hip.load().dark_shift_rows(_hip.ptr(src), S, N, _hip.stream_ptr(model.device)))
    return src


@_hip.device_guard(lambda model, *a, **k: model.device)
def get_logits(model, batch: torch.Tensor) -> torch.Tensor:
    """custom_grpo2.py:409-418: `model(batch).logits` shifted right by one position, fp32 [S, N, V] (materialised:
    use `get_per_token_logps` to avoid that)."""
    x = batch.to(model.device).to(torch.int64).contiguous()
    S, N = x.shape
    st = model._hip_state(S, N)
    st.hidden_valid = False
    st.forward_hidden(x, None, None)
    return st.head_rows(_shift_rows(model, S, N), S * N).view(S, N, model.config.vocab_size).clone()


@_hip.device_guard(lambda logits, *a, **k: logits.device)
def selective_log_softmax(logits: torch.Tensor, index: torch.Tensor, weights: Optional[torch.Tensor] = None,
                          mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """custom_grpo2.py:89-161 on materialised logits [I*3*B, N, V] (any float dtype; computed in fp32)."""
    lib = _hip.load()
    dev = logits.device
    R, N, V = logits.shape
    full = R // 3
    iters = weights.size(0) // 3
    B = full // iters
    lg = logits.to(torch.float32).contiguous()
    targets = index.view(iters, 1, B, N).expand(iters, 3, B, N).contiguous().to(torch.int64)
    lp = torch.empty(R * N, dtype=torch.float32, device=dev)
    _hip.check(lib.dark_logprob_from_logits(_hip.ptr(lg), V, _hip.ptr(targets), R * N, V, _hip.ptr(lp),
                                            _hip.stream_ptr(dev)))
    if weights is None or mask is None:
        return lp.view(iters, 3, B, N)[:, 0].reshape(full, N)
    out = torch.empty((full, N), dtype=torch.float32, device=dev)
    w = weights.to(device=dev, dtype=torch.float32).contiguous()
    _hip.check(lib.dark_grpo_combine(_hip.ptr(lp), _hip.ptr(w), _hip.ptr(_u8(mask)), iters, B, N, _hip.ptr(out),
                                     _hip.stream_ptr(dev)))
    return out


@_hip.device_guard(lambda model, *a, **k: model.device)
def get_per_token_logps(model, input_ids: torch.Tensor, mask_attention: torch.Tensor, mask_seeds: Sequence[int],
                        mask_token_id: int, gradient_accumulation_steps: int = 1,
                        _rand_fn: Optional[Callable[[int, int, int], torch.Tensor]] = None) -> torch.Tensor:
    """custom_grpo2.py:421-498: input_ids / mask_attention [num_iterations, batch, seq]; returns fp32
    [batch, num_iterations, seq]."""
    if input_ids.dim() != 3:
        raise ValueError(f"Expected input_ids to have 3 dimensions, got {input_ids.dim()}")
    iters, B, N = input_ids.size()
    if len(mask_seeds) != iters:
        raise ValueError(f"Expected mask_seeds length to be {iters}, got {len(mask_seeds)}")
    dev = model.device
    if dev.type != "cuda":
        raise RuntimeError("dark_amd runs on a HIP device only: move the model with .to('cuda')")
    lib = _hip.load()
    input_ids = input_ids.to(dev).to(torch.int64).contiguous()
    mask_attention = mask_attention.to(dev)
    seqs: List[torch.Tensor] = []
    weights: List[float] = []
    masks: List[torch.Tensor] = []
    for i, seed in enumerate(mask_seeds):
        noisy, tw, cm = forward_process(input_ids[i], mask_attention[i], mask_token_id, seed=seed,
                                        accumulate=iters > 1, gradient_accumulation_steps=gradient_accumulation_steps,
                                        _rand_fn=(lambda r, c, i=i: _rand_fn(i, r, c)) if _rand_fn else None)
        seqs.extend(noisy)
        weights.extend(tw)
        masks.append(cm)
    x = torch.cat(seqs, dim=0).contiguous()                       # [I*3*B, N] ordered (iteration, version, row)
    S = x.shape[0]
    st = model._hip_state(S, N)
    st.hidden_valid = False
    st.forward_hidden(x, None, None)
    pk = model.packed()
    M = S * N
    src = _shift_rows(model, S, N)
    targets = input_ids.view(iters, 1, B, N).expand(iters, 3, B, N).contiguous().view(-1)
    ws_bytes = lib.dark_logprob_workspace_bytes(pk.ref, M)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    lp = torch.empty(M, dtype=torch.float32, device=dev)
    _hip.check(lib.dark_lm_head_logprob(pk.ref, _hip.ptr(st.hidden), _hip.ptr(src), M, _hip.ptr(targets), M,
                                        _hip.ptr(ws), ws_bytes, _hip.ptr(lp), _hip.stream_ptr(dev)))
    out = torch.empty((iters * B, N), dtype=torch.float32, device=dev)
    w = torch.tensor(weights, dtype=torch.float32, device=dev)
    cmask = torch.cat(masks, dim=0).to(torch.uint8).contiguous()
    _hip.check(lib.dark_grpo_combine(_hip.ptr(lp), _hip.ptr(w), _hip.ptr(cmask), iters, B, N, _hip.ptr(out),
                                     _hip.stream_ptr(dev)))
    return out.view(iters, B, N).permute(1, 0, 2).to(torch.float32)


class DiffuGRPOLogpsMixin:
    """The three trainer methods under their reference names; expects `self.args.gradient_accumulation_steps` and
    `self.processing_class.mask_token_id` as the reference trainer has them."""

    def forward_process(self, batch, mask_target, mask_id, seed=None, accumulate=False):
        return forward_process(batch, mask_target, mask_id, seed=seed, accumulate=accumulate,
                               gradient_accumulation_steps=self.args.gradient_accumulation_steps)

    def get_logits(self, model, batch):
        return get_logits(model, batch)

    def _get_per_token_logps(self, model, input_ids, mask_attention, mask_seeds):
        return get_per_token_logps(model, input_ids, mask_attention, mask_seeds,
                                   self.processing_class.mask_token_id,
                                   gradient_accumulation_steps=self.args.gradient_accumulation_steps)
