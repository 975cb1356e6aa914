"""Device-side layout build and SEP-split decode around the denoise loop.

Mirrors ``test_unify_extract_sample_to_device_length`` (akgr/tokenizer.py:287-338) from the point where the
tokenizer has produced its padded id matrices, and the SEP split + special-token strip that
``main.test_batch`` (akgr/abduction_model/main.py:409-421) runs on the result; the string <-> id steps stay the
tokenizer's.  With these two the ids never leave the device between the tokenizer and the scorer.
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch

from . import _hip

MODES = {"unify": 0, "sft_abd": 1, "sft_ded": 2}


@_hip.device_guard(lambda query_ids, *a, **k: query_ids.device)
def build_inputs(query_ids: torch.Tensor, observation_ids: torch.Tensor, model_type: str, bos_token_id: int,
                 sep_token_id: int, eos_token_id: int, mask_token_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(input_ids [B, Q+A+3] int64, need_to_mask [B, Q+A+3] float32) as akgr/tokenizer.py:318-338 returns them."""
    if model_type not in MODES:
        raise ValueError(f"unknown model_type: {model_type}")
    if not query_ids.is_cuda:
        raise RuntimeError("dark_amd runs on a HIP device only")
    q = query_ids.to(torch.int64).contiguous()
    o = observation_ids.to(device=q.device, dtype=torch.int64).contiguous()
    B, Q = q.shape
    A = o.shape[1]
    ids = torch.empty((B, Q + A + 3), dtype=torch.int64, device=q.device)
    need = torch.empty((B, Q + A + 3), dtype=torch.float32, device=q.device)
    _hip.check(_hip.load().dark_build_layout(_hip.ptr(q), _hip.ptr(o), B, Q, A, MODES[model_type], int(bos_token_id),
                                             int(sep_token_id), int(eos_token_id), int(mask_token_id), _hip.ptr(ids),
                                             _hip.ptr(need), _hip.stream_ptr(q.device)))
    return ids, need


@_hip.device_guard(lambda test_ids, *a, **k: test_ids.device)
def split_strip(test_ids: torch.Tensor, sep_token_id: int, special_ids: Iterable[int]):
    """main.py:409-421 at the id level.  Returns (sep_pos, left [B,N], left_len [B], right [B,N], right_len [B]):
    the kept ids of `test_ids[:, :sep_pos]` / `test_ids[:, sep_pos+1:]`, padded with -1."""
    special_lo = 0
    for s in special_ids:
        if not 0 <= int(s) < 64:
            raise NotImplementedError(f"special id {s} outside 0..63")
        special_lo |= 1 << int(s)
    x = test_ids.to(torch.int64).contiguous()
    B, N = x.shape
    dev = x.device
    left = torch.empty((B, N), dtype=torch.int64, device=dev)
    right = torch.empty((B, N), dtype=torch.int64, device=dev)
    ll = torch.empty(B, dtype=torch.int32, device=dev)
    rl = torch.empty(B, dtype=torch.int32, device=dev)
    sp = torch.empty(1, dtype=torch.int32, device=dev)
    _hip.check(_hip.load().dark_split_strip(_hip.ptr(x), B, N, int(sep_token_id), special_lo, _hip.ptr(left),
                                            _hip.ptr(ll), _hip.ptr(right), _hip.ptr(rl), _hip.ptr(sp),
                                            _hip.stream_ptr(dev)))
    sep_pos = int(sp.item())
    if sep_pos < 0:
        raise IndexError("index 0 is out of bounds for dimension 0 with size 0")   # `sep_positions[0]`, main.py:414
    return sep_pos, left, ll, right, rl


def to_lists(ids: torch.Tensor, lens: torch.Tensor) -> List[List[int]]:
    rows, n = ids.cpu().tolist(), lens.cpu().tolist()
    return [r[:k] for r, k in zip(rows, n)]
