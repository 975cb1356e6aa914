"""CPU oracle (numpy) for the host pre / post steps around the denoise loop.

TEST INFRASTRUCTURE ONLY (same rule as oracle/dream_oracle.py).

Restates ``test_unify_extract_sample_to_device_length`` after tokenisation (akgr/tokenizer.py:318-335),
``test_unify_extract_sample_to_device`` (:160-182) and the SEP split + ``decode(skip_special_tokens=True)`` of
``main.test_batch`` (akgr/abduction_model/main.py:409-421) at the id level.  Pinned by fixtures produced by
executing the reference's own definitions and tokenizer (tests/golden/make_goldens_layout.py).
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import numpy as np

MODES = {"unify": 0, "sft_abd": 1, "sft_ded": 2}


def build_layout(query_ids: np.ndarray, observation_ids: np.ndarray, model_type: str, bos: int, sep: int, eos: int,
                 mask: int) -> Tuple[np.ndarray, np.ndarray]:
    """akgr/tokenizer.py:318-335: (input_ids [B, Q+A+3] int64, attention_mask [B, Q+A+3] float32)."""
    B, Q = query_ids.shape
    A = observation_ids.shape[1]
    col = lambda v: np.full((B, 1), v, dtype=np.int64)  # noqa: E731
    z1 = np.zeros((B, 1), dtype=np.float32)
    if model_type == "unify":
        ids = np.concatenate([col(bos), query_ids, col(sep), observation_ids, col(eos)], 1)
        am = np.concatenate([z1, np.ones((B, Q), np.float32), z1, np.ones((B, A), np.float32), z1], 1)
    elif model_type == "sft_abd":
        ids = np.concatenate([col(bos), np.full((B, Q), mask, np.int64), col(sep), observation_ids, col(eos)], 1)
        am = np.concatenate([z1, np.ones((B, Q), np.float32), z1, np.zeros((B, A), np.float32), z1], 1)
    elif model_type == "sft_ded":
        ids = np.concatenate([col(bos), query_ids, col(sep), np.full((B, A), mask, np.int64), col(eos)], 1)
        am = np.concatenate([z1, np.zeros((B, Q), np.float32), z1, np.ones((B, A), np.float32), z1], 1)
    else:
        raise ValueError(model_type)
    return ids, am


def build_single(side_ids: np.ndarray, mask_number: int, model_type: str, bos: int, sep: int, eos: int,
                 mask: int) -> np.ndarray:
    """akgr/tokenizer.py:160-182: un-padded [1, L] prompt of the variable-length path."""
    one = lambda v: np.full((1, 1), v, dtype=np.int64)  # noqa: E731
    masks = np.full((1, mask_number), mask, dtype=np.int64)
    if model_type == "sft_ded":
        return np.concatenate([one(bos), side_ids, one(sep), masks, one(eos)], 1)
    if model_type == "sft_abd":
        return np.concatenate([one(bos), masks, one(sep), side_ids, one(eos)], 1)
    raise ValueError(model_type)


def split_strip(test_ids: np.ndarray, sep_id: int, special_ids: Iterable[int]):
    """main.py:409-421 at the id level: (sep_pos, left id lists, right id lists)."""
    hits = np.nonzero(test_ids[0] == sep_id)[0]
    sep_pos = int(hits[0])                       # IndexError when row 0 has no SEP, as in the reference
    special = set(int(s) for s in special_ids)
    strip = lambda part: [int(t) for t in part if int(t) not in special]  # noqa: E731
    left: List[List[int]] = [strip(r[:sep_pos]) for r in test_ids]
    right: List[List[int]] = [strip(r[sep_pos + 1:]) for r in test_ids]
    return sep_pos, left, right
