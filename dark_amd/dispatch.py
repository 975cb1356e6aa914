"""Multi-GPU batch dispatcher for the denoise path: one process per GPU, rows sharded, no exchange
inside the loop.

Sequences are independent (attention is within a row, sampling per masked row, top-k per row), so a batch
is partitioned into contiguous row blocks, one per rank, with replicated weights.  The reference does the
same for its legacy multi-GPU evaluation (``accelerator.prepare(model, dataloader)`` shards the batches,
``mdm_unify.py:328-331``) and gathers only results: pickled per-sample score dicts with
``dist.all_gather_object`` and ``accelerator.gather(pattern_id)`` (``mdm_unify.py:434-438``).  Here the
gather is ONE ``all_gather`` of fixed-width records (RCCL over xGMI when the backend is ``nccl``; ``gloo``
in the CPU tests) -- no pickling, no per-sample objects.

The single global coupling of the loop is the transfer count ``k``, computed from the BATCH-average mask
count (``generation_utils.py:520-521``).  For the fixed layouts every row has the same number of masks and
shards agree by construction; for ragged inputs ``global_mask_count`` all-reduces the count (8 bytes per
step) so that a sharded run takes the same ``k`` as an unsharded one.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


# Tests and `bench.py --force-dist` set this to run every collective at world size 1 as well: a single rank on one
# card is the only way to take the RCCL (`nccl`) branches on a 1-GPU box (RCCL refuses two ranks on one device).
FORCE_COLLECTIVES = False


def _single(world: int) -> bool:
    return world == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: sizes differ by at most one, larger blocks first."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} / world {world}")
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _stage(t: torch.Tensor, group) -> torch.Tensor:
    """The tensor a collective can take: device memory for `nccl` (= RCCL, peer writes over xGMI), host memory
    for `gloo` (the CPU tests and single-GPU rehearsals, where several ranks share one card)."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return t.cpu()
    return t


def gather_rows(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """all_gather of row blocks produced by `shard_bounds` -> the `[total, ...]` tensor on every rank.
    Blocks are padded to the largest block so that one fixed-size collective moves everything."""
    world, rank = _world(group)
    if _single(world):
        return local
    lo, hi = shard_bounds(total, world, rank)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, shard is {hi - lo}")
    width = -(-total // world)
    device = local.device
    local = _stage(local, group)
    buf = local.new_zeros((width,) + tuple(local.shape[1:]))
    buf[: hi - lo] = local
    out = local.new_empty((world * width,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    out = out.to(device)
    parts = []
    for r in range(world):
        a, b = shard_bounds(total, world, r)
        parts.append(out[r * width: r * width + (b - a)])
    return torch.cat(parts, dim=0)


def gather_metric_records(scores: torch.Tensor, pattern_id: torch.Tensor, total: int, group=None
                          ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fixed-width replacement for ``all_gather_object(scores)`` + ``gather(pattern_id)``
    (mdm_unify.py:434-438): scores float32 ``[B_local, F]`` (F <= 8 metric columns: smatch, jaccard, ...),
    pattern_id int32 ``[B_local]``; both travel in one float32 record of width F+1."""
    if scores.dim() != 2 or pattern_id.shape != scores.shape[:1]:
        raise ValueError("scores must be [B_local, F] and pattern_id [B_local]")
    if pattern_id.numel() and int(pattern_id.abs().max()) >= 1 << 24:
        raise ValueError("pattern_id does not fit a float32 record exactly")
    rec = torch.cat([scores.float(), pattern_id.float().unsqueeze(1)], dim=1)
    full = gather_rows(rec, total, group)
    return full[:, :-1].contiguous(), full[:, -1].round().to(torch.int32)


def global_mask_count(local_count: torch.Tensor, local_batch: int, group=None) -> Tuple[int, int]:
    """(sum of masked positions, sum of batch rows) over the group: the operands of the batch-average
    mask count of generation_utils.py:520.  `local_count` is a one-element integer tensor on the device the
    backend communicates from."""
    world, _ = _world(group)
    if _single(world):
        return int(local_count), local_batch
    t = torch.stack([local_count.reshape(()).to(torch.int64),
                     torch.tensor(local_batch, dtype=torch.int64, device=local_count.device)])
    t = _stage(t, group)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t[0]), int(t[1])


@torch.no_grad()
def generate_sharded(model, inputs: torch.Tensor, generation_config=None, *, group=None, gather: bool = True,
                     exact_ragged_k: bool = False, attention_mask: Optional[torch.Tensor] = None,
                     shard_head: bool = False, **kwargs) -> torch.Tensor:
    """``model.diffusion_generate`` over this rank's row block of `inputs` (same tensor on every rank),
    then, if `gather`, an all_gather of the `[B_local, N]` sequences so every rank returns `[B, N]`.

    RNG: each rank draws from its own device generator, so stochastic parity (temperature > 0) is defined per
    shard: same seed + same rows => same result as an unsharded call on those rows alone.
    `exact_ragged_k`: all-reduce the mask count every step (only matters when rows carry different numbers of
    masks).
    `shard_head`: BASELINE config 3 -- split the LM head's vocabulary columns over the ranks of the group
    (vocab_shard.ShardedHead: all-gather of hidden rows, all-reduce of row maxima, all-to-all of per-shard candidate
    records); results are bit-identical to the unsharded head.  Every rank must hold at least one row."""
    world, rank = _world(group)
    total = inputs.shape[0]
    lo, hi = shard_bounds(total, world, rank)
    kwargs = dict(kwargs)
    kwargs["return_dict_in_generate"] = False
    if attention_mask is not None:
        kwargs["attention_mask"] = attention_mask[lo:hi]
    if exact_ragged_k and not _single(world):
        if total < world:
            raise ValueError("exact_ragged_k needs at least one row per rank (every rank joins the per-step all-reduce)")
        kwargs["_count_reduce"] = lambda cnt, b: global_mask_count(cnt, b, group)
    if shard_head and not _single(world):
        if total < world:
            raise ValueError("shard_head needs at least one row per rank (every rank joins the per-step exchanges)")
        from .vocab_shard import ShardedHead
        kwargs["_sharded_head"] = ShardedHead(model, group)
    if hi > lo:
        local = model.diffusion_generate(inputs[lo:hi], generation_config=generation_config, **kwargs)
    else:
        n = kwargs.get("max_length") or (generation_config.max_length if generation_config else inputs.shape[1] + 1)
        local = inputs.new_empty((0, n))
    if not gather:
        return local
    return gather_rows(local, total, group)


@torch.no_grad()
def generate_batch_sharded(model, prompts, generation_config=None, *, group=None, gather: bool = True,
                           **kwargs) -> torch.Tensor:
    """Variable-length path: ``model.diffusion_generate_batch`` over this rank's contiguous block of the
    ragged `prompts` list (same list on every rank), then an all_gather of the `[B_local, max_length]` sequences.
    Sequences are independent, so nothing is exchanged inside the loop; ranks simply finish at different steps."""
    world, rank = _world(group)
    total = len(prompts)
    lo, hi = shard_bounds(total, world, rank)
    kwargs = dict(kwargs)
    kwargs["return_dict_in_generate"] = False
    n = kwargs.get("max_length") or generation_config.max_length
    if hi > lo:
        local = model.diffusion_generate_batch(list(prompts[lo:hi]), generation_config=generation_config, **kwargs)
    else:
        local = prompts[0].new_empty((0, n))
    if not gather:
        return local
    return gather_rows(local, total, group)


@torch.no_grad()
def per_token_logps_sharded(model, input_ids: torch.Tensor, mask_attention: torch.Tensor, mask_seeds,
                            mask_token_id: int, *, group=None, gather: bool = True, **kwargs) -> torch.Tensor:
    """GRPO scoring with the completions `[I, B, N]` split along B: every rank scores its block with
    ``grpo_logps.get_per_token_logps`` and the `[B_local, I, N]` results are all-gathered.  The perturbation is
    drawn per rank (seed + local rows), as the reference's per-process trainers do."""
    from . import grpo_logps
    world, rank = _world(group)
    total = input_ids.shape[1]
    lo, hi = shard_bounds(total, world, rank)
    if hi > lo:
        local = grpo_logps.get_per_token_logps(model, input_ids[:, lo:hi], mask_attention[:, lo:hi], mask_seeds,
                                               mask_token_id, **kwargs)
    else:
        local = torch.empty((0, input_ids.shape[0], input_ids.shape[2]), dtype=torch.float32,
                            device=input_ids.device)
    if not gather:
        return local
    return gather_rows(local.contiguous(), total, group)
