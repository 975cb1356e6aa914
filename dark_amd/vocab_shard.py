"""Vocabulary-sharded LM head + sampler (BASELINE.json config 3: "vocab-sharded logit head + RCCL top-k gather").

The reference has one head: ``lm_head(hidden)`` over the whole vocabulary (``modeling_dream_length.py:809``) followed by
``sample_tokens`` (``generation_utils.py:58-90``).  Here the head's columns are split over the S ranks of a process
group, one process per GPU; rank s only ever forms logits for its own ``[v0, v0 + Vs)`` columns.  Per denoise step the
ranks exchange (RCCL over xGMI with the ``nccl`` backend; ``gloo`` in single-card rehearsals):

1. ``all_gather`` of the normalised hidden rows of every rank's masked positions, ``[M_r, d]`` model dtype;
2. ``all_reduce(MAX)`` of the per-row maxima of ``logits / T``;
3. ``all_to_all`` of one 280-byte record per (row, shard): the shard's probability mass in 2^-40 fixed point relative
   to the global row maximum and its <= 16 candidates within 6.0 of it (order key, mass, vocabulary id);
4. for the rows the merged candidates cannot finish (flat distributions): ``all_gather`` of those rows' logit slices.

The owner of a row merges the S records (``dark_sample_from_shards``) and finishes the row exactly as the unsharded
sampler's head path does -- integer masses add in any order, the candidates are the same set -- so ``confidence`` and
``x0`` are bit-identical to ``dark_sample_rows`` on the full row, including the Philox draw (it is evaluated by the
owner at the kept positions of its own ``[M_r, V]`` stream).

FLOP identity (why this is NOT a throughput feature under data parallelism): every rank computes
``M_total x V/S = M_local x V`` logits, the same head FLOPs as the unsharded head, plus the exchanges above.  It pays
only when the head's weights or logits do not fit next to the body (V in the millions); at V = 25 427 the head is
39 MFLOP per masked row against 8.75 GFLOP per sequence-step.  ``emulate(S)`` runs the S shards on one GPU in sequence:
that is how the tests pin bit-equality with the unsharded sampler.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _hip


def shard_bounds(V: int, S: int, s: int) -> Tuple[int, int]:
    """(first column, number of columns) of shard s: equal widths rounded up to the GEMM's 128-row weight tiles."""
    width = -(-(-(-V // S)) // 128) * 128
    v0 = min(s * width, V)
    return v0, max(0, min(V - v0, width))


class ShardedHead:
    """``head_and_sample`` replacement for one model: ``group=None`` + ``n_shards`` emulates the shards on this GPU."""

    def __init__(self, model, group=None, n_shards: Optional[int] = None):
        self.model, self.group = model, group
        self.lib = _hip.load()
        self.distributed = n_shards is None
        if self.distributed:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("ShardedHead needs an initialised process group (or n_shards for the emulation)")
            self.S, self.rank = dist.get_world_size(group), dist.get_rank(group)
            self.staged = dist.get_backend(group) == "gloo"   # host staging: several ranks may share one card
        else:
            self.S, self.rank, self.staged = int(n_shards), 0, False
        self.V = model.config.vocab_size
        self.d = model.config.hidden_size
        self.fallback_rows = 0   # rows that needed their logits gathered (statistics)
        self.rows = 0

    # ---- collectives (identity in the emulation)
    def _comm(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if (self.staged and t.is_cuda) else t

    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """[n, ...] -> [S, n, ...] (same n on every rank)."""
        src = self._comm(t.contiguous())
        out = src.new_empty((self.S,) + tuple(src.shape))
        dist.all_gather_into_tensor(out.view(self.S * src.shape[0], *src.shape[1:]), src, group=self.group)
        return out.to(t.device)

    def _shard_logits(self, pk, xn: torch.Tensor, rows: int, s: int, stream) -> Tuple[torch.Tensor, int, int]:
        v0, Vs = shard_bounds(self.V, self.S, s)
        ld = max(Vs, 4)
        logits = torch.empty((rows, ld), dtype=torch.float32, device=xn.device)
        if Vs > 0:
            _hip.check(self.lib.dark_lm_head_shard(pk.ref, _hip.ptr(xn), rows, v0, Vs, _hip.ptr(logits), ld, stream))
        return logits, v0, Vs

    def head_and_sample(self, st, M: int, temperature, top_p, top_k, conf_mode, noise, philox):
        """conf [M] fp32, x0 [M] int64 for this rank's M masked rows; `st` is the model's HipState after
        forward_begin / forward_rows (or forward_hidden)."""
        lib, pk, dev = self.lib, st.pk, st.pk.device
        stream = st._s()
        S, V, d = self.S, self.V, self.d
        mdt = torch.bfloat16 if pk.elem == 2 else torch.float32
        # 1. normalised hidden rows of my masked positions
        if self.distributed:
            cnt = torch.tensor([M], dtype=torch.int64, device=dev)
            counts = self._all_gather(cnt).view(-1).tolist()
        else:
            counts = [M]
        R = len(counts)                       # owners: ranks (distributed) or just this one (emulation)
        Mmax = max(max(counts), 1)
        xn = torch.zeros((Mmax, d), dtype=mdt, device=dev)
        hidden, src = (st.hidden_rows, None) if st.rows_valid else (st.hidden, st.src_rows)
        _hip.check(lib.dark_final_norm_rows(pk.ref, _hip.ptr(hidden), _hip.ptr(src), M, _hip.ptr(xn), stream))
        xn_all = self._all_gather(xn).view(R * Mmax, d) if self.distributed else xn
        rows = R * Mmax
        # 2. my shard(s) of the logits, row maxima
        shards = [self.rank] if self.distributed else list(range(S))
        T = float(temperature)
        logits, mx = {}, torch.full((rows,), float("-inf"), dtype=torch.float32, device=dev)
        for s in shards:
            logits[s] = self._shard_logits(pk, xn_all, rows, s, stream)
            lg, v0, Vs = logits[s]
            if Vs > 0:
                part_mx = torch.empty(rows, dtype=torch.float32, device=dev)
                _hip.check(lib.dark_shard_row_max(_hip.ptr(lg), lg.stride(0), rows, Vs, T, _hip.ptr(part_mx), stream))
                mx = torch.maximum(mx, part_mx)   # max is exact: any order
        if self.distributed:
            m_ = self._comm(mx)
            dist.all_reduce(m_, op=dist.ReduceOp.MAX, group=self.group)
            mx = m_.to(dev)
        # 3. per-shard records relative to the global maxima, sent to the rows' owners
        PB = _hip.SHARD_PART_BYTES
        parts = torch.zeros((S if not self.distributed else 1, rows, PB), dtype=torch.uint8, device=dev)
        for i, s in enumerate(shards):
            lg, v0, Vs = logits[s]
            if Vs > 0:
                _hip.check(lib.dark_shard_partials(_hip.ptr(lg), lg.stride(0), rows, Vs, v0, T, _hip.ptr(mx),
                                                   _hip.ptr(parts[i]), stream))
        if self.distributed:
            send = self._comm(parts.view(R, Mmax, PB))
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.group)   # recv[s] = shard s's records of MY rows
            mine = recv.to(dev).contiguous()
            my_mx = mx[self.rank * Mmax: self.rank * Mmax + Mmax].contiguous()
        else:
            mine, my_mx = parts, mx
        # 4. merge and finish
        if st.conf.numel() >= Mmax:
            conf, x0 = st.conf[:Mmax], st.x0[:Mmax]
        else:  # another rank holds more rows than this one's buffers were sized for
            conf = torch.empty(Mmax, dtype=torch.float32, device=dev)
            x0 = torch.empty(Mmax, dtype=torch.int64, device=dev)
        flags = torch.zeros(Mmax, dtype=torch.uint8, device=dev)
        seed, off, nth = philox if philox is not None else (0, 0, 0)
        _hip.check(lib.dark_sample_from_shards(_hip.ptr(mine), Mmax, S, M, V, T, float(top_p), int(top_k), int(conf_mode),
                                               _hip.ptr(my_mx), _hip.ptr(noise),
                                               noise.stride(0) if noise is not None else 0, seed, off, nth,
                                               _hip.ptr(conf), _hip.ptr(x0), _hip.ptr(flags), stream))
        # 5. rows the candidates could not finish: gather their logits, finish them with the full-row sampler
        flags[M:] = 0
        all_flags = self._all_gather(flags).view(R, Mmax) if self.distributed else flags.view(1, Mmax)
        need = all_flags.nonzero(as_tuple=False)      # (owner, row) pairs, the same list on every rank
        self.rows += M
        if need.numel():
            glob = (need[:, 0] * Mmax + need[:, 1]).to(dev)
            F = int(glob.numel())
            width = shard_bounds(V, S, 0)[1]
            full = torch.empty((F, S * width), dtype=torch.float32, device=dev)
            if self.distributed:
                lg, v0, Vs = logits[self.rank]
                piece = torch.zeros((F, width), dtype=torch.float32, device=dev)
                piece[:, :Vs] = lg[glob, :Vs]
                full.view(F, S, width).copy_(self._all_gather(piece).permute(1, 0, 2))
            else:
                for s in shards:
                    lg, v0, Vs = logits[s]
                    full[:, s * width: s * width + Vs] = lg[glob, :Vs]
            own = (need[:, 0] == (self.rank if self.distributed else 0)).to(dev)
            my_rows = need[:, 1].to(dev)[own].to(torch.int32).contiguous()
            n = int(my_rows.numel())
            self.fallback_rows += n
            if n:
                mine_full = full[own].contiguous()
                _hip.check(lib.dark_sample_rows_indexed(_hip.ptr(mine_full), mine_full.stride(0), n, V, T, float(top_p),
                                                        int(top_k), int(conf_mode), _hip.ptr(my_rows), _hip.ptr(noise),
                                                        noise.stride(0) if noise is not None else 0, seed, off, nth,
                                                        _hip.ptr(conf), _hip.ptr(x0), stream))
        return conf[:M], x0[:M]
