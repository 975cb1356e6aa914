"""Host side of the variable-length (EXPAND / DELETE) denoise loop.

Mirrors the reference's ``akgr/abduction_model/model/custom_mydream/generation_expand_utils.py`` interface:
``DreamGenerationConfig`` (:85-140, the expand-specific attributes ``number_transfer_tokens``, ``expand_budget``,
``pad_delete_to_right``, ``expand_token_id``, ``delete_token_id``), ``DreamModelOutput`` (:79-82) and
``DreamGenerationMixin.diffusion_generate`` / ``_sample`` (:294-356, :358-503) keep their names, argument meaning,
defaults, return types and error behaviour; ``diffusion_generate_length`` / ``_sample_length`` (:505-672, a copy of
the fixed-length sampler) map onto ``dark_amd.generation_utils``.

The reference runs this path one sequence per call (``main.test``, akgr/abduction_model/main.py:253-274: it edits
``x[0]`` only).  Here every sequence of a batch carries its own window / budget / step-limit state on the device
(``dark_expand_*`` entry points of ``include/dark_hip.h``), so

* ``diffusion_generate(inputs [1, L])`` is the reference call, and
* ``diffusion_generate(inputs [B, L])`` / ``diffusion_generate_batch([ids_0 [1, L_0], ids_1 [1, L_1], ...])``
  run B *independent* reference calls together: row b of the result (and ``history[b]``) is what the
  reference returns for row b alone, given the same categorical noise.

Per step the host reads back two int32 (masked rows, active sequences), exactly where the reference syncs on
``torch.all(~mask_index)`` (:404); finished sequences are compacted away once enough of them have stopped.
"""
from __future__ import annotations

import warnings
from typing import Callable, List, Optional, Sequence, Union

import torch
import torch.nn.functional as F

from . import _hip
from . import generation_utils as _fixed
from .generation_utils import DreamModelOutput  # noqa: F401  (generation_expand_utils.py:79-82)

_CONF_MODE = {"maskgit_plus": _hip.CONF_PROB, "topk_margin": _hip.CONF_MARGIN, "entropy": _hip.CONF_NEG_ENTROPY}


class DreamGenerationConfig(_fixed.DreamGenerationConfig):
    """generation_expand_utils.py:85-140: the fixed-length attributes plus the expand / delete controls;
    ``alg`` defaults to 'entropy' and ``think_sample_interval`` to 8 here."""

    def __init__(self, **kwargs):
        ntt = kwargs.pop("number_transfer_tokens", 1)
        budget = kwargs.pop("expand_budget", None)
        pad_right = kwargs.pop("pad_delete_to_right", False)
        expand_id = kwargs.pop("expand_token_id", None)
        delete_id = kwargs.pop("delete_token_id", None)
        kwargs.setdefault("alg", "entropy")
        kwargs.setdefault("think_sample_interval", 8)
        super().__init__(**kwargs)
        self.number_transfer_tokens = ntt
        self.expand_budget = budget
        self.pad_delete_to_right = pad_right
        self.expand_token_id = expand_id
        self.delete_token_id = delete_id


class _ExpandBuffers:
    """Device state of one batched run: per-sequence window / budget / limit, the window mask and position ids
    the forward consumes, optional history snapshots."""

    def __init__(self, B: int, N: int, device, want_history: bool):
        i32 = dict(dtype=torch.int32, device=device)
        self.key_mask = torch.zeros(B * N, dtype=torch.uint8, device=device)
        self.pos = torch.zeros(B * N, **i32)
        self.active = torch.zeros(B, dtype=torch.uint8, device=device)
        self.snaps = torch.zeros((3, B, N), dtype=torch.int64, device=device) if want_history else None
        self.snap_len = torch.zeros((3, B), **i32) if want_history else None


class DreamGenerationMixin(_fixed.DreamGenerationMixin):
    """Drop-in for the reference's expand mixin; the model class supplies ``_hip_state``."""

    # ---- the fixed-length copy the reference keeps in this module (:505-672).  (In the reference
    # `diffusion_generate_length` calls `_prepare_generated_length` without its fourth argument and raises
    # TypeError; here it runs the fixed-length path.)
    @torch.no_grad()
    def diffusion_generate_length(self, inputs=None, generation_config=None, **kwargs):
        return self._generate_fixed(inputs, generation_config, **kwargs)

    def _sample_length(self, *args, **kwargs):
        return self._sample_fixed(*args, **kwargs)

    def _validate_generated_length(self, generation_config, input_ids_length, has_default_max_length):
        """generation_expand_utils.py:160-183 (`>` instead of `>=`: infilling without expansion is allowed)."""
        if has_default_max_length and generation_config.max_new_tokens is None and generation_config.max_length == 20:
            warnings.warn(
                f"Using the model-agnostic default `max_length` (={generation_config.max_length}) to control the "
                "generation length. We recommend setting `max_new_tokens` to control the maximum length of the "
                "generation.", UserWarning)
        if input_ids_length > generation_config.max_length:
            raise ValueError(
                f"Input length of input_ids is {input_ids_length}, but `max_length` is set to"
                f" {generation_config.max_length}. This can lead to unexpected behavior. You should consider"
                " increasing `max_length` or, better yet, setting `max_new_tokens`.")

    def _prepare_generated_length(self, generation_config, has_default_max_length, input_ids_length,
                                  num_mask_tokens=None):
        """generation_expand_utils.py:185-208; without `num_mask_tokens` the fixed-length rule applies."""
        if num_mask_tokens is None:
            return super()._prepare_generated_length(generation_config, has_default_max_length, input_ids_length)
        if generation_config.max_new_tokens is not None:
            generation_config.max_length = generation_config.max_new_tokens + input_ids_length - num_mask_tokens
        elif has_default_max_length:
            if generation_config.max_length == DreamGenerationConfig().max_length:
                generation_config.max_length = generation_config.max_length + input_ids_length
                mpe = getattr(self.config, "max_position_embeddings", None)
                if mpe is not None:
                    generation_config.max_length = min(generation_config.max_length, mpe)
        return generation_config

    def _prepare_generation_config(self, generation_config, **kwargs):
        """generation_expand_utils.py:210-241."""
        using_model_generation_config = generation_config is None
        if generation_config is None:
            generation_config = DreamGenerationConfig.from_model_config(self.config)
        elif not isinstance(generation_config, DreamGenerationConfig):
            # a fixed-length config object: lift it so that the expand attributes exist and `update` sees them
            generation_config = DreamGenerationConfig(**generation_config.to_dict())
        generation_config = super()._prepare_generation_config(generation_config, **kwargs)
        if not using_model_generation_config and generation_config.expand_token_id is None:
            generation_config.expand_token_id = getattr(self.generation_config, "expand_token_id", None)
        return generation_config

    # ------------------------------------------------------------------ entry points
    @torch.no_grad()
    def diffusion_generate(self, inputs: Optional[torch.Tensor] = None,
                           generation_config: Optional[DreamGenerationConfig] = None,
                           **kwargs) -> Union[DreamModelOutput, torch.LongTensor]:
        """generation_expand_utils.py:294-356.  `inputs` [1, L] is the reference call; [B, L] runs B independent
        reference calls (see the module docstring)."""
        assert inputs is not None
        if inputs.dim() != 2:
            raise ValueError("inputs must be [B, L]")
        rows = [inputs[b:b + 1] for b in range(inputs.shape[0])]
        return self._generate_rows(rows, generation_config, single=inputs.shape[0] == 1, **kwargs)

    @torch.no_grad()
    def diffusion_generate_batch(self, inputs: Sequence[torch.Tensor],
                                 generation_config: Optional[DreamGenerationConfig] = None,
                                 **kwargs) -> Union[DreamModelOutput, torch.LongTensor]:
        """Ragged batch: `inputs[b]` is the un-padded `[1, L_b]` (or `[L_b]`) prompt the reference would receive
        in its b-th call (akgr/tokenizer.py:154-184 builds them without padding).  No counterpart in the
        reference, which loops over samples on the host (main.py:253-274)."""
        rows = [t.reshape(1, -1) for t in inputs]
        if not rows:
            raise ValueError("empty batch")
        return self._generate_rows(rows, generation_config, single=False, **kwargs)

    def _generate_rows(self, rows: List[torch.Tensor], generation_config, single: bool, **kwargs):
        kwargs.pop("tokenizer", None)                                                            # :301
        noise_fn = kwargs.pop("_noise_fn", None)   # parity tests: (row, step, M_row, V) -> Exp(1) draws
        rank_noise_fn = kwargs.pop("_rank_noise_fn", None)   # parity tests: (row, step, M_row) -> multinomial's draws
        trace = kwargs.pop("_trace", None)         # parity tests: dict receiving per-step M / conf / x0 / rows
        compact = kwargs.pop("compact_finished", True)
        lazy_noise = bool(kwargs.pop("lazy_noise", True))   # see generation_utils.diffusion_generate
        # parity tests, teacher forcing: step -> (x [B, max_length], win [B], budget [B]) or None to stop
        state_hook = kwargs.pop("_state_hook", None)
        gc = self._prepare_generation_config(generation_config, **kwargs)
        attention_mask = kwargs.pop("attention_mask", None)
        assert attention_mask is None, ("We currently do not support attention_mask for DreamOn since we recompute "
                                        "attention mask after each denoising step.")             # :308-309
        mask_id = gc.mask_token_id
        if mask_id is None:
            raise ValueError("mask_token_id must be set in the generation config")

        # ---- per-row bookkeeping of :313-334, vectorised on the host (one device->host copy for the whole batch)
        lengths = [int(r.shape[1]) for r in rows]
        flat = torch.cat([r.detach().reshape(-1) for r in rows]).to("cpu", torch.int64)
        lens_t = torch.tensor(lengths, dtype=torch.int64)
        row_of = torch.repeat_interleave(torch.arange(len(rows)), lens_t)
        col_of = torch.arange(flat.numel()) - torch.repeat_interleave(torch.cumsum(lens_t, 0) - lens_t, lens_t)
        is_mask = flat == mask_id
        n_masks_t = torch.bincount(row_of[is_mask], minlength=len(rows))
        if bool((n_masks_t == 0).any()):
            raise ValueError("No mask tokens found in the input_ids.")                           # :318-319
        first = torch.full((len(rows),), 1 << 30, dtype=torch.int64)
        first.scatter_reduce_(0, row_of[is_mask], col_of[is_mask], reduce="amin")
        n_masks, prefixes = n_masks_t.tolist(), first.tolist()
        packed = (flat, row_of, col_of, lengths)
        has_default_max_length = kwargs.get("max_length") is None and gc.max_length is not None
        max_lengths = set()
        for L, nm in set(zip(lengths, n_masks)):
            g = self._prepare_generated_length(_copy_lengths(gc), has_default_max_length, L, nm)
            self._validate_generated_length(g, L, has_default_max_length)
            max_lengths.add(int(g.max_length))
        if len(max_lengths) != 1:
            raise NotImplementedError(f"rows resolve to different max_length values {sorted(max_lengths)}: pass an "
                                      "explicit max_length (the batched loop uses one row width)")
        gc.max_length = max_lengths.pop()
        if self.device.type != rows[0].device.type:
            warnings.warn(
                "You are calling .generate() with the `input_ids` being on a device type different"
                f" than your model's device. `input_ids` is on {rows[0].device.type}, whereas the model"
                f" is on {self.device.type}. You may experience unexpected behaviors or slower generation.",
                UserWarning)
        return self._sample(packed, prefixes, n_masks, gc, _single=single, _noise_fn=noise_fn, _trace=trace,
                            _compact=compact, _state_hook=state_hook, _lazy_noise=lazy_noise,
                            _rank_noise_fn=rank_noise_fn)

    # ------------------------------------------------------------------ the loop
    def _sample(self, input_ids, prefix_len, start_gen_len, generation_config, _single: Optional[bool] = None,
                _noise_fn: Optional[Callable] = None, _trace: Optional[dict] = None, _compact: bool = True,
                _state_hook: Optional[Callable] = None, _lazy_noise: bool = True,
                _rank_noise_fn: Optional[Callable] = None):
        """generation_expand_utils.py:358-503 for every row at once.  `input_ids`: a `[1, L]` tensor with scalar
        `prefix_len` / `start_gen_len` (the reference signature) or lists of them."""
        if self.device.type != "cuda":
            raise RuntimeError("dark_amd runs on a HIP device only: move the model with .to('cuda')")
        with torch.cuda.device(self.device):  # the library launches on the CURRENT device's stream
            return self._sample_impl(input_ids, prefix_len, start_gen_len, generation_config, _single, _noise_fn,
                                     _trace, _compact, _state_hook, _lazy_noise, _rank_noise_fn)

    def _sample_impl(self, input_ids, prefix_len, start_gen_len, generation_config, _single, _noise_fn, _trace,
                     _compact, _state_hook, _lazy_noise, _rank_noise_fn):
        gc = generation_config
        if isinstance(input_ids, torch.Tensor):   # the reference signature: one [1, L] prompt
            L = int(input_ids.shape[1])
            input_ids = (input_ids.detach().reshape(-1).to("cpu", torch.int64), torch.zeros(L, dtype=torch.int64),
                         torch.arange(L), [L])
            prefix_len, start_gen_len = [int(prefix_len)], [int(start_gen_len)]
            _single = True if _single is None else _single
        flat, row_of, col_of, lengths = input_ids
        device = self.device
        if device.type != "cuda":
            raise RuntimeError("dark_amd runs on a HIP device only: move the model with .to('cuda')")
        alg, alg_temp = gc.alg, gc.alg_temp
        if alg not in _CONF_MODE:
            raise RuntimeError(f"Unknown alg: {alg}")
        temperature = float(gc.temperature)
        top_p_f = max(float(gc.top_p), 0.0) if (gc.top_p is not None and gc.top_p < 1) else -1.0  # < 0 = off
        top_k_i = int(min(gc.top_k, self.config.vocab_size)) if gc.top_k is not None else 0
        k = int(gc.number_transfer_tokens)
        N = int(gc.max_length)
        mask_id, eos_id = int(gc.mask_token_id), gc.eos_token_id
        if isinstance(eos_id, (list, tuple)):
            eos_id = eos_id[0]
        if eos_id is None:
            raise ValueError("eos_token_id must be set: rows are padded with it (generation_expand_utils.py:394)")
        delete_id = -1 if gc.delete_token_id is None else int(gc.delete_token_id)
        expand_id = gc.expand_token_id
        if expand_id is None:
            raise ValueError("expand_token_id must be set in the generation config")
        stochastic_rank = not (alg_temp is None or alg_temp == 0)
        B0, V = len(lengths), self.config.vocab_size
        want_history = bool(gc.return_dict_in_generate and gc.output_history)

        # ---- per-row state (:378-395)
        x_host = torch.full((B0, N), int(eos_id), dtype=torch.int64)
        x_host[row_of, col_of] = flat
        max_gen = [N - p for p in prefix_len]
        budget0 = [(2 * g if gc.expand_budget is None else int(gc.expand_budget)) for g in max_gen]
        limit0 = [2 * g + 2 * e for g, e in zip(max_gen, budget0)]
        i32 = dict(dtype=torch.int32, device=device)
        x = x_host.to(device)
        win = torch.tensor(lengths, **i32)   # L - start_gen_len + num_generation_tokens
        budget = torch.tensor(budget0, **i32)
        limit = torch.tensor(limit0, **i32)
        origin = torch.arange(B0, device=device)        # row of the caller's batch each live row belongs to
        origin_host = list(range(B0))
        out = x.clone()
        histories: Optional[List[List[torch.Tensor]]] = [[] for _ in range(B0)] if want_history else None

        B = B0
        st = self._hip_state(B, N)
        buf = _ExpandBuffers(B, N, device, want_history)
        lib, s = st.lib, st._s()
        state_hook = _state_hook
        if state_hook is not None:
            _compact = False
        for i in range(max(limit0)):
            if state_hook is not None:
                forced = state_hook(i)
                if forced is None:
                    break
                x.copy_(forced[0].to(device))
                win.copy_(torch.as_tensor(forced[1], dtype=torch.int32).reshape(-1).to(device))
                budget.copy_(torch.as_tensor(forced[2], dtype=torch.int32).reshape(-1).to(device))
            _hip.check(lib.dark_expand_prepare(_hip.ptr(x), _hip.ptr(win), _hip.ptr(limit), i, B, N, mask_id,
                                               _hip.ptr(buf.key_mask), _hip.ptr(buf.pos), _hip.ptr(st.row_index),
                                               _hip.ptr(st.src_rows), _hip.ptr(buf.active), _hip.ptr(st.count), s))
            st.count_host.copy_(st.count, non_blocking=True)
            st.count_event.record(torch.cuda.current_stream(device))
            st.count_event.synchronize()          # the reference syncs here too: `if torch.all(~mask_index)` (:404)
            M, n_active = int(st.count_host[0]), int(st.count_host[1])
            if n_active == 0:
                break
            if _compact and B > 64 and n_active * 4 <= B * 3:
                # drop finished sequences: their rows are final (x never changes once a row is inactive)
                out[origin] = x
                keep = buf.active[:B].nonzero(as_tuple=False).flatten()
                x, win, budget, limit, origin = (t[keep].contiguous() for t in (x, win, budget, limit, origin))
                origin_host = origin.tolist() if want_history or _noise_fn or _rank_noise_fn or _trace is not None else origin_host
                B = int(keep.numel())
                st = self._hip_state(B, N)
                buf = _ExpandBuffers(B, N, device, want_history)
                lib, s = st.lib, st._s()
                _hip.check(lib.dark_expand_prepare(_hip.ptr(x), _hip.ptr(win), _hip.ptr(limit), i, B, N, mask_id,
                                                   _hip.ptr(buf.key_mask), _hip.ptr(buf.pos), _hip.ptr(st.row_index),
                                                   _hip.ptr(st.src_rows), _hip.ptr(buf.active), _hip.ptr(st.count),
                                                   s))
            st.hidden_valid = False
            st.forward_begin(x, buf.pos, buf.key_mask)   # the last layer's o_proj / MLP only for the M masked rows
            logits = st.head_rows(None, M, st.forward_rows(M))
            _hip.check(lib.dark_expand_block_logit(_hip.ptr(logits), V, V, _hip.ptr(st.row_index), M, B, N,
                                                   _hip.ptr(win), _hip.ptr(budget), int(expand_id), s))
            noise = philox = None
            if temperature > 0:
                if _noise_fn is not None:
                    seq_of = (st.row_index[:M] // N).tolist()
                    per = {}
                    for b_live in seq_of:
                        per[b_live] = per.get(b_live, 0) + 1
                    noise = torch.cat([_noise_fn(origin_host[b_live], i, m_b, V).to(torch.float32)
                                       for b_live, m_b in per.items()]).to(device).contiguous()
                    assert noise.shape == (M, V)
                elif _lazy_noise:
                    philox = _fixed.philox_fill_plan(device, M * V)
                else:
                    noise = torch.empty((M, V), device=device, dtype=torch.float32).exponential_()
            conf, x0 = st.conf[:M], st.x0[:M]
            if philox is not None:
                _hip.check(lib.dark_sample_rows_philox(_hip.ptr(logits), V, M, V, temperature, top_p_f, top_k_i,
                                                       _CONF_MODE[alg], philox[0], philox[1], philox[2],
                                                       _hip.ptr(conf), _hip.ptr(x0), s))
            else:
                _hip.check(lib.dark_sample_rows(_hip.ptr(logits), V, M, V, temperature, top_p_f, top_k_i,
                                                _CONF_MODE[alg], _hip.ptr(noise), V if noise is not None else 0,
                                                _hip.ptr(conf), _hip.ptr(x0), s))
            if _trace is not None:
                _trace.setdefault("M", []).append(M)
                _trace.setdefault("rows", []).append([origin_host[b] for b in (st.row_index[:M] // N).tolist()])
                _trace.setdefault("confidence", []).append(conf.clone())
                _trace.setdefault("x0", []).append(x0.clone())
            q = None
            if stochastic_rank:
                # :436-439: torch.multinomial(softmax(confidence / alg_temp), k) == topk(p / q, k) per sequence, q =
                # the Exp(1) fill ATen's multinomial draws (one [M] fill per step: the reference's own stream when
                # the batch is one sequence)
                if _single and k > M:
                    raise RuntimeError(f"cannot sample {k} positions out of {M} masked ones without replacement")
                if _rank_noise_fn is not None:
                    seq_of = (st.row_index[:M] // N).tolist()
                    per = {}
                    for b_live in seq_of:
                        per[b_live] = per.get(b_live, 0) + 1
                    q = torch.cat([_rank_noise_fn(origin_host[b_live], i, m_b).to(torch.float32).reshape(-1)
                                   for b_live, m_b in per.items()]).to(device).contiguous()
                else:
                    q = st.exponential_like_torch(M, _fixed.philox_fill_plan(device, M))
            elif _single and k > M:
                raise RuntimeError(f"selected index k out of range: topk({k}) over {M} masked positions (:435)")
            _hip.check(lib.dark_expand_commit_edit_sampled(
                _hip.ptr(x), B, N, _hip.ptr(st.row_index), M, _hip.ptr(conf), _hip.ptr(x0), k, _hip.ptr(buf.active),
                _hip.ptr(win), _hip.ptr(budget), mask_id, delete_id, int(expand_id), 1 if gc.pad_delete_to_right else 0,
                _hip.ptr(buf.snaps), _hip.ptr(buf.snap_len), float(alg_temp) if stochastic_rank else 0.0, _hip.ptr(q), s))
            if histories is not None:
                lens = buf.snap_len.cpu()
                snaps = buf.snaps.clone()      # one device copy per step; the appended rows are views of it
                step_snaps = []
                for b_live in range(B):
                    for which in range(3):
                        n = int(lens[which, b_live])
                        if n >= 0:
                            histories[origin_host[b_live]].append(snaps[which, b_live, :n])
                            step_snaps.append((origin_host[b_live], snaps[which, b_live, :n]))
                if _trace is not None:
                    _trace.setdefault("snaps", []).append(step_snaps)
        out[origin] = x
        if gc.return_dict_in_generate:
            hist = None
            if histories is not None:
                hist = histories[0] if _single else histories
            return DreamModelOutput(sequences=out, history=hist)
        return out


def _copy_lengths(gc):
    """A throw-away view of the two attributes `_prepare_generated_length` rewrites."""
    class _G:
        pass
    g = _G()
    g.max_length, g.max_new_tokens = gc.max_length, gc.max_new_tokens
    return g
