"""Host side of the fixed-length denoise / unmask loop.

Mirrors the reference's ``akgr/abduction_model/model/custom_mydream/generation_utils.py``
interface for this path: ``DreamGenerationConfig`` (:99-150), ``DreamModelOutput`` (:93-96) and
``DreamGenerationMixin.diffusion_generate`` / ``_sample`` (:300-365, :435-548) keep their names,
argument meaning, defaults, return types and error behaviour.  What runs underneath is the HIP
library (``dark_amd/_hip.py``); this file only sequences entry points on torch's current stream
and draws the categorical noise from torch's device generator exactly where the reference does.

Per step (confidence algorithms, default hooks) the loop is sync-free except for ONE 4-byte
count read-back (the reference syncs at ``int(num_mask_token * ...)``, :521); the body forward is
enqueued before the host waits for it, so the GPU never idles on that read.
"""
from __future__ import annotations

import copy
import json
import os
import warnings
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn.functional as F

from . import _hip

_CONF_MODE = {"maskgit_plus": _hip.CONF_PROB, "topk_margin": _hip.CONF_MARGIN, "entropy": _hip.CONF_NEG_ENTROPY}


@dataclass
class DreamModelOutput:
    """generation_utils.py:93-96."""
    sequences: torch.LongTensor = None
    history: Optional[Tuple[torch.LongTensor]] = None

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else (self.sequences, self.history)[k]


class DreamGenerationConfig:
    """Sampler parameters, same names and defaults as generation_utils.py:99-150."""

    def __init__(self, **kwargs):
        self.temperature: float = kwargs.pop("temperature", 0.0)
        self.top_p: Optional[float] = kwargs.pop("top_p", None)
        self.top_k: Optional[int] = kwargs.pop("top_k", None)
        self.max_length = kwargs.pop("max_length", 20)
        self.max_new_tokens = kwargs.pop("max_new_tokens", None)
        self.eps: float = kwargs.pop("eps", 1e-3)
        self.steps: int = kwargs.pop("steps", 512)
        self.alg: str = kwargs.pop("alg", "origin")
        self.alg_temp: Optional[float] = kwargs.pop("alg_temp", None)
        self.num_return_sequences: int = kwargs.pop("num_return_sequences", 1)
        self.return_dict_in_generate: bool = kwargs.pop("return_dict_in_generate", False)
        self.output_history: bool = kwargs.pop("output_history", False)
        self.mask_token_id = kwargs.pop("mask_token_id", None)
        self.pad_token_id = kwargs.pop("pad_token_id", None)
        self.bos_token_id = kwargs.pop("bos_token_id", None)
        self.eos_token_id = kwargs.pop("eos_token_id", None)
        self.sep_token_id = kwargs.pop("sep_token_id", None)
        self.think_sample = kwargs.pop("think_sample", False)
        self.think_sample_interval = kwargs.pop("think_sample_interval", 2)
        self.generation_kwargs = kwargs.pop("generation_kwargs", {})
        self._from_model_config = kwargs.pop("_from_model_config", False)
        kwargs.pop("_commit_hash", None)
        kwargs.pop("transformers_version", None)
        if not self._from_model_config:
            for key, value in kwargs.items():  # e.g. expand_token_id / delete_token_id (main.py:359-360)
                setattr(self, key, value)

    def update(self, **kwargs) -> Dict[str, Any]:
        """Set attributes that already exist, return the kwargs that were not used
        (``GenerationConfig.update`` semantics: unknown keys such as the callers'
        ``number_transfer_tokens=1`` are ignored, generation_utils.py:237-239)."""
        unused = {}
        for k, v in kwargs.items():
            if hasattr(self, k):
                setattr(self, k, v)
            else:
                unused[k] = v
        return unused

    def validate(self, is_init=False):
        pass

    def to_dict(self) -> Dict[str, Any]:
        return {k: copy.deepcopy(v) for k, v in self.__dict__.items() if not k.startswith("_")}

    @classmethod
    def from_model_config(cls, model_config) -> "DreamGenerationConfig":
        d = model_config.to_dict() if hasattr(model_config, "to_dict") else dict(model_config)
        d.pop("_from_model_config", None)
        return cls(**d, _from_model_config=True)

    @classmethod
    def from_pretrained(cls, path: str, **_ignored) -> "DreamGenerationConfig":
        with open(os.path.join(path, "generation_config.json")) as f:
            return cls(**json.load(f))

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "generation_config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)


def transfer_count(num_masked: int, batch: int, t: torch.Tensor, s: torch.Tensor, last: bool) -> int:
    """``number_transfer_tokens`` of generation_utils.py:520-521, in the same fp32 tensor arithmetic
    (batch-average mask count, float truncation)."""
    num_mask_token = torch.tensor(num_masked, dtype=torch.int64) / batch
    return int(num_mask_token) if last else int(num_mask_token * (1 - s / t))


def philox_fill_plan(device: torch.device, numel: int) -> Tuple[int, int, int]:
    """Reserve, on torch's device generator, exactly what ``torch.empty(numel, device=device).exponential_()`` would
    consume, and return (seed, offset, nthreads) for ``dark_sample_rows_philox`` to reproduce that fill lazily.
    ATen's launch plan (ATen/native/cuda/DistributionTemplates.h, calc_execution_policy): 256-thread blocks,
    grid = min(CUs * (max threads per CU / 256), ceil(numel / 256)), 4 values per Philox call."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    props = torch.cuda.get_device_properties(idx)
    grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)
    nthreads = grid * 256
    gen = torch.cuda.default_generators[idx]
    seed, offset = gen.initial_seed(), gen.get_offset()
    gen.set_offset(offset + ((numel - 1) // (nthreads * 4) + 1) * 4)
    return seed, offset, nthreads


@dataclass
class _LoopCtx:
    """Per-call options of one ``diffusion_generate`` (never stored on the model: calls on different threads or
    nested calls from hooks do not see each other's settings)."""
    skip_noop: bool = True          # reuse the hidden states of an unchanged x on steps that commit nothing
    lazy_noise: bool = True         # Exp(1) draws evaluated inside the sampler from torch's Philox stream
    noise_fn: Optional[Callable] = None      # parity tests: inject the Exp(1) draws, (step, M, V) -> [M, V]
    rand_fn: Optional[Callable] = None       # parity tests: inject alg='origin' uniforms, (step, M) -> [M]
    rank_noise_fn: Optional[Callable] = None  # parity tests: inject multinomial's Exp(1) draws, (step, B, N) -> [B, N]
    state_hook: Optional[Callable] = None    # parity tests: teacher forcing, (step, x) -> x
    count_reduce: Optional[Callable] = None  # dispatch.generate_sharded(exact_ragged_k=True)
    trace: Optional[dict] = None             # parity tests: receives per-step k / conf / x0
    head: Optional[Any] = None               # vocab_shard.ShardedHead: head + sampling split over a process group


class DreamGenerationMixin:
    """Drop-in for the reference mixin; the model class supplies the ``_hip_*`` launch helpers."""

    @staticmethod
    def _expand_inputs_for_generation(expand_size=1, input_ids=None, attention_mask=None):
        """generation_utils.py:153-168."""
        if expand_size == 1:
            return input_ids, attention_mask
        if input_ids is not None:
            input_ids = input_ids.repeat_interleave(expand_size, dim=0)
        if attention_mask is not None:
            attention_mask = attention_mask.repeat_interleave(expand_size, dim=0)
        return input_ids, attention_mask

    def _validate_generated_length(self, generation_config, input_ids_length, has_default_max_length):
        """generation_utils.py:170-192."""
        if has_default_max_length and generation_config.max_new_tokens is None and generation_config.max_length == 20:
            warnings.warn(
                f"Using the model-agnostic default `max_length` (={generation_config.max_length}) to control the "
                "generation length. We recommend setting `max_new_tokens` to control the maximum length of the "
                "generation.", UserWarning)
        if input_ids_length >= generation_config.max_length:
            raise ValueError(
                f"Input length of input_ids is {input_ids_length}, but `max_length` is set to"
                f" {generation_config.max_length}. This can lead to unexpected behavior. You should consider"
                " increasing `max_length` or, better yet, setting `max_new_tokens`.")

    def _prepare_generated_length(self, generation_config, has_default_max_length, input_ids_length):
        """generation_utils.py:194-219."""
        if generation_config.max_new_tokens is not None:
            generation_config.max_length = generation_config.max_new_tokens + input_ids_length
        elif has_default_max_length:
            if generation_config.max_length == DreamGenerationConfig().max_length:
                generation_config.max_length = generation_config.max_length + input_ids_length
                mpe = getattr(self.config, "max_position_embeddings", None)
                if mpe is not None:
                    generation_config.max_length = min(generation_config.max_length, mpe)
        return generation_config

    def _prepare_generation_config(self, generation_config, **kwargs):
        """generation_utils.py:221-251."""
        using_model_generation_config = False
        if generation_config is None:
            generation_config = DreamGenerationConfig.from_model_config(self.config)
            using_model_generation_config = True
        generation_config = copy.deepcopy(generation_config)
        generation_config.update(**kwargs)
        if not using_model_generation_config:
            for name in ("bos_token_id", "eos_token_id", "pad_token_id", "mask_token_id"):
                if getattr(generation_config, name) is None:
                    setattr(generation_config, name, getattr(self.generation_config, name, None))
        return generation_config

    @torch.no_grad()
    def diffusion_generate(self, inputs: Optional[torch.Tensor] = None,
                           generation_config: Optional[DreamGenerationConfig] = None,
                           **kwargs) -> Union[DreamModelOutput, torch.LongTensor]:
        """generation_utils.py:300-365."""
        return self._generate_fixed(inputs, generation_config, **kwargs)

    def _sample(self, *args, **kwargs):
        """generation_utils.py:435-548."""
        return self._sample_fixed(*args, **kwargs)

    @torch.no_grad()
    def _generate_fixed(self, inputs: Optional[torch.Tensor] = None,
                        generation_config: Optional[DreamGenerationConfig] = None,
                        **kwargs) -> Union[DreamModelOutput, torch.LongTensor]:
        """The fixed-length entry (generation_utils.py:300-365; also generation_expand_utils.py:505-570)."""
        generation_config = self._prepare_generation_config(generation_config, **kwargs)
        tokens_hook = kwargs.pop("generation_tokens_hook_func", None)
        logits_hook = kwargs.pop("generation_logits_hook_func", None)
        # Steps whose transfer count is 0 leave x untouched (with 11 masks and 32/64 steps the reference's own
        # schedule makes 22/54 of them): their only observable effect is the RNG draw.  With skip_noop_steps the
        # loop keeps that draw and reuses the hidden states of the unchanged x instead of recomputing them.
        # lazy_noise: the Exp(1) draws of Categorical.sample() are computed inside the sampler from torch's Philox
        # stream (same generator state, same values as the [M, V] `exponential_` fill the reference makes) instead
        # of being written to and read back from HBM; False materialises them with torch.
        # _count_reduce (dispatch.generate_sharded): (count tensor, B_local) -> (global masks, global rows)
        ctx = _LoopCtx(skip_noop=bool(kwargs.pop("skip_noop_steps", True)),
                       lazy_noise=bool(kwargs.pop("lazy_noise", True)),
                       noise_fn=kwargs.pop("_noise_fn", None), rand_fn=kwargs.pop("_rand_fn", None),
                       rank_noise_fn=kwargs.pop("_rank_noise_fn", None),
                       state_hook=kwargs.pop("_state_hook", None), count_reduce=kwargs.pop("_count_reduce", None),
                       trace=kwargs.pop("_trace", None), head=kwargs.pop("_sharded_head", None))

        assert inputs is not None
        input_ids = inputs
        attention_mask = kwargs.pop("attention_mask", None)
        if generation_config.pad_token_id is None and generation_config.eos_token_id is not None:
            eos = generation_config.eos_token_id
            generation_config.pad_token_id = eos[0] if isinstance(eos, (list, tuple)) else eos

        input_ids_length = input_ids.shape[-1]
        has_default_max_length = kwargs.get("max_length") is None and generation_config.max_length is not None
        generation_config = self._prepare_generated_length(generation_config, has_default_max_length,
                                                           input_ids_length)
        self._validate_generated_length(generation_config, input_ids_length, has_default_max_length)

        if self.device.type != input_ids.device.type:
            warnings.warn(
                "You are calling .generate() with the `input_ids` being on a device type different"
                f" than your model's device. `input_ids` is on {input_ids.device.type}, whereas the model"
                f" is on {self.device.type}. You may experience unexpected behaviors or slower generation.",
                UserWarning)
        if (generation_config.pad_token_id is not None and attention_mask is None
                and bool(torch.any(input_ids == generation_config.pad_token_id))):
            warnings.warn(
                "Padding was detected but no attention mask is passed here. For correct "
                "generation results, please set `attention_mask` when batch-padding inputs.", UserWarning)

        input_ids, attention_mask = self._expand_inputs_for_generation(
            expand_size=generation_config.num_return_sequences, input_ids=input_ids, attention_mask=attention_mask)
        return self._sample_fixed(input_ids, attention_mask=attention_mask, generation_config=generation_config,
                                  generation_tokens_hook_func=tokens_hook, generation_logits_hook_func=logits_hook,
                                  ctx=ctx)

    # ------------------------------------------------------------------ the loop
    def _sample_fixed(self, input_ids: torch.LongTensor, attention_mask: Optional[torch.Tensor],
                      generation_config: DreamGenerationConfig, generation_tokens_hook_func=None,
                      generation_logits_hook_func=None, ctx: Optional[_LoopCtx] = None
                      ) -> Union[DreamModelOutput, torch.LongTensor]:
        """generation_utils.py:435-548 (and its copy generation_expand_utils.py:572-672)."""
        if self.device.type != "cuda":
            raise RuntimeError("dark_amd runs on a HIP device only: move the model with .to('cuda')")
        with torch.cuda.device(self.device):  # the library launches on the CURRENT device's stream
            return self._sample_fixed_impl(input_ids, attention_mask, generation_config, generation_tokens_hook_func,
                                           generation_logits_hook_func, ctx or _LoopCtx())

    def _sample_fixed_impl(self, input_ids, attention_mask, generation_config, generation_tokens_hook_func,
                           generation_logits_hook_func, ctx: _LoopCtx):
        gc = generation_config
        steps, eps, alg, alg_temp = gc.steps, gc.eps, gc.alg, gc.alg_temp
        temperature, top_p, top_k = gc.temperature, gc.top_p, gc.top_k
        mask_token_id = gc.mask_token_id
        histories = [] if (gc.return_dict_in_generate and gc.output_history) else None
        think = bool(gc.think_sample)
        if mask_token_id is None:
            raise ValueError("mask_token_id must be set in the generation config")
        if alg not in ("origin", "maskgit_plus", "topk_margin", "entropy"):
            raise RuntimeError(f"Unknown alg: {alg}")
        device = self.device
        # the kernels read int64 ids; nn.Embedding in the reference also takes int32 (modeling_dream_length.py:665)
        input_ids = input_ids.to(device=device, dtype=torch.int64)

        x = F.pad(input_ids, (0, gc.max_length - input_ids.shape[1]), value=mask_token_id).contiguous()
        B, N = x.shape
        key_mask = pos = None
        if attention_mask is not None and bool(torch.any(attention_mask == 0.0)):
            am = F.pad(attention_mask.to(device), (0, gc.max_length - attention_mask.shape[1]), value=1.0)
            tok_idx = am.long().cumsum(-1) - 1
            tok_idx.masked_fill_(am == 0, 1)
            key_mask = (am != 0).to(torch.uint8).contiguous()
            pos = tok_idx.to(torch.int32).contiguous()
        timesteps = torch.linspace(1, eps, steps + 1)  # host fp32: drives the k schedule without a device sync

        default_hooks = generation_tokens_hook_func is None and generation_logits_hook_func is None
        tokens_hook = generation_tokens_hook_func or (lambda step, x, logits: x)
        logits_hook = generation_logits_hook_func or (lambda step, x, logits: logits)
        # Native loop for every algorithm; only user hooks and think_sample need the materialised [B, N, V] logits
        fast = default_hooks and not think
        rank_temp = float(alg_temp) if (alg_temp is not None and alg_temp != 0) else 0.0
        if ctx.head is not None and (not fast or alg == "origin"):
            raise NotImplementedError("the vocabulary-sharded head serves the confidence algorithms with default hooks")
        # the reference filters iff `top_p is not None and top_p < 1` (:62): 0 keeps the top-1 token only; < 0 = off
        top_p_f = max(float(top_p), 0.0) if (top_p is not None and top_p < 1) else -1.0
        top_k_i = int(min(top_k, self.config.vocab_size)) if top_k is not None else 0

        x = tokens_hook(None, x, None).contiguous()
        st = self._hip_state(B, N)
        st.hidden_valid = False
        for i in range(steps):
            t, s = timesteps[i], timesteps[i + 1]
            last = i == steps - 1
            if fast and alg == "origin":
                self._step_origin(st, x, pos, key_mask, mask_token_id, temperature, top_p_f, top_k_i, t, s, last, i, ctx)
            elif fast:
                self._step_fast(st, x, pos, key_mask, mask_token_id, temperature, top_p_f, top_k_i, _CONF_MODE[alg],
                                t, s, last, i, ctx, rank_temp)
            else:
                # think_sample steps (generation_utils.py:490-491, :411-432): `check_and_select` forwards 4 identical
                # copies of every row with arange positions (`calculate_results`, :369-372), resamples them twice and
                # returns the first-pass logits of the copy whose resampled observation side is closest -- the
                # copies' logits are identical, so what reaches the sampler is the forward with arange positions.
                # That forward is what runs here; `_think_side_effects` reproduces what else the call does: the
                # generator advance of its two discarded categorical draws and its IndexError.
                think_step = think and i % gc.think_sample_interval == 0
                if think_step:
                    self._think_side_effects(x, mask_token_id, gc.sep_token_id, ctx)
                x = self._step_general(st, x, None if think_step else pos, key_mask, mask_token_id, temperature,
                                       top_p_f, top_k_i, alg, alg_temp, t, s, last, i, tokens_hook, logits_hook,
                                       ctx).contiguous()
            if ctx.state_hook is not None:
                x.copy_(ctx.state_hook(i, x).to(x.device))
                st.hidden_valid = False
            if histories is not None:
                histories.append(x.clone())

        if gc.return_dict_in_generate:
            return DreamModelOutput(sequences=x, history=histories)
        return x

    def _think_side_effects(self, x: torch.Tensor, mask_id: int, sep_id, ctx: _LoopCtx) -> None:
        """What `check_and_select` (generation_utils.py:411-432) does besides returning logits.
        * `sep_positions[0]` (:418-419) raises IndexError when row 0 holds no SEP.  The reference searches row 0
          AFTER its masked positions were filled with the first think draw; here the unmasked positions of row 0 are
          searched, so the two differ only if that discarded draw itself produced a SEP token in row 0.
        * `calculate_results` runs `Categorical(probs=softmax(logits[4B, N, V])).sample()` twice with its default
          temperature 1.0 (:374-377, :414, :426): two `exponential_` fills of [4*B*N, V] each.  The samples never
          reach the result, but the fills advance the global generator; the same Philox offsets are reserved here
          (nothing is generated).  fp32 models only: in bf16 the reference's Categorical fails its simplex check on
          a V-way softmax, falls into the bare `except` and draws nothing."""
        row0 = x[0]
        if sep_id is None or not bool(((row0 == sep_id) & (row0 != mask_id)).any()):
            raise IndexError("index 0 is out of bounds for dimension 0 with size 0")  # sep_positions[0], :419
        if ctx.noise_fn is None and self.dtype == torch.float32:
            B, N = x.shape
            for _ in range(2):
                philox_fill_plan(x.device, 4 * B * N * self.config.vocab_size)

    def _draw_noise(self, step: int, M: int, V: int, device, noise_fn) -> torch.Tensor:
        """The ``q = empty_like(probs).exponential_()`` of ATen multinomial, drawn from torch's device
        generator with the shape and at the point the reference draws it (generation_utils.py:70)."""
        if noise_fn is not None:
            q = noise_fn(step, M, V).to(device=device, dtype=torch.float32).contiguous()
            assert q.shape == (M, V)
            return q
        return torch.empty((M, V), device=device, dtype=torch.float32).exponential_()

    def _step_origin(self, st, x, pos, key_mask, mask_id, temperature, top_p, top_k, t, s, last, step,
                     ctx: _LoopCtx) -> None:
        """One iteration of generation_utils.py:488-510 for alg='origin', on device: every masked position draws
        `rand < 1 - s/t` (torch's generator, as the reference), the positions that pass are sampled -- the last
        layer's tail, the head and the sampler run for those rows ONLY -- and written into x; the others stay MASK."""
        B, N = x.shape
        V = self.config.vocab_size
        st.mask_rows(x, mask_id)
        st.forward_begin(x, pos, key_mask)
        st.hidden_valid = False
        M = st.wait_count()
        if M == 0:
            return
        p_transfer = 1 - s / t if not last else 1
        u = torch.rand(M, device=x.device) if ctx.rand_fn is None else ctx.rand_fn(step, M).to(x.device)
        transfer = u < (p_transfer.to(x.device) if isinstance(p_transfer, torch.Tensor) else p_transfer)
        sel = transfer.nonzero(as_tuple=False).flatten()   # (the reference syncs here too: boolean indexing, :509)
        n = int(sel.numel())
        if n == 0:
            return
        rows = st.src_rows[:M][sel].contiguous()
        logits = st.head_rows(None, n, st.forward_rows(n, rows))
        noise = philox = None
        if temperature > 0:
            if ctx.noise_fn is None and ctx.lazy_noise:
                philox = philox_fill_plan(x.device, n * V)
            else:
                noise = self._draw_noise(step, n, V, x.device, ctx.noise_fn)
        _, x0 = st.sample_into(logits, n, temperature, top_p, top_k, _hip.CONF_PROB, noise, philox)
        st.scatter(x, M, sel, x0, n)

    def _step_fast(self, st, x, pos, key_mask, mask_id, temperature, top_p, top_k, conf_mode, t, s, last, step,
                   ctx: _LoopCtx, rank_temp: float = 0.0) -> None:
        """One iteration of generation_utils.py:488-540 for the confidence algorithms, fully on device.
        rank_temp = alg_temp when the transfer positions are drawn (multinomial, :527-530) instead of ranked."""
        B, N = x.shape
        V = self.config.vocab_size
        noise_fn, trace = ctx.noise_fn, ctx.trace
        # with a sharded head every rank must join every step's collectives: nothing is skipped
        lockstep = ctx.head is not None and ctx.head.distributed
        skip = ctx.skip_noop and trace is None and not lockstep
        if not (skip and st.hidden_valid):
            st.mask_rows(x, mask_id)              # row_index / src_rows / count, then async count read-back
            # enqueued before the host waits for the count: everything up to the last layer's attention; the rest of
            # that layer is only needed for the M rows the head reads and follows once M is known (forward_rows)
            st.forward_begin(x, pos, key_mask)
            st.hidden_valid = True                # ... for as long as x does not change
        M = st.wait_count()
        Mg, Bg = (M, B) if ctx.count_reduce is None else ctx.count_reduce(st.count[:1], B)
        k = transfer_count(Mg, Bg, t, s, last)
        if trace is not None:
            trace.setdefault("k", []).append(k)
            trace.setdefault("M", []).append(M)
        if M == 0 and not lockstep:
            return
        noise = philox = None
        if temperature > 0 and M > 0:
            # lazy noise (the draws evaluated inside the sampler) unless the rows are flat, where one Philox call per
            # kept token costs more than the [M, V] fill it replaces (HipState.note_entropy); a step that samples
            # nothing only reserves the generator offsets.  Same stream, same bits either way.
            if noise_fn is None and ctx.lazy_noise and (not st.flat_rows or (skip and k == 0)):
                philox = philox_fill_plan(x.device, M * V)
            else:
                noise = self._draw_noise(step, M, V, x.device, noise_fn)
        if skip and k == 0:
            return  # nothing is committed: the draw above was this step's only effect
        if not st.rows_valid and M > 0:
            st.forward_rows(M)
        if ctx.head is not None:
            conf, x0 = ctx.head.head_and_sample(st, M, temperature, top_p, top_k, conf_mode, noise, philox)
        else:
            conf, x0 = st.head_and_sample(M, temperature, top_p, top_k, conf_mode, noise, philox)
        if conf_mode == _hip.CONF_NEG_ENTROPY and temperature > 0 and noise_fn is None and ctx.lazy_noise:
            st.note_entropy(conf, M)
        if trace is not None:
            trace.setdefault("confidence", []).append(conf.clone())
            trace.setdefault("x0", []).append(x0.clone())
        if k > 0 and M > 0:
            if rank_temp > 0:
                # torch.multinomial(softmax(conf / alg_temp), k) == topk(p / q, k), q = one exponential_ fill of [B, N]
                if ctx.rank_noise_fn is not None:
                    q = ctx.rank_noise_fn(step, B, N).to(device=x.device, dtype=torch.float32).contiguous()
                else:
                    q = st.exponential_like_torch(B * N, philox_fill_plan(x.device, B * N))
                st.commit_sampled(x, mask_id, M, conf, x0, k, rank_temp, q)
            else:
                st.commit(x, mask_id, M, conf, x0, k)
            st.hidden_valid = False

    def _step_general(self, st, x, pos, key_mask, mask_id, temperature, top_p, top_k, alg, alg_temp, t, s, last, step,
                      tokens_hook, logits_hook, ctx: _LoopCtx) -> torch.Tensor:
        """Same iteration with materialised ``[B,N,V]`` logits for hooks, ``alg='origin'`` and
        ``alg_temp > 0`` (host-sequenced tensor ops around the same HIP entry points)."""
        B, N = x.shape
        V = self.config.vocab_size
        noise_fn, trace = ctx.noise_fn, ctx.trace
        mask_index = x == mask_id
        st.hidden_valid = False
        st.forward_hidden(x, pos, key_mask)
        logits = st.head_all_rows().view(B, N, V)
        logits = torch.cat([logits[:, :1], logits[:, :-1]], dim=1)
        logits = logits_hook(step, x, logits)
        mask_logits = logits[mask_index].contiguous().float()
        M = mask_logits.shape[0]
        if alg == "origin":
            p_transfer = 1 - s / t if not last else 1
            x0 = torch.zeros(M, device=x.device, dtype=torch.long) + mask_id
            u = torch.rand(M, device=x.device) if ctx.rand_fn is None else ctx.rand_fn(step, M).to(x.device)
            transfer = u < p_transfer.to(x.device) if isinstance(p_transfer, torch.Tensor) else u < p_transfer
            sel = mask_logits[transfer].contiguous()
            if sel.shape[0] > 0:
                noise = self._draw_noise(step, sel.shape[0], V, x.device, noise_fn) if temperature > 0 else None
                _, x0_sel = st.sample(sel, temperature, top_p, top_k, _hip.CONF_PROB, noise)
                x0[transfer] = x0_sel
            x[mask_index] = x0
        else:
            noise = self._draw_noise(step, M, V, x.device, noise_fn) if (temperature > 0 and M > 0) else None
            if M > 0:
                confidence, x0 = st.sample(mask_logits, temperature, top_p, top_k, _CONF_MODE[alg], noise)
            else:
                confidence = torch.empty(0, device=x.device)
                x0 = torch.empty(0, device=x.device, dtype=torch.long)
            Mg, Bg = (M, B) if ctx.count_reduce is None else ctx.count_reduce(mask_index.sum().reshape(1), B)
            k = transfer_count(Mg, Bg, t, s, last)
            if trace is not None:
                trace.setdefault("k", []).append(k)
                trace.setdefault("confidence", []).append(confidence.clone())
                trace.setdefault("x0", []).append(x0.clone())
            full_confidence = torch.full((B, N), -torch.inf, device=x.device, dtype=torch.float32)
            full_confidence[mask_index] = confidence
            if k > 0:
                if alg_temp is None or alg_temp == 0:
                    st.mask_rows(x, mask_id)
                    st.wait_count()
                    st.commit(x, mask_id, M, confidence, x0, k)
                else:
                    full_confidence = F.softmax(full_confidence / alg_temp, dim=-1)
                    transfer_index = torch.multinomial(full_confidence, num_samples=k)
                    x_ = torch.zeros_like(x) + mask_id
                    x_[mask_index] = x0
                    rows = torch.arange(B, device=x.device).unsqueeze(1).expand_as(transfer_index)
                    x[rows, transfer_index] = x_[rows, transfer_index]
        return tokens_hook(step, x, logits)
