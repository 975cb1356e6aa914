"""``DreamModel_length``: the bidirectional denoiser behind the reference's Python surface.

Mirrors ``akgr/abduction_model/model/custom_mydream/modeling_dream_length.py``:
``DreamModel_length(config)``, ``.from_pretrained(dir)`` (HF layout: ``config.json``, safetensors,
``generation_config.json``; :549-600), ``.forward(input_ids, attention_mask, position_ids)`` returning an
object with ``.logits [B,N,V]`` (:771-824), ``.diffusion_generate`` (via the mixin), ``.device``,
``.config``, ``.to()``, ``.eval()``.  The parameter tree has the reference's state_dict names
(SURVEY.md §5), so checkpoints load unchanged.

Parameters are only containers: every flop runs in ``libdark_hip.so``.  On first use on a device the
weights are packed once into the layout the kernels want (``PackedWeights``): q/k/v concatenated,
gate/up interleaved for the fused SwiGLU epilogue, rows zero-padded to the 128-wide GEMM tile,
RoPE cos/sin tables precomputed in fp32 the way DreamRotaryEmbedding.forward (:151-171) computes them.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _hip
from .configuration_dream import DreamConfig
from .generation_utils import DreamGenerationConfig, DreamGenerationMixin, DreamModelOutput  # noqa: F401

_TILE = 128


@dataclass
class MaskedLMOutput:
    """Stand-in for transformers.modeling_outputs.MaskedLMOutput (modeling_dream_length.py:819-824)."""
    logits: torch.Tensor = None
    loss: Optional[torch.Tensor] = None
    hidden_states: Optional[Tuple[torch.Tensor]] = None
    attentions: Optional[Tuple[torch.Tensor]] = None

    def __getitem__(self, i):
        return (self.logits,)[i]


def _pad_rows(w: torch.Tensor, mult: int = _TILE) -> torch.Tensor:
    r = (-w.shape[0]) % mult
    if r == 0:
        return w.contiguous()
    return torch.cat([w, w.new_zeros((r,) + tuple(w.shape[1:]))], dim=0).contiguous()


def gate_up_group(dtype: torch.dtype) -> int:
    """Interleave granularity of w_gu = the MFMA tile width of the GEMM kernel of that dtype."""
    return 16 if dtype == torch.bfloat16 else 32


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor, group: int = 32) -> torch.Tensor:
    """[ff,d],[ff,d] -> [2ff,d]: rows [2Gj, 2Gj+G) = gate[Gj:Gj+G], [2Gj+G, 2Gj+2G) = up[Gj:Gj+G], G = group
    (the fused SwiGLU epilogue wants gate and up of one output column in the same lane: adjacent MFMA tiles)."""
    ff, d = gate.shape
    assert ff % group == 0
    return torch.stack([gate.view(ff // group, group, d), up.view(ff // group, group, d)], dim=1).reshape(2 * ff, d)


def rope_tables(head_dim: int, theta: float, max_pos: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [max_pos, head_dim/2] fp32: inv_freq (x) position, as modeling_dream_length.py:156-165
    (the second half of `emb = cat(freqs, freqs)` is a copy, so only one half is stored)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    pos = torch.arange(max_pos, dtype=torch.int64)[None, :]
    freqs = (inv_freq[None, :, None].float() @ pos[:, None, :].float()).transpose(1, 2)[0]
    return freqs.cos().contiguous(), freqs.sin().contiguous()


class PackedWeights:
    """Device-resident, kernel-ready copy of the parameters + the C descriptor that points at it."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: DreamConfig, device: torch.device, dtype: torch.dtype,
                 residual_in_fp32: bool = False):
        assert dtype in (torch.float32, torch.bfloat16)
        self.device, self.dtype = device, dtype
        # element type of the residual stream: the model dtype (what the reference's bf16 model carries,
        # modeling_dream_length.py:508, :514), or fp32 for a bf16 model on request
        self.resid_dtype = torch.float32 if (residual_in_fp32 or dtype == torch.float32) else torch.bfloat16
        self.keep: List[torch.Tensor] = []

        def dev(t, dt=None):
            t = t.detach().to(device=device, dtype=dt or dtype).contiguous()
            self.keep.append(t)
            return t

        d, ff, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        if d % 64 or ff % 64:
            raise NotImplementedError(f"hidden_size={d} and intermediate_size={ff} must be multiples of 64")
        if cfg.head_dim not in (64, 96):
            raise NotImplementedError(f"head_dim={cfg.head_dim}: kernels exist for 64 and 96")
        self.max_pos = max(int(cfg.max_position_embeddings), 128)
        cos, sin = rope_tables(cfg.head_dim, float(cfg.rope_theta), self.max_pos)
        self.layers = (_hip.LayerWeights * max(L, 1))()
        for i in range(L):
            p = f"model.layers.{i}."
            a = p + "self_attn."
            w_qkv = torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0)
            b_qkv = torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0)
            lw = self.layers[i]
            lw.w_qkv = dev(_pad_rows(w_qkv)).data_ptr()
            lw.b_qkv = dev(_pad_rows(b_qkv), torch.float32).data_ptr()
            lw.w_o = dev(_pad_rows(sd[a + "o_proj.weight"])).data_ptr()
            lw.w_gu = dev(_pad_rows(interleave_gate_up(sd[p + "mlp.gate_proj.weight"],
                                                       sd[p + "mlp.up_proj.weight"],
                                                       gate_up_group(dtype)))).data_ptr()
            lw.w_down = dev(_pad_rows(sd[p + "mlp.down_proj.weight"])).data_ptr()
            lw.ln1 = dev(sd[p + "input_layernorm.weight"], torch.float32).data_ptr()
            lw.ln2 = dev(sd[p + "post_attention_layernorm.weight"], torch.float32).data_ptr()
        head = _pad_rows(sd["lm_head.weight"])
        self.desc = _hip.ModelDesc(
            dtype=_hip.DARK_BF16 if dtype == torch.bfloat16 else _hip.DARK_F32,
            vocab=cfg.vocab_size, vocab_pad=head.shape[0], d=d, ff=ff, n_layers=L,
            n_heads=cfg.num_attention_heads, head_dim=cfg.head_dim, max_pos=self.max_pos,
            rms_eps=float(cfg.rms_norm_eps),
            resid_dtype=_hip.DARK_BF16 if self.resid_dtype == torch.bfloat16 else _hip.DARK_F32,
            embed=dev(sd["model.embed_tokens.weight"]).data_ptr(),
            final_norm=dev(sd["model.norm.weight"], torch.float32).data_ptr(),
            lm_head=dev(head).data_ptr(),
            rope_cos=dev(cos, torch.float32).data_ptr(), rope_sin=dev(sin, torch.float32).data_ptr(),
            layers=C.cast(self.layers, C.POINTER(_hip.LayerWeights)))
        self.ref = C.byref(self.desc)
        self.elem = 2 if dtype == torch.bfloat16 else 4


class HipState:
    """Per-(B,N) device buffers and the launch helpers the sampler loop calls (all on torch's current
    stream).  Buffers are torch allocations: torch is the allocator and stream provider, nothing else."""

    def __init__(self, packed: PackedWeights, cfg: DreamConfig, B: int, N: int):
        self.lib = _hip.load()
        self.pk, self.cfg, self.B, self.N = packed, cfg, B, N
        dev = packed.device
        rows = B * N
        self.fwd_bytes = self._workspace_bytes(B, N)
        self.fwd_ws = torch.empty(self.fwd_bytes, dtype=torch.uint8, device=dev)
        self.hidden = torch.empty((rows, cfg.hidden_size), dtype=packed.resid_dtype, device=dev)
        self.row_index = torch.zeros(rows, dtype=torch.int32, device=dev)
        self.src_rows = torch.zeros(rows, dtype=torch.int32, device=dev)
        self.count = torch.zeros(4, dtype=torch.int32, device=dev)
        self.count_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        self.count_event = torch.cuda.Event()
        self.head_ws = None
        self.logits = None
        self.fuse_ws = None
        # head_and_sample through dark_head_sample_rows (same results; see its header): True / False, or "auto" = on
        # until the device-side statistics (read back without waiting) show that more than 30 % of the rows needed
        # their logits anyway (flat distributions, e.g. random-init weights), then off for the next 256 calls
        self.fused_head = "auto"
        self.head_stats = torch.zeros(2, dtype=torch.int32, device=dev)  # [rows sampled, rows that needed their logits]
        self.head_stats_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.head_stats_event = None
        # flat-row statistic of the lazy-noise policy (note_entropy): mean -confidence of the last sampled step
        self.flat_rows = False
        self.ent_host = torch.zeros(1, dtype=torch.float32).pin_memory()
        self.ent_event = None
        self._stats_seen = (0, 0)
        self._fuse_calls, self._fuse_off_until = 0, 0
        self.conf = torch.empty(rows, dtype=torch.float32, device=dev)
        self.x0 = torch.empty(rows, dtype=torch.int64, device=dev)
        self.hidden_rows = torch.empty((rows, cfg.hidden_size), dtype=packed.resid_dtype, device=dev)
        self.hidden_valid = False  # hidden / row_index / count describe the caller's current x
        self.rows_valid = False    # hidden_rows holds the last layer's output for src_rows[:M] of that x

    def _s(self) -> int:
        return _hip.stream_ptr(self.pk.device)

    def _workspace_bytes(self, B: int, N: int) -> int:
        return self.lib.dark_forward_workspace_bytes(self.pk.ref, B, N)

    def _ensure_head(self, M: int) -> None:
        if self.head_ws is None or self.head_ws.numel() < M * self.cfg.hidden_size * self.pk.elem + 256:
            self.head_ws = torch.empty(M * self.cfg.hidden_size * self.pk.elem + 256, dtype=torch.uint8,
                                       device=self.pk.device)
        ldl = (self.cfg.vocab_size + 3) & ~3
        if self.logits is None or self.logits.numel() < M * ldl:
            self.logits = torch.empty(M * ldl, dtype=torch.float32, device=self.pk.device)

    def forward_hidden(self, x, pos, key_mask) -> torch.Tensor:
        self.rows_valid = False
        _hip.check(self.lib.dark_forward_hidden(self.pk.ref, _hip.ptr(x), _hip.ptr(pos), _hip.ptr(key_mask),
                                                self.B, self.N, _hip.ptr(self.fwd_ws), self.fwd_bytes,
                                                _hip.ptr(self.hidden), self._s()))
        return self.hidden

    def forward_begin(self, x, pos, key_mask) -> None:
        """Everything of the body that does not depend on which rows the head will read (dark_forward_hidden_begin)."""
        _hip.check(self.lib.dark_forward_hidden_begin(self.pk.ref, _hip.ptr(x), _hip.ptr(pos), _hip.ptr(key_mask),
                                                      self.B, self.N, _hip.ptr(self.fwd_ws), self.fwd_bytes,
                                                      _hip.ptr(self.hidden), self._s()))
        self.rows_valid = False

    def forward_rows(self, M: int, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The rest of the last layer for src_rows[:M] (or the given int32 row list) only -> hidden_rows[:M]
        (dark_forward_hidden_rows)."""
        src = self.src_rows if rows is None else rows
        _hip.check(self.lib.dark_forward_hidden_rows(self.pk.ref, self.B, self.N, _hip.ptr(src), M,
                                                     _hip.ptr(self.fwd_ws), self.fwd_bytes, _hip.ptr(self.hidden),
                                                     _hip.ptr(self.hidden_rows), self._s()))
        self.rows_valid = rows is None   # head_and_sample may only reuse the rows of src_rows[:M]
        return self.hidden_rows

    def mask_rows(self, x, mask_id) -> None:
        _hip.check(self.lib.dark_mask_rows(_hip.ptr(x), self.B, self.N, int(mask_id), _hip.ptr(self.row_index),
                                           _hip.ptr(self.src_rows), _hip.ptr(self.count), self._s()))
        self.count_host.copy_(self.count, non_blocking=True)
        self.count_event.record(torch.cuda.current_stream(self.pk.device))

    def wait_count(self) -> int:
        self.count_event.synchronize()
        return int(self.count_host[0])

    def head_rows(self, src_rows: Optional[torch.Tensor], M: int, hidden: Optional[torch.Tensor] = None
                  ) -> torch.Tensor:
        V = self.cfg.vocab_size
        self._ensure_head(M)
        hidden = self.hidden if hidden is None else hidden
        _hip.check(self.lib.dark_lm_head_rows(self.pk.ref, _hip.ptr(hidden), _hip.ptr(src_rows), M,
                                              _hip.ptr(self.head_ws), self.head_ws.numel(), _hip.ptr(self.logits),
                                              V, self._s()))
        return self.logits[:M * V].view(M, V)

    def head_rows_padded(self, src_rows: Optional[torch.Tensor], M: int, hidden: Optional[torch.Tensor] = None):
        """head_rows for the sampler: rows on a stride of the next multiple of four floats, so that the head GEMM's and
        the sampler's 16-byte accesses apply to vocabularies like 25 427 and 41 066 too.  Returns (buffer, stride)."""
        V = self.cfg.vocab_size
        ldl = (V + 3) & ~3
        self._ensure_head(M)
        hidden = self.hidden if hidden is None else hidden
        _hip.check(self.lib.dark_lm_head_rows(self.pk.ref, _hip.ptr(hidden), _hip.ptr(src_rows), M,
                                              _hip.ptr(self.head_ws), self.head_ws.numel(), _hip.ptr(self.logits),
                                              ldl, self._s()))
        return self.logits, ldl

    def head_all_rows(self) -> torch.Tensor:
        return self.head_rows(None, self.B * self.N)

    def sample(self, logits: torch.Tensor, temperature, top_p, top_k, conf_mode, noise):
        M, V = logits.shape
        conf = torch.empty(M, dtype=torch.float32, device=logits.device)
        x0 = torch.empty(M, dtype=torch.int64, device=logits.device)
        _hip.check(self.lib.dark_sample_rows(_hip.ptr(logits), logits.stride(0), M, V, float(temperature),
                                             float(top_p), int(top_k), int(conf_mode), _hip.ptr(noise),
                                             noise.stride(0) if noise is not None else 0, _hip.ptr(conf),
                                             _hip.ptr(x0), self._s()))
        return conf, x0

    def head_and_sample(self, M, temperature, top_p, top_k, conf_mode, noise, philox=None):
        """`philox` = (seed, offset, nthreads) from generation_utils.philox_fill_plan: draws computed in the kernel.
        Reads hidden_rows[:M] (forward_begin + forward_rows) when it is current, else hidden[src_rows[:M]]."""
        V = self.cfg.vocab_size
        conf, x0 = self.conf[:M], self.x0[:M]
        if self._use_fused_head():
            # dark_head_sample_rows: same conf / x0; logits are only materialised for the rows its records cannot finish
            self._ensure_head(M)
            ldl = (V + 3) & ~3   # 16-byte rows for the fallback GEMM's stores
            if self.logits.numel() < M * ldl:
                self.logits = torch.empty(M * ldl, dtype=torch.float32, device=self.pk.device)
            need = self.lib.dark_head_sample_workspace_bytes(self.pk.ref, M)
            if self.fuse_ws is None or self.fuse_ws.numel() < need:
                self.fuse_ws = torch.empty(need, dtype=torch.uint8, device=self.pk.device)
            hidden, src = (self.hidden_rows, None) if self.rows_valid else (self.hidden, self.src_rows)
            seed, off, nth = philox if philox is not None else (0, 0, 0)
            _hip.check(self.lib.dark_head_sample_rows(
                self.pk.ref, _hip.ptr(hidden), _hip.ptr(src), M, _hip.ptr(self.fuse_ws), self.fuse_ws.numel(),
                _hip.ptr(self.logits), ldl, float(temperature), float(top_p), int(top_k), int(conf_mode),
                _hip.ptr(noise) if philox is None else None,
                noise.stride(0) if (noise is not None and philox is None) else 0, seed, off, nth, _hip.ptr(conf),
                _hip.ptr(x0), _hip.ptr(self.head_stats), self._s()))
            if self.fused_head == "auto" and self.head_stats_event is None:
                self.head_stats_host.copy_(self.head_stats, non_blocking=True)
                self.head_stats_event = torch.cuda.Event()
                self.head_stats_event.record(torch.cuda.current_stream(self.pk.device))
            return conf, x0
        logits, ldl = (self.head_rows_padded(None, M, self.hidden_rows) if self.rows_valid
                       else self.head_rows_padded(self.src_rows, M))
        if philox is not None:
            _hip.check(self.lib.dark_sample_rows_philox(_hip.ptr(logits), ldl, M, V, float(temperature), float(top_p),
                                                        int(top_k), int(conf_mode), philox[0], philox[1], philox[2],
                                                        _hip.ptr(conf), _hip.ptr(x0), self._s()))
            return conf, x0
        _hip.check(self.lib.dark_sample_rows(_hip.ptr(logits), ldl, M, V, float(temperature), float(top_p),
                                             int(top_k), int(conf_mode), _hip.ptr(noise),
                                             noise.stride(0) if noise is not None else 0, _hip.ptr(conf),
                                             _hip.ptr(x0), self._s()))
        return conf, x0

    # exp(mean entropy) above which the Exp(1) noise is materialised instead of evaluated inside the sampler: the lazy
    # form costs one Philox4x32-10 call per KEPT token (a row as flat as T = 0.5 makes random-init logits keeps ~9 000
    # of 15 080: 2.41 ms per 16 896 rows against 1.66 ms + a 0.11 ms fill, profiles/r03r_sampler_flat_rows_phases.txt),
    # the materialised form one [M, V] fill and read whatever is kept; break-even is near 2 000 kept tokens a row
    FLAT_ROWS_KEPT = 2000.0

    def note_entropy(self, conf: torch.Tensor, M: int) -> None:
        """Policy statistic only (no result depends on it: both noise forms give the same bits): the mean confidence of
        an entropy-mode step is read back without waiting; `flat_rows` follows one or two steps behind."""
        ev = self.ent_event
        if ev is not None:
            if not ev.query():
                return
            self.flat_rows = float(-self.ent_host[0]) > math.log(self.FLAT_ROWS_KEPT)
            self.ent_event = None
        if M > 0:
            self.ent_host.copy_(conf[:M].mean().reshape(1), non_blocking=True)
            self.ent_event = torch.cuda.Event()
            self.ent_event.record(torch.cuda.current_stream(self.pk.device))

    def _use_fused_head(self) -> bool:
        if self.fused_head != "auto":
            return bool(self.fused_head)
        self._fuse_calls += 1
        ev = self.head_stats_event
        if ev is not None and ev.query():   # never waits: a read-back that has not landed is looked at next time
            rows, fb = int(self.head_stats_host[0]), int(self.head_stats_host[1])
            d_rows, d_fb = rows - self._stats_seen[0], fb - self._stats_seen[1]
            self._stats_seen, self.head_stats_event = (rows, fb), None
            if d_rows > 0 and d_fb > 0.3 * d_rows:
                self._fuse_off_until = self._fuse_calls + 256
        return self._fuse_calls >= self._fuse_off_until

    def sample_into(self, logits, n, temperature, top_p, top_k, conf_mode, noise, philox):
        """dark_sample_rows(_philox) on `logits` [n, V] into the state's conf / x0 buffers."""
        V = self.cfg.vocab_size
        conf, x0 = self.conf[:n], self.x0[:n]
        if philox is not None:
            _hip.check(self.lib.dark_sample_rows_philox(_hip.ptr(logits), V, n, V, float(temperature), float(top_p),
                                                        int(top_k), int(conf_mode), philox[0], philox[1], philox[2],
                                                        _hip.ptr(conf), _hip.ptr(x0), self._s()))
        else:
            _hip.check(self.lib.dark_sample_rows(_hip.ptr(logits), V, n, V, float(temperature), float(top_p),
                                                 int(top_k), int(conf_mode), _hip.ptr(noise),
                                                 noise.stride(0) if noise is not None else 0, _hip.ptr(conf),
                                                 _hip.ptr(x0), self._s()))
        return conf, x0

    def scatter(self, x, M, sel, tokens, n) -> None:
        _hip.check(self.lib.dark_scatter_tokens(_hip.ptr(x), self.B, self.N, _hip.ptr(self.row_index), M,
                                                _hip.ptr(sel.contiguous()), _hip.ptr(tokens.contiguous()), n, self._s()))

    def exponential_like_torch(self, numel: int, philox) -> torch.Tensor:
        """What `torch.empty(numel, device=...).exponential_()` would hold for the reserved Philox state."""
        out = torch.empty(numel, dtype=torch.float32, device=self.pk.device)
        _hip.check(self.lib.dark_philox_exponential(philox[0], philox[1], philox[2], numel, _hip.ptr(out), self._s()))
        return out

    def commit_sampled(self, x, mask_id, M, conf, x0, k, alg_temp, q) -> None:
        _hip.check(self.lib.dark_commit_sampled(_hip.ptr(x), self.B, self.N, int(mask_id), _hip.ptr(self.row_index), M,
                                                _hip.ptr(conf.contiguous()), _hip.ptr(x0.contiguous()), int(k),
                                                float(alg_temp), _hip.ptr(q), self._s()))

    def commit(self, x, mask_id, M, conf, x0, k) -> None:
        _hip.check(self.lib.dark_commit_topk(_hip.ptr(x), self.B, self.N, int(mask_id), _hip.ptr(self.row_index),
                                             M, _hip.ptr(conf.contiguous()), _hip.ptr(x0.contiguous()), int(k),
                                             self._s()))


class _Linear(nn.Module):
    """Parameter container with nn.Linear's state_dict names."""

    def __init__(self, fin, fout, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(fout), requires_grad=False) if bias else None


class _Weight(nn.Module):
    def __init__(self, *shape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape), requires_grad=False)


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj = _Linear(d, d, True), _Linear(d, d, True), _Linear(d, d, True)
        self.o_proj = _Linear(d, d, False)


class _MLP(nn.Module):
    def __init__(self, d, ff):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = _Linear(d, ff, False), _Linear(d, ff, False), _Linear(ff, d, False)


class _Layer(nn.Module):
    def __init__(self, d, ff):
        super().__init__()
        self.self_attn, self.mlp = _Attn(d), _MLP(d, ff)
        self.input_layernorm, self.post_attention_layernorm = _Weight(d), _Weight(d)


class _Base(nn.Module):
    def __init__(self, cfg: DreamConfig):
        super().__init__()
        self.embed_tokens = _Weight(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([_Layer(cfg.hidden_size, cfg.intermediate_size)
                                     for _ in range(cfg.num_hidden_layers)])
        self.norm = _Weight(cfg.hidden_size)


class DreamModel_length(DreamGenerationMixin, nn.Module):
    """modeling_dream_length.py:736-824 (inference surface)."""
    config_class = DreamConfig
    generation_config_class = DreamGenerationConfig
    # bf16 models: carry the residual stream in fp32 instead of the model dtype (no counterpart in the reference,
    # whose bf16 model adds bf16 tensors; costs ~5 % of the denoise step).  Set on the instance
    # or the class before the first call; fp32 models always use fp32.
    residual_in_fp32 = False

    def __init__(self, config: DreamConfig):
        super().__init__()
        self.config = config
        self.vocab_size = config.vocab_size
        self.model = _Base(config)
        self.lm_head = _Linear(config.hidden_size, config.vocab_size, False)
        self.generation_config = self.generation_config_class.from_model_config(config)
        self._packed: Optional[PackedWeights] = None
        self._packed_key = None
        self._states: Dict[Tuple[int, int], HipState] = {}
        self._init_weights()
        self.eval()

    def _init_weights(self) -> None:
        """modeling_dream_length.py:538-547: normal(0, initializer_range), zero biases/pad row, unit norms."""
        std = self.config.initializer_range
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    p.zero_()
                else:
                    p.normal_(mean=0.0, std=std)
            pad = self.config.pad_token_id
            if pad is not None and 0 <= pad < self.config.vocab_size:
                self.model.embed_tokens.weight[pad].zero_()

    # ---- HF-style surface
    @property
    def device(self) -> torch.device:
        return self.lm_head.weight.device

    @property
    def dtype(self) -> torch.dtype:
        return self.lm_head.weight.dtype

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    @classmethod
    def from_pretrained(cls, path: str, *model_args, config: Optional[DreamConfig] = None, torch_dtype=None,
                        **kwargs) -> "DreamModel_length":
        """modeling_dream_length.py:549-600: weights + DreamGenerationConfig from a local HF directory."""
        from safetensors.torch import load_file
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local directory (hub downloads are not available)")
        cfg = config or DreamConfig.from_pretrained(path)
        model = cls(cfg)
        index = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(index):   # sharded by HF's writer: the shards its weight_map names, nothing else
            with open(index) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            absent = [f for f in files if not os.path.exists(os.path.join(path, f))]
            if absent:
                raise OSError(f"{index} names shards that are not in {path}: {absent}")
        elif os.path.exists(os.path.join(path, "model.safetensors")):
            files = ["model.safetensors"]
        else:
            files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if not files:
            raise OSError(f"no .safetensors file in {path}")
        sd: Dict[str, torch.Tensor] = {}
        for f in files:
            sd.update(load_file(os.path.join(path, f)))
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing:
            raise RuntimeError(f"checkpoint is missing weights: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        if torch_dtype is not None:
            model.to(torch_dtype)
        if os.path.exists(os.path.join(path, "generation_config.json")):
            model.generation_config = cls.generation_config_class.from_pretrained(path)
        return model

    def save_pretrained(self, path: str) -> None:
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        self.config.save_pretrained(path)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"))
        self.generation_config.save_pretrained(path)

    # ---- packing / device state
    def _weights_key(self):
        return (str(self.device), self.dtype, bool(self.residual_in_fp32), tuple(p._version for p in self.parameters()),
                tuple(p.data_ptr() for p in self.parameters()))

    def packed(self) -> PackedWeights:
        if self.device.type != "cuda":
            raise RuntimeError("dark_amd runs on a HIP device only: move the model with .to('cuda')")
        key = self._weights_key()
        if self._packed is None or self._packed_key != key:
            dt = torch.bfloat16 if self.dtype == torch.bfloat16 else torch.float32
            self._packed = PackedWeights(self.state_dict(), self.config, self.device, dt, bool(self.residual_in_fp32))
            self._packed_key = key
            self._states.clear()
        return self._packed

    def _hip_state(self, B: int, N: int) -> HipState:
        pk = self.packed()
        st = self._states.get((B, N))
        if st is None:
            if len(self._states) >= 4:  # keep the cache small: buffers are O(B*N*ff)
                self._states.clear()
            st = self._states[(B, N)] = HipState(pk, self.config, B, N)
        return st

    # ---- forward
    @torch.no_grad()
    def forward(self, input_ids: torch.LongTensor = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, num_logits_to_keep: int = 0, **kwargs
                ) -> MaskedLMOutput:
        """modeling_dream_length.py:771-824.  `attention_mask`: bool [B,1,N,N] built as the outer AND of a
        pad vector (what every caller of the reference passes, generation_utils.py:469-472), a [B,N] pad
        vector, or None / non-tensor for no masking (:428)."""
        if input_ids is None:
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")  # :654-655
        if kwargs.get("inputs_embeds") is not None or kwargs.get("labels") is not None:
            raise NotImplementedError("inputs_embeds / labels are outside the inference hot path")
        if self.device.type != "cuda":
            raise RuntimeError("dark_amd runs on a HIP device only: move the model with .to('cuda')")
        with torch.cuda.device(self.device):  # the library launches on the CURRENT device's stream
            return self._forward_impl(input_ids, attention_mask, position_ids, num_logits_to_keep)

    def _forward_impl(self, input_ids, attention_mask, position_ids, num_logits_to_keep) -> MaskedLMOutput:
        # the kernels read int64 ids; nn.Embedding in the reference also takes int32 (:665)
        x = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, N = x.shape
        key_mask = None
        if isinstance(attention_mask, torch.Tensor):
            am = attention_mask.to(self.device)
            if am.dim() == 4:
                vec = torch.diagonal(am[:, 0], dim1=-2, dim2=-1).bool()
                if not bool(torch.equal(am[:, 0].bool(), vec.unsqueeze(-1) & vec.unsqueeze(-2))):
                    raise NotImplementedError("only pad-style masks (outer AND of a [B,N] vector) are supported")
                am = vec
            if not bool(am.bool().all()):
                key_mask = am.bool().to(torch.uint8).contiguous()
        pos = None
        if position_ids is not None:
            pos = position_ids.to(self.device).expand(B, N).to(torch.int32).contiguous()
        st = self._hip_state(B, N)
        st.hidden_valid = False
        st.forward_hidden(x, pos, key_mask)
        logits = st.head_all_rows().view(B, N, self.config.vocab_size)
        if num_logits_to_keep:
            logits = logits[:, -num_logits_to_keep:, :]
        # the reference's lm_head returns the model's dtype (modeling_dream_length.py:809: bf16 logits from a bf16 model,
        # e.g. under the autocast of custom_grpo2.py:409-418); the head GEMM here accumulates and stores fp32
        return MaskedLMOutput(logits=logits.to(self.dtype) if self.dtype == torch.bfloat16 else logits.clone())
