"""GPT-2 backbone of the legacy diffusion denoiser.

Mirrors ``akgr/abduction_model/model/custom_gpt.py``: ``DiffusionModel`` (:25-114, a ``GPT2LMHeadModel`` whose forward
takes ``(input_ids, attention_mask [B, N], position_ids)`` and returns ``.logits``), its variable-length
``diffusion_generate`` / ``_sample`` (:240-444 -- the same EXPAND / DELETE loop as
``custom_mydream/generation_expand_utils.py``, with the window passed as a key-padding mask) and
``create_gpt2_transformer`` (:446-472).  The parameter tree carries transformers' GPT-2 ``state_dict`` names
(``transformer.wte.weight``, ``transformer.h.0.attn.c_attn.weight`` [in, out] Conv1D layout, ..., ``lm_head.weight``
tied to ``wte``), so a checkpoint of the reference class loads unchanged.

Compute: ``dark_gpt2_forward_hidden`` / ``dark_gpt2_lm_head_rows`` (``csrc/gpt2.hip``) + the sampler, commit and edit
kernels of the Dream path.  No CPU or PyTorch fallback.

Attention semantics: the reference sets every block's ``attn.bias`` buffer to ones "to allow bidirectional attention"
(:34-36), but the pinned transformers 4.50.2 selects its sdpa attention, which never reads that buffer and builds the
causal mask itself (5.x has no such buffer at all): what the reference COMPUTES, trains and evaluates with is causal
attention.  ``attn_causal=True`` (the default) is therefore the reference's behaviour -- the goldens of the reference's
own class are reproduced with it -- and ``attn_causal=False`` is the explicit opt-in for the bidirectional attention
the comment states as the intent.  Both are pinned against transformers' own GPT-2
(``tests/golden/make_goldens_gpt2.py``).
"""
from __future__ import annotations

import copy
import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from . import _hip
from .generation_expand_utils import DreamGenerationConfig, DreamGenerationMixin, DreamModelOutput  # noqa: F401
from .modeling_dream_length import HipState, _pad_rows


class GPT2DiffusionConfig:
    """The fields of ``transformers.GPT2Config`` this backbone reads, same names and defaults (the stock ``gpt2``
    config the reference starts from, custom_gpt.py:462-465), plus the special-token ids create_gpt2_transformer
    overwrites (:451-458).  Any object with these attributes (e.g. a real ``GPT2Config``) is accepted by the model."""
    model_type = "gpt2"

    def __init__(self, vocab_size=50257, n_positions=1024, n_embd=768, n_layer=12, n_head=12, n_inner=None,
                 activation_function="gelu_new", layer_norm_epsilon=1e-5, initializer_range=0.02,
                 bos_token_id=50256, eos_token_id=50256, pad_token_id=None, attn_causal=True, **kwargs):
        self.vocab_size, self.n_positions, self.n_embd = vocab_size, n_positions, n_embd
        self.n_layer, self.n_head, self.n_inner = n_layer, n_head, n_inner
        self.activation_function = activation_function
        self.layer_norm_epsilon = layer_norm_epsilon
        self.initializer_range = initializer_range
        self.bos_token_id, self.eos_token_id, self.pad_token_id = bos_token_id, eos_token_id, pad_token_id
        self.attn_causal = bool(attn_causal)
        self.use_cache = False
        for k, v in kwargs.items():   # unk_token_id, sep_token_id, mask_token_id, ... (HF configs keep unknown keys)
            setattr(self, k, v)

    # GPT2Config's attribute_map
    hidden_size = property(lambda self: self.n_embd)
    max_position_embeddings = property(lambda self: self.n_positions)
    num_attention_heads = property(lambda self: self.n_head)
    num_hidden_layers = property(lambda self: self.n_layer)

    def to_dict(self) -> Dict[str, Any]:
        d = {k: copy.deepcopy(v) for k, v in self.__dict__.items() if not k.startswith("_")}
        d["model_type"] = self.model_type
        return d

    @classmethod
    def from_pretrained(cls, path: str, **overrides) -> "GPT2DiffusionConfig":
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        for k in ("model_type", "architectures", "transformers_version", "torch_dtype", "dtype"):
            d.pop(k, None)
        d.update(overrides)
        return cls(**d)

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)


def _check_supported(cfg) -> None:
    act = getattr(cfg, "activation_function", "gelu_new")
    if act != "gelu_new":
        raise NotImplementedError(f"activation_function={act!r}; only 'gelu_new' (the stock gpt2 config)")
    if cfg.n_embd % cfg.n_head or (cfg.n_embd // cfg.n_head) not in (64, 96):
        raise NotImplementedError("head_dim must be 64 or 96")
    inner = getattr(cfg, "n_inner", None) or 4 * cfg.n_embd
    if cfg.n_embd % 64 or inner % 64:
        raise NotImplementedError("n_embd and n_inner must be multiples of 64")
    for flag in ("scale_attn_by_inverse_layer_idx", "reorder_and_upcast_attn", "add_cross_attention"):
        if getattr(cfg, flag, False):
            raise NotImplementedError(flag)
    if not getattr(cfg, "scale_attn_weights", True):
        raise NotImplementedError("scale_attn_weights=False")


class PackedGpt2Weights:
    """Device-resident, kernel-ready copy of the GPT-2 parameters + the C descriptor that points at it."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg, device: torch.device, dtype: torch.dtype,
                 residual_in_fp32: bool = False):
        assert dtype in (torch.float32, torch.bfloat16)
        self.device, self.dtype = device, dtype
        self.resid_dtype = torch.float32 if (residual_in_fp32 or dtype == torch.float32) else torch.bfloat16
        self.keep: List[torch.Tensor] = []

        def dev(t, dt=None):
            t = t.detach().to(device=device, dtype=dt or dtype).contiguous()
            self.keep.append(t)
            return t

        d, L, H = cfg.n_embd, cfg.n_layer, cfg.n_head
        ff = getattr(cfg, "n_inner", None) or 4 * d
        hd = d // H
        self.layers = (_hip.Gpt2LayerWeights * max(L, 1))()
        f32 = torch.float32
        for i in range(L):
            p = f"transformer.h.{i}."
            lw = self.layers[i]
            # Conv1D stores [in, out] and computes x @ W + b: the kernels want [out, in] (K-contiguous)
            lw.w_qkv = dev(_pad_rows(sd[p + "attn.c_attn.weight"].t())).data_ptr()
            lw.b_qkv = dev(sd[p + "attn.c_attn.bias"], f32).data_ptr()
            lw.w_o = dev(_pad_rows(sd[p + "attn.c_proj.weight"].t())).data_ptr()
            lw.b_o = dev(sd[p + "attn.c_proj.bias"], f32).data_ptr()
            lw.w_fc = dev(_pad_rows(sd[p + "mlp.c_fc.weight"].t())).data_ptr()
            lw.b_fc = dev(sd[p + "mlp.c_fc.bias"], f32).data_ptr()
            lw.w_proj = dev(_pad_rows(sd[p + "mlp.c_proj.weight"].t())).data_ptr()
            lw.b_proj = dev(sd[p + "mlp.c_proj.bias"], f32).data_ptr()
            lw.ln1_w, lw.ln1_b = dev(sd[p + "ln_1.weight"], f32).data_ptr(), dev(sd[p + "ln_1.bias"], f32).data_ptr()
            lw.ln2_w, lw.ln2_b = dev(sd[p + "ln_2.weight"], f32).data_ptr(), dev(sd[p + "ln_2.bias"], f32).data_ptr()
        wte = _pad_rows(sd["transformer.wte.weight"])
        self.desc = _hip.Gpt2Desc(
            dtype=_hip.DARK_BF16 if dtype == torch.bfloat16 else _hip.DARK_F32,
            vocab=cfg.vocab_size, vocab_pad=wte.shape[0], d=d, ff=ff, n_layers=L, n_heads=H, head_dim=hd,
            n_positions=cfg.n_positions, ln_eps=float(cfg.layer_norm_epsilon),
            resid_dtype=_hip.DARK_BF16 if self.resid_dtype == torch.bfloat16 else _hip.DARK_F32,
            wte=dev(wte).data_ptr(), wpe=dev(sd["transformer.wpe.weight"]).data_ptr(),
            lnf_w=dev(sd["transformer.ln_f.weight"], f32).data_ptr(),
            lnf_b=dev(sd["transformer.ln_f.bias"], f32).data_ptr(),
            ident_cos=dev(torch.ones(hd // 2), f32).data_ptr(), ident_sin=dev(torch.zeros(hd // 2), f32).data_ptr(),
            layers=C.cast(self.layers, C.POINTER(_hip.Gpt2LayerWeights)))
        self.ref = C.byref(self.desc)
        self.elem = 2 if dtype == torch.bfloat16 else 4


@dataclass
class _StateShape:
    """What HipState reads from a model config."""
    hidden_size: int
    vocab_size: int


class Gpt2HipState(HipState):
    """HipState of the Dream path with the body and the head swapped for the GPT-2 entry points; row lists,
    confidence / token buffers, sampler, commit and edit helpers are inherited."""

    def __init__(self, packed: PackedGpt2Weights, cfg, B: int, N: int, causal: bool):
        self.causal = int(bool(causal))
        super().__init__(packed, _StateShape(cfg.n_embd, cfg.vocab_size), B, N)

    def _workspace_bytes(self, B: int, N: int) -> int:
        return self.lib.dark_gpt2_forward_workspace_bytes(self.pk.ref, B, N)

    def forward_hidden(self, x, pos, key_mask) -> torch.Tensor:
        self.rows_valid = False
        _hip.check(self.lib.dark_gpt2_forward_hidden(self.pk.ref, _hip.ptr(x), _hip.ptr(pos), _hip.ptr(key_mask),
                                                     self.causal, self.B, self.N, _hip.ptr(self.fwd_ws),
                                                     self.fwd_bytes, _hip.ptr(self.hidden), self._s()))
        return self.hidden

    # the loop's two-call form of the body: here the whole body runs in the first call and the head gathers its rows
    def forward_begin(self, x, pos, key_mask) -> None:
        self.forward_hidden(x, pos, key_mask)

    def forward_rows(self, M: int, rows: Optional[torch.Tensor] = None):
        self._rows = self.src_rows if rows is None else rows
        return None

    def head_rows(self, src_rows: Optional[torch.Tensor], M: int, hidden: Optional[torch.Tensor] = None
                  ) -> torch.Tensor:
        V = self.cfg.vocab_size
        self._ensure_head(M)
        if hidden is None and src_rows is None and getattr(self, "_rows", None) is not None:
            src_rows = self._rows       # head_rows(None, M, forward_rows(M)): the rows forward_rows named
        self._rows = None
        _hip.check(self.lib.dark_gpt2_lm_head_rows(self.pk.ref, _hip.ptr(self.hidden if hidden is None else hidden),
                                                   _hip.ptr(src_rows), M, _hip.ptr(self.head_ws),
                                                   self.head_ws.numel(), _hip.ptr(self.logits), V, self._s()))
        return self.logits[:M * V].view(M, V)

    def head_and_sample(self, *a, **k):
        raise NotImplementedError("the GPT-2 backbone is driven by the variable-length sampler only")


class CausalLMOutput:
    """The two fields of ``transformers.modeling_outputs.CausalLMOutput`` the callers read (custom_gpt.py:110-114)."""

    def __init__(self, logits):
        self.logits, self.hidden_states, self.attentions = logits, None, None

    def __getitem__(self, k):
        return self.logits if k in (0, "logits") else None


class _Conv1D(nn.Module):
    """transformers.pytorch_utils.Conv1D parameter container: weight [in, out], bias [out]."""

    def __init__(self, nf, nx):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(nx, nf), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(nf), requires_grad=False)


class _LN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(d), requires_grad=False)


class _Emb(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d), requires_grad=False)


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_attn, self.c_proj = _Conv1D(3 * d, d), _Conv1D(d, d)


class _MLP(nn.Module):
    def __init__(self, d, inner):
        super().__init__()
        self.c_fc, self.c_proj = _Conv1D(inner, d), _Conv1D(d, inner)


class _Block(nn.Module):
    def __init__(self, d, inner):
        super().__init__()
        self.ln_1, self.attn, self.ln_2, self.mlp = _LN(d), _Attn(d), _LN(d), _MLP(d, inner)


class _Transformer(nn.Module):
    def __init__(self, cfg, inner):
        super().__init__()
        self.wte, self.wpe = _Emb(cfg.vocab_size, cfg.n_embd), _Emb(cfg.n_positions, cfg.n_embd)
        self.h = nn.ModuleList([_Block(cfg.n_embd, inner) for _ in range(cfg.n_layer)])
        self.ln_f = _LN(cfg.n_embd)


class _Head(nn.Module):
    def __init__(self, weight):
        super().__init__()
        self.weight = weight   # the SAME Parameter as transformer.wte.weight (tie_word_embeddings)


class DiffusionModel(DreamGenerationMixin, nn.Module):
    """custom_gpt.py:25-114 (inference surface) + :240-444 (`diffusion_generate` / `_sample`, inherited: the batched
    EXPAND / DELETE loop of dark_amd.generation_expand_utils)."""
    config_class = GPT2DiffusionConfig
    generation_config_class = DreamGenerationConfig
    residual_in_fp32 = False   # bf16 models: fp32 residual stream on request (see DreamModel_length.residual_in_fp32)

    def __init__(self, config):
        super().__init__()
        _check_supported(config)
        self.config = config
        inner = getattr(config, "n_inner", None) or 4 * config.n_embd
        self.transformer = _Transformer(config, inner)
        self.lm_head = _Head(self.transformer.wte.weight)
        self.config.use_cache = False                                     # :39
        self.attn_causal = bool(getattr(config, "attn_causal", True))
        self.generation_config = self.generation_config_class.from_model_config(config)
        self._packed: Optional[PackedGpt2Weights] = None
        self._packed_key = None
        self._states: Dict[Any, Gpt2HipState] = {}
        self._init_weights()
        self.eval()

    def _init_weights(self) -> None:
        """GPT2PreTrainedModel._init_weights: normal(0, initializer_range) for Linear / Conv1D / Embedding weights,
        zero biases, unit LayerNorm weights, and c_proj weights rescaled by 1/sqrt(2 * n_layer)."""
        std = float(self.config.initializer_range)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if ".ln_" in name:   # ln_1 / ln_2 / ln_f
                    p.fill_(1.0) if name.endswith("weight") else p.zero_()
                elif name.endswith(".bias"):
                    p.zero_()
                elif name.endswith("c_proj.weight"):
                    p.normal_(mean=0.0, std=std / (2 * self.config.n_layer) ** 0.5)
                else:
                    p.normal_(mean=0.0, std=std)

    # ---- HF-style surface
    @property
    def device(self) -> torch.device:
        return self.transformer.wte.weight.device

    @property
    def dtype(self) -> torch.dtype:
        return self.transformer.wte.weight.dtype

    def get_input_embeddings(self):
        return self.transformer.wte                                       # :42-44

    def get_output_embeddings(self):
        return self.lm_head                                               # :50-52

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = dict(state_dict)
        # causal-mask buffers of older transformers checkpoints are not parameters here
        sd = {k: v for k, v in sd.items() if not (k.endswith(".attn.bias") or k.endswith(".attn.masked_bias"))}
        if "lm_head.weight" not in sd and "transformer.wte.weight" in sd:
            sd["lm_head.weight"] = sd["transformer.wte.weight"]         # tied: safetensors keeps one copy
        return super().load_state_dict(sd, strict=strict, **kw)

    @classmethod
    def from_pretrained(cls, path: str, config=None, torch_dtype=None, **kwargs) -> "DiffusionModel":
        from safetensors.torch import load_file
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local directory (hub downloads are not available)")
        model = cls(config or GPT2DiffusionConfig.from_pretrained(path, **kwargs))
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if not files:
            raise OSError(f"no .safetensors file in {path}")
        sd: Dict[str, torch.Tensor] = {}
        for f in files:
            sd.update(load_file(os.path.join(path, f)))
        missing, _ = model.load_state_dict(sd, strict=False)
        if missing:
            raise RuntimeError(f"checkpoint is missing weights: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        if torch_dtype is not None:
            model.to(torch_dtype)
        return model

    def save_pretrained(self, path: str) -> None:
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        self.config.save_pretrained(path)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items() if k != "lm_head.weight"}
        save_file(sd, os.path.join(path, "model.safetensors"))

    # ---- packing / device state
    def packed(self) -> PackedGpt2Weights:
        if self.device.type != "cuda":
            raise RuntimeError("dark_amd runs on a HIP device only: move the model with .to('cuda')")
        key = (str(self.device), self.dtype, bool(self.residual_in_fp32), tuple(p._version for p in self.parameters()),
               tuple(p.data_ptr() for p in self.parameters()))
        if self._packed is None or self._packed_key != key:
            dt = torch.bfloat16 if self.dtype == torch.bfloat16 else torch.float32
            self._packed = PackedGpt2Weights(self.state_dict(), self.config, self.device, dt,
                                             bool(self.residual_in_fp32))
            self._packed_key = key
            self._states.clear()
        return self._packed

    def _hip_state(self, B: int, N: int) -> Gpt2HipState:
        pk = self.packed()
        key = (B, N, self.attn_causal)
        st = self._states.get(key)
        if st is None:
            if len(self._states) >= 4:
                self._states.clear()
            st = self._states[key] = Gpt2HipState(pk, self.config, B, N, self.attn_causal)
        return st

    # ---- forward (custom_gpt.py:58-114)
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=True):
        if input_ids is None or inputs_embeds is not None or labels is not None:
            raise NotImplementedError("input_ids only: inputs_embeds / labels are outside the inference hot path")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("attention maps / hidden states are not materialised")
        if self.device.type != "cuda":
            raise RuntimeError("dark_amd runs on a HIP device only: move the model with .to('cuda')")
        with torch.cuda.device(self.device):
            x = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
            B, N = x.shape
            key_mask = None
            if attention_mask is not None:                                # default: all ones (:82-83)
                am = attention_mask.to(self.device)
                if am.dim() != 2:
                    raise NotImplementedError("attention_mask must be the [B, N] padding mask the sampler passes")
                if not bool(am.bool().all()):
                    key_mask = am.bool().to(torch.uint8).contiguous()
            pos = None
            if position_ids is not None:
                pos = position_ids.to(self.device).expand(B, N).to(torch.int32).contiguous()
            st = self._hip_state(B, N)
            st.hidden_valid = False
            st.forward_hidden(x, pos, key_mask)
            logits = st.head_rows(None, B * N, st.hidden).view(B, N, self.config.vocab_size).clone()
        return CausalLMOutput(logits) if return_dict else (logits,)


def create_gpt2_transformer(ntoken: int, tokenizer, model_name: str, config_dir: Optional[str] = None,
                            **config_overrides) -> DiffusionModel:
    """custom_gpt.py:446-472: the stock `gpt2` shape (or the config.json in `config_dir` -- the reference reads an
    author-local directory, :462-465) with `vocab_size = ntoken + 1` and the tokenizer's special-token ids."""
    common = dict(vocab_size=ntoken + 1, pad_token_id=tokenizer.pad_token_id, bos_token_id=tokenizer.bos_token_id,
                  eos_token_id=tokenizer.eos_token_id, unk_token_id=tokenizer.unk_token_id,
                  sep_token_id=tokenizer.sep_token_id)
    common.update(config_overrides)
    cfg = GPT2DiffusionConfig.from_pretrained(config_dir, **common) if config_dir else GPT2DiffusionConfig(**common)
    model = DiffusionModel(cfg)
    model.model_name = model_name                                         # :471
    return model
