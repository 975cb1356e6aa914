"""``DreamConfig``: shape contract of the denoiser.

Mirrors the reference's ``DreamConfig`` (akgr/abduction_model/model/custom_mydream/
configuration_dream.py:25-86: same keyword names and defaults) without depending on
``transformers``; reads/writes the same ``config.json`` layout a HF checkpoint holds.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict


class DreamConfig:
    model_type = "Dream"

    def __init__(self, vocab_size=151936, hidden_size=768, intermediate_size=3072, num_hidden_layers=10,
                 num_attention_heads=8, num_key_value_heads=8, hidden_act="silu", max_position_embeddings=64,
                 initializer_range=0.02, rms_norm_eps=1e-6, use_cache=False, tie_word_embeddings=False,
                 rope_theta=1000.0, rope_scaling=None, use_sliding_window=False, sliding_window=4096,
                 max_window_layers=28, attention_dropout=0.0, mask_token_id=151666, pad_token_id=151643,
                 **kwargs):
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.use_sliding_window = use_sliding_window
        self.sliding_window = sliding_window if use_sliding_window else None
        self.max_window_layers = max_window_layers
        if num_key_value_heads is None:
            num_key_value_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.hidden_act = hidden_act
        self.initializer_range = initializer_range
        self.rms_norm_eps = rms_norm_eps
        self.use_cache = use_cache
        self.rope_theta = rope_theta
        self.rope_scaling = rope_scaling
        self.attention_dropout = attention_dropout
        self.tie_word_embeddings = tie_word_embeddings
        self.mask_token_id = mask_token_id
        self.pad_token_id = pad_token_id
        self.bos_token_id = kwargs.pop("bos_token_id", None)
        self.eos_token_id = kwargs.pop("eos_token_id", None)
        for k, v in kwargs.items():  # HF configs keep unknown keys as attributes
            setattr(self, k, v)
        self._check_supported()

    def _check_supported(self) -> None:
        """What the HIP path implements; everything else fails loudly instead of silently differing."""
        if self.num_key_value_heads != self.num_attention_heads:
            raise NotImplementedError("grouped-query attention (num_key_value_heads != num_attention_heads)")
        if self.hidden_act != "silu":
            raise NotImplementedError(f"hidden_act={self.hidden_act!r}; only 'silu' (SwiGLU)")
        if self.rope_scaling is not None:
            raise NotImplementedError("rope_scaling; only the default RoPE")
        if self.use_sliding_window:
            raise NotImplementedError("sliding-window attention")
        if self.tie_word_embeddings:
            raise NotImplementedError("tied lm_head (the reference's head is untied)")
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be divisible by num_attention_heads")

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def to_dict(self) -> Dict[str, Any]:
        d = {k: copy.deepcopy(v) for k, v in self.__dict__.items() if not k.startswith("_")}
        d["model_type"] = self.model_type
        return d

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "DreamConfig":
        d = {k: v for k, v in d.items() if k not in ("model_type", "architectures", "transformers_version",
                                                     "torch_dtype", "dtype", "auto_map")}
        return cls(**d)

    @classmethod
    def from_pretrained(cls, path: str) -> "DreamConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_dict(json.load(f))

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        d = self.to_dict()
        d["architectures"] = ["DreamModel_length"]
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(d, f, indent=2, sort_keys=True)

    def __repr__(self) -> str:
        return f"DreamConfig {json.dumps(self.to_dict(), indent=2, sort_keys=True, default=str)}"
