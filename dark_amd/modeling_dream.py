"""``DreamModel``: the same denoiser with the variable-length (EXPAND / DELETE) sampler mixed in.

Mirrors ``akgr/abduction_model/model/custom_mydream/modeling_dream.py``, which is ``modeling_dream_length.py``
with the generation mixin taken from ``generation_expand_utils`` (:46-47) and the class named ``DreamModel``
(:736); weights, forward and checkpoint layout are shared.
"""
from __future__ import annotations

from .configuration_dream import DreamConfig  # noqa: F401
from .generation_expand_utils import DreamGenerationConfig, DreamGenerationMixin, DreamModelOutput  # noqa: F401
from .modeling_dream_length import DreamModel_length, MaskedLMOutput  # noqa: F401


class DreamModel(DreamGenerationMixin, DreamModel_length):
    """modeling_dream.py:736-824 (inference surface): `diffusion_generate` is the variable-length sampler,
    `diffusion_generate_length` the fixed-length one."""
    generation_config_class = DreamGenerationConfig
