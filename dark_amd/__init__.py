"""dark_amd: MI355X-native denoise / unmask hot path of DARK behind the reference's Python surface.

    from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length

`dark_amd.install_akgr_alias()` additionally registers the package under the reference's module path
(`akgr.abduction_model.model.custom_mydream.*`) so callers written against the reference import it unchanged.
"""
import sys
import types

from .configuration_dream import DreamConfig
from .generation_utils import DreamGenerationConfig, DreamGenerationMixin, DreamModelOutput
from .modeling_dream_length import DreamModel_length
from .modeling_dream import DreamModel

__all__ = ["DreamConfig", "DreamGenerationConfig", "DreamGenerationMixin", "DreamModelOutput",
           "DreamModel_length", "DreamModel", "install_akgr_alias"]


def install_akgr_alias() -> None:
    """Expose this package as akgr.abduction_model.model.custom_mydream."""
    from . import (configuration_dream, generation_expand_utils, generation_utils, modeling_dream,
                   modeling_dream_length)
    chain = ["akgr", "akgr.abduction_model", "akgr.abduction_model.model",
             "akgr.abduction_model.model.custom_mydream"]
    for name in chain:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    base = chain[-1]
    for sub, mod in (("configuration_dream", configuration_dream), ("generation_utils", generation_utils),
                     ("modeling_dream_length", modeling_dream_length),
                     ("generation_expand_utils", generation_expand_utils), ("modeling_dream", modeling_dream)):
        sys.modules[f"{base}.{sub}"] = mod
        setattr(sys.modules[base], sub, mod)
    from . import custom_gpt   # the legacy GPT-2 backbone lives one level up: akgr.abduction_model.model.custom_gpt
    sys.modules[f"{chain[2]}.custom_gpt"] = custom_gpt
    setattr(sys.modules[chain[2]], "custom_gpt", custom_gpt)
