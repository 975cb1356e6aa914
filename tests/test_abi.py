"""CPU-side checks of the drop-in boundary: libdark_hip.so loads without a GPU, exports every symbol
include/dark_hip.h declares, the ctypes binding covers exactly that set, and argument validation fails
with a message instead of launching (no compute call is made here)."""
import ctypes as C
import os
import re

import pytest

from dark_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dark_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(dark_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    lib = _hip.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dark_hip.h but not exported"
    assert names == set(_hip.SIGNATURES), names ^ set(_hip.SIGNATURES)
    assert lib.dark_hip_abi_version() == _hip.ABI_VERSION


def test_argument_validation_without_gpu():
    lib = _hip.load()
    rc = lib.dark_sample_rows(None, 10, 1, 10, 0.0, 0.0, 0, 0, None, 0, None, None, None)
    assert rc == 1 and b"null pointer" in lib.dark_hip_last_error()
    with pytest.raises(_hip.DarkHipError):
        _hip.check(lib.dark_commit_topk(None, 4, 46, 6, None, 3, None, None, 1, None))
    assert lib.dark_forward_workspace_bytes(None, 4, 46) == 0
    desc = _hip.ModelDesc(dtype=1, vocab=512, vocab_pad=512, d=192, ff=384, n_layers=0, n_heads=2, head_dim=96,
                          max_pos=64, rms_eps=1e-6)
    assert lib.dark_forward_workspace_bytes(C.byref(desc), 4, 46) >= 4 * 46 * (192 * 5 + 384) * 2


def test_product_path_refuses_cpu():
    """No CPU fallback: the model raises on a CPU device instead of computing anything."""
    import torch
    from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, synth
    cfg = DreamConfig(vocab_size=512, hidden_size=192, intermediate_size=384, num_hidden_layers=1,
                      num_attention_heads=2, num_key_value_heads=2, mask_token_id=synth.MASK, pad_token_id=synth.END)
    m = DreamModel_length(cfg)
    ids = synth.make_inputs(2, "abd", nentity=300, nrelation=111)
    with pytest.raises(RuntimeError, match="HIP device"):
        m.diffusion_generate(ids, generation_config=DreamGenerationConfig(mask_token_id=synth.MASK, max_length=46),
                             steps=2, alg="entropy")
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.nn.functional.pad(ids, (0, 1), value=synth.MASK))
