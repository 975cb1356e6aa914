"""A checkpoint directory written by the REFERENCE's own writer (HF ``save_pretrained`` of its ``DreamModel_length``,
what ``Trainer.save_model(<out>/final)`` produces, akgr/abduction_model/main.py:225-226) is read by
``dark_amd.DreamModel_length.from_pretrained`` (modeling_dream_length.py:549-600): single-file and sharded layouts,
``generation_config.json`` as HF's ``GenerationConfig`` writes it.  Fixtures: tests/golden/ref_ckpt_tiny*/ and
reference_goldens_ckpt.npz, written by tests/golden/make_goldens_ckpt.py."""
import glob
import json
import os
import shutil

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from dark_amd import DreamGenerationConfig, DreamModel_length
from conftest import GOLDEN_DIR

DIRS = ("ref_ckpt_tiny", "ref_ckpt_tiny_sharded")


@pytest.mark.parametrize("name", DIRS)
def test_reference_written_directory_loads(name):
    d = os.path.join(GOLDEN_DIR, name)
    with open(os.path.join(d, "config.json")) as f:
        raw = json.load(f)
    # the producer's keys a hand-written config would not have (what made this file format worth pinning)
    assert raw["architectures"] == ["DreamModel_length"] and "transformers_version" in raw
    m = DreamModel_length.from_pretrained(d)
    assert m.config.vocab_size == raw["vocab_size"] and m.config.hidden_size == raw["hidden_size"]
    assert m.config.rope_theta == raw["rope_theta"] and m.config.pad_token_id == raw["pad_token_id"]
    want = {}
    for f in glob.glob(os.path.join(d, "*.safetensors")):
        want.update(load_file(f))
    got = m.state_dict()
    assert set(got) == set(want)
    for k, v in want.items():
        assert torch.equal(got[k], v), k
    gc = m.generation_config
    assert isinstance(gc, DreamGenerationConfig)
    with open(os.path.join(d, "generation_config.json")) as f:
        graw = json.load(f)
    assert "transformers_version" in graw and "_from_model_config" in graw     # HF's extra keys are tolerated
    for k in ("steps", "temperature", "top_p", "alg", "alg_temp", "mask_token_id", "max_length", "sep_token_id",
              "expand_token_id", "delete_token_id", "eps"):
        assert getattr(gc, k) == graw[k], k


def test_sharded_index_is_honoured(tmp_path):
    src = os.path.join(GOLDEN_DIR, "ref_ckpt_tiny_sharded")
    d = str(tmp_path / "ckpt")
    shutil.copytree(src, d)
    with open(os.path.join(d, "model.safetensors.index.json")) as f:
        shards = sorted(set(json.load(f)["weight_map"].values()))
    assert len(shards) > 1
    # a stray safetensors file the index does not name must be ignored ...
    shutil.copy(os.path.join(d, shards[0]), os.path.join(d, "zz_stray.safetensors"))
    DreamModel_length.from_pretrained(d)
    # ... and a shard the index names must be there
    os.remove(os.path.join(d, shards[-1]))
    with pytest.raises(OSError):
        DreamModel_length.from_pretrained(d)


@pytest.mark.gpu
@pytest.mark.parametrize("name", DIRS)
def test_reference_checkpoint_reproduces_reference_logits(name):
    g = np.load(os.path.join(GOLDEN_DIR, "reference_goldens_ckpt.npz"))
    m = DreamModel_length.from_pretrained(os.path.join(GOLDEN_DIR, name)).to("cuda")
    ids = torch.from_numpy(g["ids"]).to("cuda")
    out = m(ids).logits
    assert out.dtype == torch.float32
    got, ref = out.cpu().numpy(), g["logits"]
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-5)   # north-star bound: 1e-3 relative
    # the loop runs from the settings that travelled with the checkpoint
    seq = m.diffusion_generate(inputs=ids[:, :-1], generation_config=m.generation_config)
    assert seq.shape == (ids.shape[0], m.generation_config.max_length) and not bool((seq == m.config.mask_token_id).any())
    # no config passed: the reference builds one from the MODEL config, not from the loaded file (generation_utils.py:
    # 230-232), so the length is min(20 + 45, max_position_embeddings) (:212-217) and the default alg / steps apply
    seq = m.diffusion_generate(inputs=ids[:, :-1], steps=8)
    assert seq.shape == (ids.shape[0], min(20 + 45, m.config.max_position_embeddings))


@pytest.mark.gpu
def test_bf16_model_returns_bf16_logits():
    """modeling_dream_length.py:809: the reference's lm_head output carries the model's dtype."""
    g = np.load(os.path.join(GOLDEN_DIR, "reference_goldens_ckpt.npz"))
    d = os.path.join(GOLDEN_DIR, "ref_ckpt_tiny")
    m = DreamModel_length.from_pretrained(d, torch_dtype=torch.bfloat16).to("cuda")
    ids = torch.from_numpy(g["ids"]).to("cuda")
    out = m(ids).logits
    assert out.dtype == torch.bfloat16
    ref = g["logits"]
    err = np.abs(out.float().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 0.03, err   # the stated bf16 bound
