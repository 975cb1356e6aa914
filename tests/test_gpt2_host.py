"""CPU tests of the GPT-2 backbone's host side (dark_amd/custom_gpt.py): parameter names are transformers' GPT-2
state_dict names, the head is tied, checkpoints round-trip, and the product path refuses to run without a HIP device."""
import pytest
import torch

from dark_amd import synth
from dark_amd.custom_gpt import DiffusionModel, GPT2DiffusionConfig, create_gpt2_transformer

SHAPE = dict(vocab_size=512, n_positions=64, n_embd=128, n_layer=2, n_head=2)


def test_state_dict_names_and_tied_head():
    cfg = GPT2DiffusionConfig(**SHAPE)
    m = DiffusionModel(cfg)
    assert cfg.attn_causal and m.attn_causal   # the default is what the reference computes
    names = {n for n, _ in synth.gpt2_state_dict_layout(cfg)} | {"lm_head.weight"}
    assert set(m.state_dict().keys()) == names
    for n, shape in synth.gpt2_state_dict_layout(cfg):
        assert tuple(m.state_dict()[n].shape) == shape, n
    assert m.lm_head.weight is m.transformer.wte.weight
    w = synth.make_gpt2_weights(cfg, seed=1, rich=True)
    m.load_state_dict(w, strict=True)
    assert torch.equal(m.get_output_embeddings().weight, w["transformer.wte.weight"])
    # a checkpoint written by an older transformers carries causal-mask buffers: ignored
    w2 = dict(w)
    w2["transformer.h.0.attn.bias"] = torch.ones(1, 1, 64, 64, dtype=torch.bool)
    m.load_state_dict(w2, strict=True)


def test_save_and_from_pretrained_roundtrip(tmp_path):
    cfg = GPT2DiffusionConfig(**SHAPE, mask_token_id=synth.MASK, attn_causal=True)
    m = DiffusionModel(cfg)
    m.save_pretrained(str(tmp_path))
    m2 = DiffusionModel.from_pretrained(str(tmp_path))
    assert m2.attn_causal and m2.config.mask_token_id == synth.MASK
    for (k, a), (_, b) in zip(sorted(m.state_dict().items()), sorted(m2.state_dict().items())):
        assert torch.equal(a, b), k
    assert m2.lm_head.weight is m2.transformer.wte.weight


def test_create_gpt2_transformer_and_cpu_refusal():
    class Tok:
        pad_token_id, bos_token_id, eos_token_id, unk_token_id, sep_token_id = 2, 0, 2, 3, 4

    m = create_gpt2_transformer(299, Tok(), "gpt2-diffusion", n_layer=1, n_positions=64)
    assert m.config.vocab_size == 300 and m.config.n_embd == 768 and m.config.n_head == 12 and m.model_name == "gpt2-diffusion"
    assert m.config.sep_token_id == 4 and m.config.use_cache is False
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        DiffusionModel(GPT2DiffusionConfig(**SHAPE, activation_function="relu"))
