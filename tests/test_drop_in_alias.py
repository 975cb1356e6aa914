"""The drop-in import path: ``dark_amd.install_akgr_alias()`` registers the package under the reference's module path, and
a caller written against the reference -- ``main.test_batch`` (akgr/abduction_model/main.py:347-421) and the GRPO
rollout (akgr/abduction_model/custom_grpo2.py:605-632), restated here line for line from the point where the tokenizer
has produced ids -- runs unchanged on it: same imports, same constructor keywords, same call keywords
(``number_transfer_tokens=1`` included), same use of the result.  The decode step is checked at the id level through
``dark_split_strip`` (the strings are the tokenizer's)."""
import importlib
import os
import sys

import pytest
import torch

import dark_amd
from dark_amd import layout, synth
from oracle import dream_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
SPECIAL = (0, 2, 3, 4, 5, 6)   # START END UNK SEP EXPAND MASK (configs/config-dataloader.yml:1-15)


class _Tok:
    """The attributes of the reference's tokenizer object that test_batch reads (main.py:349-354)."""
    mask_token_id, pad_token_id, bos_token_id, eos_token_id = synth.MASK, synth.END, synth.BOS, synth.END
    expand_token_id, sep_token_id = synth.EXPAND, synth.SEP


@pytest.fixture(scope="module")
def akgr_modules():
    for k in [k for k in sys.modules if k == "akgr" or k.startswith("akgr.")]:
        del sys.modules[k]
    dark_amd.install_akgr_alias()
    gu = importlib.import_module("akgr.abduction_model.model.custom_mydream.generation_utils")
    md = importlib.import_module("akgr.abduction_model.model.custom_mydream.modeling_dream_length")
    cf = importlib.import_module("akgr.abduction_model.model.custom_mydream.configuration_dream")
    yield gu, md, cf
    for k in [k for k in sys.modules if k == "akgr" or k.startswith("akgr.")]:
        del sys.modules[k]


def _build(md, cf, layers=2):
    kw = dict(vocab_size=15080, hidden_size=768, intermediate_size=3072, num_hidden_layers=layers, num_attention_heads=8,
              num_key_value_heads=8, max_position_embeddings=64, mask_token_id=synth.MASK, pad_token_id=synth.END)
    cfg = cf.DreamConfig(**kw)
    w = synth.make_weights(cfg, seed=0, rich=False)
    model = md.DreamModel_length(cfg)
    model.load_state_dict(w, strict=True)
    return kw, w, model


def test_alias_modules_are_this_package(akgr_modules):
    gu, md, cf = akgr_modules
    assert md.DreamModel_length is dark_amd.DreamModel_length
    assert gu.DreamGenerationConfig is dark_amd.DreamGenerationConfig and gu.DreamModelOutput is dark_amd.DreamModelOutput
    assert cf.DreamConfig is dark_amd.DreamConfig
    # the imports main.py itself makes (main.py:46-58: `from akgr.abduction_model.model.custom_mydream.... import ...`)
    from akgr.abduction_model.model.custom_mydream.generation_utils import DreamGenerationConfig  # noqa: F401
    from akgr.abduction_model.model.custom_mydream.modeling_dream_length import DreamModel_length  # noqa: F401
    from akgr.abduction_model.model.custom_mydream.modeling_dream import DreamModel  # noqa: F401
    from akgr.abduction_model.model.custom_gpt import DiffusionModel  # noqa: F401


def test_restated_test_batch_call(akgr_modules):
    gu, md, cf = akgr_modules
    DreamGenerationConfig = gu.DreamGenerationConfig
    kw, w, model = _build(md, cf)
    tokenizer, device = _Tok(), torch.device(DEV)
    # ---- main.py:349-363
    mask_token_id = tokenizer.mask_token_id
    bos_token_id = tokenizer.bos_token_id
    eos_token_id = tokenizer.eos_token_id
    expand_token_id = tokenizer.expand_token_id
    sep_token_id = tokenizer.sep_token_id
    generation_config = DreamGenerationConfig(mask_token_id=mask_token_id,
                                              pad_token_id=eos_token_id,
                                              bos_token_id=bos_token_id,
                                              eos_token_id=eos_token_id,
                                              expand_token_id=expand_token_id,
                                              delete_token_id=eos_token_id,
                                              sep_token_id=sep_token_id,
                                              max_length=46,
                                              )
    steps = 16   # (main.py:401 runs 64: the same loop, four times as long)
    for test_type, seed in (("abd", 5), ("ded", 6)):
        input_ids = synth.make_inputs(8, test_type, seed=seed)   # stands in for tokenizer.py:287-338
        host_ids = input_ids.clone()
        # ---- main.py:387-404
        input_ids = input_ids.to(device)
        model.to(device)
        g = torch.Generator().manual_seed(7)
        noises = {}

        def noise_fn(step, M, V):
            if step not in noises:
                noises[step] = torch.empty(M, V).exponential_(1, generator=g)
            return noises[step]

        output = model.diffusion_generate(
            inputs=input_ids,
            generation_config=generation_config,
            temperature=0.1,
            alg='entropy',
            alg_temp=0,
            top_p=0.9,
            return_dict_in_generate=True,
            steps=steps,
            output_history=True,
            number_transfer_tokens=1,
            _noise_fn=noise_fn,   # (test hook: the categorical noise the oracle is fed below)
        )
        # ---- main.py:406-421
        test_ids = output.sequences
        history = output.history
        assert isinstance(output, gu.DreamModelOutput) and len(history) == steps and test_ids.shape == (8, 46)
        assert torch.equal(input_ids.cpu(), host_ids)   # the caller's tensor is not mutated
        sep_positions = (test_ids[0] == sep_token_id).nonzero(as_tuple=True)[0]
        sep_pos = sep_positions[0].item()
        left_parts = test_ids[:, :sep_pos]
        right_parts = test_ids[:, sep_pos + 1:]
        strip = lambda part: [t for t in part.tolist() if t not in SPECIAL]   # decode(skip_special_tokens=True), ids
        want_left = [strip(p) for p in left_parts.cpu()]
        want_right = [strip(p) for p in right_parts.cpu()]
        got_sep, left, ll, right, rl = layout.split_strip(test_ids, sep_token_id, SPECIAL)
        assert got_sep == sep_pos == 11
        assert layout.to_lists(left, ll) == want_left and layout.to_lists(right, rl) == want_right
        # ---- and the sequences are the reference's: the oracle on the same ids, settings and noise
        ocfg = O.OracleConfig(**{k: v for k, v in kw.items() if k in O.OracleConfig.__dataclass_fields__})
        seq, hist = O.sample_loop(w, ocfg, host_ids, max_length=46, steps=steps, mask_token_id=mask_token_id,
                                  alg="entropy", alg_temp=0, temperature=0.1, top_p=0.9,
                                  noise_fn=lambda s, M, V: noises[s], tie_break="index")
        assert torch.equal(torch.stack(history).cpu(), torch.stack(hist))
        assert torch.equal(test_ids.cpu(), seq)


def test_restated_grpo_rollout_call(akgr_modules):
    """custom_grpo2.py:605-632: no sep_token_id in the config, T = 0.5, 32 steps, chunks of generation_batch_size."""
    gu, md, cf = akgr_modules
    _, _, model = _build(md, cf, layers=1)
    model.to(DEV)
    tok = _Tok()
    generation_config = gu.DreamGenerationConfig(mask_token_id=tok.mask_token_id,
                                                 pad_token_id=tok.eos_token_id,
                                                 bos_token_id=tok.bos_token_id,
                                                 eos_token_id=tok.eos_token_id,
                                                 expand_token_id=tok.expand_token_id,
                                                 delete_token_id=tok.eos_token_id,
                                                 max_length=46
                                                 )
    prompt_ids = synth.make_inputs(6, "abd", seed=9).repeat_interleave(4, 0).to(DEV)   # num_generations = 4
    generation_batch_size = 16
    torch.manual_seed(3)
    prompt_completion_ids_all = []
    for i in range(0, prompt_ids.size(0), generation_batch_size):
        end_idx = min(i + generation_batch_size, prompt_ids.size(0))
        batch_prompt_id = prompt_ids[i:end_idx]
        batch_prompt_completion_ids = model.diffusion_generate(
            inputs=batch_prompt_id,
            generation_config=generation_config,
            temperature=0.5,
            alg='entropy',
            alg_temp=0,
            top_p=0.9,
            return_dict_in_generate=True,
            output_history=True,
            number_transfer_tokens=1,
            steps=32
        )
        prompt_completion_ids_all.append(batch_prompt_completion_ids.sequences)
    out = torch.cat(prompt_completion_ids_all, 0)
    assert out.shape == (24, 46) and out.dtype == torch.int64 and out.device.type == "cuda"
    assert not bool((out == tok.mask_token_id).any())
    # prompt positions are untouched; the four trajectories of a query differ (stochastic rollouts)
    keep = prompt_ids != tok.mask_token_id
    assert torch.equal(out[:, :45][keep], prompt_ids[keep])
    groups = out.view(6, 4, 46)
    assert any(not torch.equal(groups[q, 0], groups[q, j]) for q in range(6) for j in range(1, 4))


@pytest.mark.gpu
def test_integration_md_ctypes_stub_runs_and_matches_the_oracle():
    """The ctypes stub INTEGRATION.md section 2 shows a maintainer (`sample_tokens` over `dark_sample_rows`) is executed
    as printed -- only the library path is made absolute -- and checked against the oracle's `sample_tokens`
    (generation_utils.py:58-90) for `top_p=None` and `top_p=0.9` at T = 0 (no draw involved) and T > 0 (finite tokens)."""
    import re
    from dark_amd import _hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes as C, torch\n.*?)```", md, re.S).group(1)
    assert 'C.CDLL("libdark_hip.so")' in block
    block = block.replace('C.CDLL("libdark_hip.so")', f"C.CDLL({_hip.LIB_PATH!r})")
    block = block.replace("lib.dark_hip_last_error().decode()", "str(rc)")   # (restype of the message call is not set in the stub)
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(64, 512, generator=g) * 3).cuda()
    for top_p in (None, 0.9):
        conf, x0 = ns["sample_tokens"](logits, temperature=0.0, top_p=top_p, neg_entropy=True)
        oc, ox = O.sample_tokens(logits.cpu(), temperature=0.0, top_p=top_p, neg_entropy=True)
        torch.cuda.synchronize()
        assert torch.equal(x0.cpu(), ox), top_p
        assert torch.allclose(conf.cpu(), oc, rtol=1e-4, atol=1e-6), top_p
        conf, x0 = ns["sample_tokens"](logits, temperature=0.7, top_p=top_p)
        torch.cuda.synchronize()
        assert int(x0.min()) >= 0 and int(x0.max()) < 512 and bool(torch.isfinite(conf).all())
