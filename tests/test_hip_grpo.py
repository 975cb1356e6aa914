"""GPU parity tests of the GRPO forward-only log-prob path: the fused LM-head log-softmax-gather
and the trainer-level `get_per_token_logps`, against the reference goldens (tests/golden/make_goldens_grpo.py)."""
import json
import os

import numpy as np
import pytest
import torch

from dark_amd import DreamConfig, DreamModel_length, _hip, grpo_logps, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "reference_goldens_grpo.json")) as f:
        meta = json.load(f)
    return np.load(os.path.join(HERE, "golden", "reference_goldens_grpo.npz")), meta


def _model(cfg_kw, dtype, rich=True):
    cfg = DreamConfig(**cfg_kw)
    m = DreamModel_length(cfg)
    m.load_state_dict(synth.make_weights(cfg, seed=0, rich=rich), strict=True)
    return m.to(DEV).to(dtype)


def _cpu_rand(i, rows, cols):
    return torch.rand((rows, cols))   # the CPU generator `set_seed(seed)` has just seeded, as in the goldens


def test_forward_process_bit_exact(gold):
    g, meta = gold
    for c in meta["cases"]:
        ids = torch.from_numpy(g[c["name"] + "_ids"]).to(DEV)
        tgt = torch.from_numpy(g[c["name"] + "_target"]).to(DEV)
        for i, seed in enumerate(c["seeds"]):
            noisy, wts, cm = grpo_logps.forward_process(ids[i], tgt[i], synth.MASK, seed=seed,
                                                        accumulate=c["iterations"] > 1,
                                                        gradient_accumulation_steps=c["grad_accum"],
                                                        _rand_fn=lambda r, c_: torch.rand((r, c_)))
            assert np.array_equal(torch.stack(noisy).cpu().numpy(), g[f"{c['name']}_noisy{i}"]), (c["name"], i)
            assert np.array_equal(cm.cpu().numpy(), g[f"{c['name']}_cmask{i}"])
            np.testing.assert_allclose(np.array(wts, dtype=np.float64), g[f"{c['name']}_weights{i}"], rtol=0, atol=0)


def test_selective_log_softmax_device(gold):
    g, _ = gold
    logits, index = torch.from_numpy(g["sls_logits"]).to(DEV), torch.from_numpy(g["sls_index"]).to(DEV)
    weights, mask = torch.from_numpy(g["sls_weights"]).to(DEV), torch.from_numpy(g["sls_mask"]).to(DEV)
    got = grpo_logps.selective_log_softmax(logits, index, weights, mask).cpu().numpy()
    np.testing.assert_allclose(got, g["sls_out"], rtol=1e-5, atol=1e-5)
    got = grpo_logps.selective_log_softmax(logits, index, weights, None).cpu().numpy()
    np.testing.assert_allclose(got, g["sls_out_plain"], rtol=1e-5, atol=1e-5)


def test_per_token_logps_fp32_matches_reference(gold):
    """fp32 model: within the north star's 1e-3 relative of the reference's log-probs (measured ~1e-5)."""
    g, meta = gold
    m = _model(meta["mini"], torch.float32)
    for c in meta["cases"]:
        ids, tgt = torch.from_numpy(g[c["name"] + "_ids"]), torch.from_numpy(g[c["name"] + "_target"])
        got = grpo_logps.get_per_token_logps(m, ids.to(DEV), tgt.to(DEV), c["seeds"], synth.MASK,
                                             gradient_accumulation_steps=c["grad_accum"], _rand_fn=_cpu_rand)
        want = g[c["name"] + "_logps"]
        assert got.shape == want.shape and got.dtype == torch.float32
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-3, atol=1e-4)
        assert np.abs(got.cpu().numpy() - want).max() < 2e-4


def test_per_token_logps_bf16_fused(gold):
    """bf16 model (fused log-sum-exp epilogue, logits never written).  Stated bf16 bound: |delta logp| < 0.12
    everywhere and < 0.03 in the median against the fp32 reference (log-probs are ~ -10)."""
    g, meta = gold
    m = _model(meta["mini"], torch.bfloat16)
    for c in meta["cases"]:
        ids, tgt = torch.from_numpy(g[c["name"] + "_ids"]), torch.from_numpy(g[c["name"] + "_target"])
        got = grpo_logps.get_per_token_logps(m, ids.to(DEV), tgt.to(DEV), c["seeds"], synth.MASK,
                                             gradient_accumulation_steps=c["grad_accum"], _rand_fn=_cpu_rand)
        err = np.abs(got.cpu().numpy() - g[c["name"] + "_logps"])
        assert err.max() < 0.12 and np.median(err) < 0.03, (c["name"], err.max(), np.median(err))


@pytest.mark.parametrize("vocab", [512, 15080, 25427, 41066])
def test_fused_logprob_equals_unfused(vocab):
    """dark_lm_head_logprob (bf16, fused) against the same head materialised with dark_lm_head_rows + a torch fp32
    log_softmax/gather of those logits, on random hidden states; also edge targets (0, V-1) and gathered rows."""
    lib = _hip.load()
    cfg = dict(vocab_size=vocab, hidden_size=192, intermediate_size=384, num_hidden_layers=1, num_attention_heads=2,
               num_key_value_heads=2, max_position_embeddings=64, mask_token_id=synth.MASK, pad_token_id=synth.END)
    m = _model(cfg, torch.bfloat16, rich=False)
    pk = m.packed()
    g = torch.Generator().manual_seed(0)
    R, M = 300, 517
    hidden = (torch.randn(R, 192, generator=g) * 3).to(DEV).to(pk.resid_dtype)   # (the model's residual-stream dtype)
    src = torch.randint(0, R, (M,), generator=g).int().to(DEV)
    targets = torch.randint(0, vocab, (M,), generator=g)
    targets[:4] = torch.tensor([0, vocab - 1, 127, 128])
    targets = targets.to(DEV)
    s = _hip.stream_ptr(torch.device(DEV))
    ws_b = lib.dark_logprob_workspace_bytes(pk.ref, M)
    ws = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
    lp = torch.empty(M, dtype=torch.float32, device=DEV)
    _hip.check(lib.dark_lm_head_logprob(pk.ref, _hip.ptr(hidden), _hip.ptr(src), R, _hip.ptr(targets), M, _hip.ptr(ws),
                                        ws_b, _hip.ptr(lp), s))
    hws_b = lib.dark_head_workspace_bytes(pk.ref, M)
    hws = torch.empty(hws_b, dtype=torch.uint8, device=DEV)
    logits = torch.empty((M, vocab), dtype=torch.float32, device=DEV)
    _hip.check(lib.dark_lm_head_rows(pk.ref, _hip.ptr(hidden), _hip.ptr(src), M, _hip.ptr(hws), hws_b,
                                     _hip.ptr(logits), vocab, s))
    want = torch.log_softmax(logits.double(), -1).gather(-1, targets[:, None])[:, 0].float()
    np.testing.assert_allclose(lp.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-5)
    lp2 = torch.empty(M, dtype=torch.float32, device=DEV)
    _hip.check(lib.dark_logprob_from_logits(_hip.ptr(logits), vocab, _hip.ptr(targets), M, vocab, _hip.ptr(lp2), s))
    np.testing.assert_allclose(lp2.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-5)


def test_errors_and_mixin(gold):
    g, meta = gold
    m = _model(meta["mini"], torch.float32)
    with pytest.raises(ValueError, match="3 dimensions"):
        grpo_logps.get_per_token_logps(m, torch.zeros(2, 46, dtype=torch.long, device=DEV),
                                       torch.zeros(2, 46, device=DEV), [1], synth.MASK)
    with pytest.raises(ValueError, match="mask_seeds length"):
        grpo_logps.get_per_token_logps(m, torch.zeros(1, 2, 46, dtype=torch.long, device=DEV),
                                       torch.zeros(1, 2, 46, device=DEV), [1, 2], synth.MASK)

    class Trainer(grpo_logps.DiffuGRPOLogpsMixin):
        class args:
            gradient_accumulation_steps = 1

        class processing_class:
            mask_token_id = synth.MASK

    c = meta["cases"][0]
    ids, tgt = torch.from_numpy(g[c["name"] + "_ids"]).to(DEV), torch.from_numpy(g[c["name"] + "_target"]).to(DEV)
    out = Trainer()._get_per_token_logps(m, ids, tgt, c["seeds"])       # device RNG: shape / finiteness only
    assert out.shape == (c["batch"], c["iterations"], 46) and bool(torch.isfinite(out).all())
    logits = Trainer().get_logits(m, ids[0])
    assert logits.shape == (c["batch"], 46, meta["mini"]["vocab_size"])
