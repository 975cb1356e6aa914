"""Pins oracle/grpo_oracle.py (GRPO forward-only log-probs) against fixtures produced by executing
the reference's own definitions (tests/golden/make_goldens_grpo.py)."""
import json
import os

import numpy as np
import pytest
import torch

from dark_amd import synth
from oracle import dream_oracle as O
from oracle import grpo_oracle as G

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "reference_goldens_grpo.json")) as f:
        meta = json.load(f)
    return np.load(os.path.join(HERE, "golden", "reference_goldens_grpo.npz")), meta


@pytest.fixture(scope="module")
def mini(gold):
    g, meta = gold
    cfg = O.OracleConfig(**{k: v for k, v in meta["mini"].items() if k in O.OracleConfig.__dataclass_fields__})
    w = synth.make_weights(cfg, seed=0, rich=True)
    chk = np.array([float(w[k].double().sum()) for k in sorted(w)])
    np.testing.assert_allclose(chk, g["w_checksum"], rtol=0, atol=0)
    return cfg, w


def test_selective_log_softmax(gold):
    g, _ = gold
    logits, index = torch.from_numpy(g["sls_logits"]), torch.from_numpy(g["sls_index"])
    weights, mask = torch.from_numpy(g["sls_weights"]), torch.from_numpy(g["sls_mask"])
    np.testing.assert_allclose(G.selective_log_softmax(logits, index, weights, mask).numpy(), g["sls_out"],
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(G.selective_log_softmax(logits, index, weights, None).numpy(), g["sls_out_plain"],
                               rtol=1e-6, atol=1e-6)


def test_forward_process_is_bit_exact(gold):
    g, meta = gold
    for c in meta["cases"]:
        ids, tgt = torch.from_numpy(g[c["name"] + "_ids"]), torch.from_numpy(g[c["name"] + "_target"])
        for i, seed in enumerate(c["seeds"]):
            noisy, wts, cm = G.forward_process(ids[i], tgt[i], synth.MASK, seed=seed, accumulate=c["iterations"] > 1,
                                               gradient_accumulation_steps=c["grad_accum"])
            assert np.array_equal(torch.stack(noisy).numpy(), g[f"{c['name']}_noisy{i}"]), (c["name"], i)
            assert np.array_equal(cm.numpy(), g[f"{c['name']}_cmask{i}"])
            np.testing.assert_allclose(np.array(wts, dtype=np.float64), g[f"{c['name']}_weights{i}"], rtol=0, atol=0)


def test_per_token_logps(gold, mini):
    g, meta = gold
    cfg, w = mini
    for c in meta["cases"]:
        ids, tgt = torch.from_numpy(g[c["name"] + "_ids"]), torch.from_numpy(g[c["name"] + "_target"])
        got = G.per_token_logps(w, cfg, ids, tgt, c["seeds"], synth.MASK, gradient_accumulation_steps=c["grad_accum"])
        want = g[c["name"] + "_logps"]
        assert got.shape == want.shape and got.dtype == torch.float32
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-5, atol=2e-5)


def test_shape_errors(mini):
    cfg, w = mini
    with pytest.raises(ValueError, match="3 dimensions"):
        G.per_token_logps(w, cfg, torch.zeros(2, 46, dtype=torch.long), torch.zeros(2, 46), [1], synth.MASK)
    with pytest.raises(ValueError, match="mask_seeds length"):
        G.per_token_logps(w, cfg, torch.zeros(1, 2, 46, dtype=torch.long), torch.zeros(1, 2, 46), [1, 2], synth.MASK)
