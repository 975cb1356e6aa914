"""GPU parity tests at the edges of the fixed-length loop (generation_utils.py:435-548): batch of one, a single
denoise step, inputs that hold no MASK of their own, rows with very different mask counts, the longest supported
sequence (N = 128), top_k = 1 and top_p = 0, and the argument errors -- each against the oracle on injected noise."""
import numpy as np
import pytest
import torch

from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, _hip, synth
from oracle import dream_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(vocab_size=1200, hidden_size=192, intermediate_size=384, num_hidden_layers=2, num_attention_heads=2,
          num_key_value_heads=2, max_position_embeddings=128, mask_token_id=synth.MASK, pad_token_id=synth.END)


@pytest.fixture(scope="module")
def mw():
    cfg = DreamConfig(**KW)
    w = synth.make_weights(cfg, seed=3, rich=True)
    m = DreamModel_length(cfg)
    m.load_state_dict(w, strict=True)
    return m.to(DEV), w, O.OracleConfig(**{k: v for k, v in KW.items() if k in O.OracleConfig.__dataclass_fields__})


def _gc(max_length):
    return DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                                 eos_token_id=synth.END, sep_token_id=synth.SEP, max_length=max_length)


def _both(mw, ids, max_length, seed=0, **kw):
    """Histories of this build and of the oracle (device tie order) on the same categorical draws."""
    m, w, ocfg = mw
    g = torch.Generator().manual_seed(seed)
    noises = {}

    def noise_fn(step, M, V):
        if step not in noises:
            noises[step] = torch.empty(M, V).exponential_(1, generator=g)
        return noises[step]

    out = m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(max_length), return_dict_in_generate=True,
                               output_history=True, _noise_fn=noise_fn, **kw)
    seq, hist = O.sample_loop(w, ocfg, ids, max_length=max_length, mask_token_id=synth.MASK,
                              noise_fn=lambda s, M, V: noise_fn(s, M, V), tie_break="index", **kw)
    return out, seq, hist


def _assert_same(out, seq, hist, what):
    assert len(out.history) == len(hist), what
    for i, (a, b) in enumerate(zip(out.history, hist)):
        assert torch.equal(a.cpu(), b), f"{what}: history differs at step {i}"
    assert torch.equal(out.sequences.cpu(), seq), what


def _ids(B, n_mask, n_in=45, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(100, KW["vocab_size"] - 1, (B, n_in), generator=g)
    x[:, 0] = synth.BOS
    x[:, 1:1 + n_mask] = synth.MASK
    return x


def test_batch_of_one_and_odd_batches(mw):
    for B in (1, 3, 7):
        out, seq, hist = _both(mw, _ids(B, 9, seed=B), 46, seed=B, steps=8, temperature=0.5, top_p=0.9, alg="entropy",
                               alg_temp=0)
        _assert_same(out, seq, hist, f"B={B}")
        assert not bool((out.sequences == synth.MASK).any())


def test_single_step_commits_everything(mw):
    """steps = 1: the only step is the last one, k = the batch-average mask count (generation_utils.py:520-521)."""
    out, seq, hist = _both(mw, _ids(4, 12, seed=5), 46, seed=5, steps=1, temperature=0.3, top_p=0.9, alg="maskgit_plus",
                           alg_temp=0)
    _assert_same(out, seq, hist, "steps=1")
    assert not bool((out.sequences == synth.MASK).any())


def test_input_without_masks_only_the_pad_position_is_generated(mw):
    """An input that holds no MASK: the loop still pads one MASK to max_length (:460) and generates that position."""
    ids = _ids(5, 0, seed=6)
    out, seq, hist = _both(mw, ids, 46, seed=6, steps=4, temperature=0.0, alg="entropy", alg_temp=0)
    _assert_same(out, seq, hist, "no masks")
    assert torch.equal(out.sequences[:, :45].cpu(), ids)
    assert not bool((out.sequences[:, 45] == synth.MASK).any())


def test_many_pad_positions(mw):
    """max_length far beyond the input: every padded position starts as MASK (:460)."""
    ids = _ids(3, 2, n_in=20, seed=7)
    out, seq, hist = _both(mw, ids, 64, seed=7, steps=10, temperature=0.4, top_p=0.95, alg="topk_margin", alg_temp=0)
    _assert_same(out, seq, hist, "max_length 64, input 20")


def test_ragged_mask_counts_incl_remask_quirk(mw):
    """Rows with 1 .. 30 masks under the batch-average k: rows that run out of masks early get MASK written back
    over the positions topk picks among their -inf confidences (:531-534); the oracle reproduces it, so must this."""
    ids = _ids(6, 0, seed=8)
    for b, n in enumerate((0, 1, 4, 12, 20, 30)):
        ids[b, 1:1 + n] = synth.MASK
    out, seq, hist = _both(mw, ids, 46, seed=8, steps=6, temperature=0.5, top_p=0.9, alg="entropy", alg_temp=0)
    _assert_same(out, seq, hist, "ragged")


def test_longest_sequence(mw):
    """N = 128, the longest sequence the attention kernel serves (8 tiles), 64 masks a row (BASELINE config 4's layout).
    (maskgit_plus: with 64 masked positions the ENTROPIES of this flat random-init model sit within 1e-6 of each other
    around the k-th place -- tools/probe/dbg_n128.py prints them -- and an fp32 near-tie parts a free run; the
    probability of the sampled token does not cluster.)"""
    ids = _ids(4, 64, n_in=127, seed=9)
    out, seq, hist = _both(mw, ids, 128, seed=9, steps=8, temperature=0.5, top_p=0.9, alg="maskgit_plus", alg_temp=0)
    _assert_same(out, seq, hist, "N=128")
    m = mw[0]
    with pytest.raises(_hip.DarkHipError, match="N must be in"):
        m.diffusion_generate(inputs=_ids(1, 4, n_in=128).to(DEV), generation_config=_gc(129), steps=2)


def test_top_k_one_and_top_p_zero_are_greedy_with_unit_confidence(mw):
    """top_k = 1 and top_p = 0 keep the arg-max only (generation_utils.py:37-55): x0 is the greedy token whatever
    the noise, the kept distribution is a point mass."""
    ids = _ids(4, 10, seed=10)
    greedy, _, _ = _both(mw, ids, 46, seed=10, steps=5, temperature=0.0, alg="maskgit_plus", alg_temp=0)
    for extra in (dict(top_k=1), dict(top_p=0.0)):
        out, seq, hist = _both(mw, ids, 46, seed=10, steps=5, temperature=0.7, alg="maskgit_plus", alg_temp=0, **extra)
        _assert_same(out, seq, hist, str(extra))
    # with a point mass every confidence is 1.0: which positions transfer is the tie order, the tokens are the greedy ones
    tr = {}
    m = mw[0]
    m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(46), steps=1, temperature=0.7, top_k=1,
                         alg="maskgit_plus", alg_temp=0, _trace=tr)
    tg = {}
    m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(46), steps=1, temperature=0.0, alg="maskgit_plus",
                         alg_temp=0, _trace=tg)
    assert torch.equal(tr["x0"][0], tg["x0"][0])
    assert torch.allclose(tr["confidence"][0], torch.ones_like(tr["confidence"][0]))


def test_argument_errors(mw):
    m = mw[0]
    ids = _ids(2, 4).to(DEV)
    with pytest.raises(AssertionError):
        m.diffusion_generate(inputs=None, generation_config=_gc(46))
    with pytest.raises(ValueError):
        m.diffusion_generate(inputs=ids, generation_config=_gc(45))          # N_in >= max_length (:186-192)
    with pytest.raises(RuntimeError, match="Unknown alg"):
        m.diffusion_generate(inputs=ids, generation_config=_gc(46), alg="nope", steps=2)
    with pytest.raises(ValueError, match="CPU tensor|device memory"):
        _hip.ptr(torch.zeros(4))
