"""GPU parity tests for the BASELINE.json configurations and boundary cases that round 1 left open:
config 3's vocabulary (V = 25427) through the whole loop, config 5's repeated rows under the lazy Philox noise,
real-width goldens for deduction batches and for config 1 itself, the one-call `dark_denoise_step`, int32 ids,
think-sample generator accounting and the k-schedule's linspace."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, _hip, synth
from dark_amd.generation_utils import philox_fill_plan, transfer_count
from oracle import dream_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def real():
    with open(os.path.join(HERE, "reference_goldens_real.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(HERE, "reference_goldens_real.npz"))


def _model(cfg_kw, dtype=torch.float32, rich=False):
    cfg = DreamConfig(**cfg_kw)
    m = DreamModel_length(cfg)
    m.load_state_dict(synth.make_weights(cfg, seed=0, rich=rich), strict=True)
    return m.to(DEV).to(dtype)


def _gc(n_in=45):
    return DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                                 eos_token_id=synth.END, sep_token_id=synth.SEP, max_length=n_in + 1)


def _ocfg(kw):
    return O.OracleConfig(**{k: v for k, v in kw.items() if k in O.OracleConfig.__dataclass_fields__})


def _replay(seed):
    """The reference's RNG stream: one exponential_ fill of [M, V] per step from the global CPU generator."""
    torch.manual_seed(seed)
    return lambda step, M, V: torch.empty(M, V).exponential_(1)


NEAR_TIE = 2e-6  # relative; see tests/test_hip_model.py


def _check_against_golden(m, c, ids, hist):
    """Free run first; if it leaves the reference's history, replay under teacher forcing (every step starts from
    the reference's state) and require of every disagreeing step that (a) the sampler agrees -- the token the
    reference wrote is the x0 this build holds for that position -- and (b) the positions the two sides transferred
    instead of each other carry EQUAL confidences up to NEAR_TIE: `torch.topk` on the CPU that made the goldens
    orders exact ties (e.g. entropy == 0 at several positions of a row) differently from a device topk / this
    kernel (lowest index).  Returns True when the free run was bit-equal."""
    kw = dict(inputs=ids.to(DEV), generation_config=_gc(), return_dict_in_generate=True, output_history=True, **c["kw"])
    out = m.diffusion_generate(_noise_fn=_replay(c["seed"]), **kw)
    if np.array_equal(torch.stack(out.history).cpu().numpy(), hist):
        return True
    gold = torch.from_numpy(hist)
    forced, tr = [], {}
    m.diffusion_generate(_noise_fn=_replay(c["seed"]), _trace=tr,
                         _state_hook=lambda i, x: (forced.append(x.cpu().clone()), gold[i])[1], **kw)
    prev = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
    for i, hx in enumerate(forced):
        if not torch.equal(hx, gold[i]):
            mi = prev == synth.MASK
            off = torch.cumsum(mi.sum(1), 0) - mi.sum(1)
            conf, x0 = tr["confidence"][i].cpu(), tr["x0"][i].cpu()
            for b in (hx != gold[i]).any(1).nonzero().flatten().tolist():
                pos = mi[b].nonzero().flatten().tolist()
                cb, xb = conf[off[b]: off[b] + len(pos)], x0[off[b]: off[b] + len(pos)]
                only_h = [q for q, j in enumerate(pos) if hx[b, j] != prev[b, j] and gold[i][b, j] == prev[b, j]]
                only_g = [q for q, j in enumerate(pos) if gold[i][b, j] != prev[b, j] and hx[b, j] == prev[b, j]]
                assert only_h and len(only_h) == len(only_g), (c["name"], i, b)
                for q in only_g:
                    assert int(gold[i][b, pos[q]]) == int(xb[q]), f"{c['name']} step {i} row {b}: sampled token differs"
                both = cb[only_h + only_g]
                assert float(both.max() - both.min()) <= NEAR_TIE * max(1.0, float(both.abs().max())), \
                    f"{c['name']} step {i} row {b}: not a tie {both.tolist()}"
        prev = gold[i]
    return False


def _golden_cases(real, prefix):
    meta, g = real
    return [(c, torch.from_numpy(g[c["name"] + "_ids"]), g[c["name"] + "_hist"]) for c in meta["cases"]
            if c["name"].startswith(prefix)]


# --------------------------------------------------------------------------- config 3 (V = 25427)
def test_config3_vocab_loop_vs_reference_goldens(real):
    """BASELINE config 3's vocabulary (DBpedia50, V = 25427) through the WHOLE loop in fp32 on a 2-layer cut of the
    real width, B = 32, against the reference's own histories (tests/golden/make_goldens_real.py): abduction at
    T = 0.1 and deduction at T = 0.5 (flat rows: the sampler's histogram path at V > 16384).  Equal up to the order of
    exact confidence ties (see _check_against_golden); test_config3_vocab_loop_vs_oracle is exact."""
    meta, _ = real
    m = _model(meta["cfg3"])
    exact = [c["name"] for c, ids, hist in _golden_cases(real, "cfg3_") if _check_against_golden(m, c, ids, hist)]
    print("config 3, free run bit-equal to the reference golden:", exact)


def test_config3_vocab_loop_vs_oracle(real):
    """The same vocabulary against the oracle on injected noise with per-step x0 / confidence compared (T = 0.5)."""
    meta, _ = real
    kw = meta["cfg3"]
    m = _model(kw)
    w = synth.make_weights(m.config, seed=0, rich=False)
    ids = synth.make_inputs(32, "abd", seed=43, **meta["dbpedia"])
    g = torch.Generator().manual_seed(17)
    noises = {}

    def noise_fn(step, M, V):
        if step not in noises:
            noises[step] = torch.empty(M, V).exponential_(1, generator=g)
        return noises[step]

    trace = {}
    out = m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(), return_dict_in_generate=True,
                               output_history=True, steps=16, temperature=0.5, alg="entropy", alg_temp=0, top_p=0.9,
                               _noise_fn=noise_fn, _trace=trace)
    otr = O.SampleTrace()
    _, hist = O.sample_loop(w, _ocfg(kw), ids, max_length=46, steps=16, mask_token_id=synth.MASK, alg="entropy",
                            alg_temp=0, temperature=0.5, top_p=0.9, noise_fn=lambda s, M, V: noises[s], trace=otr,
                            tie_break="index")
    assert trace["k"] == otr.k
    for i in range(16):
        assert torch.equal(trace["x0"][i].cpu(), otr.x0[i]), f"x0 differs at step {i}"
        np.testing.assert_allclose(trace["confidence"][i].cpu().numpy(), otr.confidence[i].numpy(), rtol=1e-3, atol=1e-5)
        assert torch.equal(out.history[i].cpu(), hist[i]), f"history differs at step {i}"


def test_config3_full_size_properties_bf16(real):
    """Config 3 at full per-GPU size (512 deduction rows x 33 masks, 32 steps, bf16, V = 25427): the size-independent
    invariants of the loop, and the same trajectory twice."""
    meta, _ = real
    kw = dict(meta["real"], vocab_size=25427)
    m = _model(kw, torch.bfloat16)
    ids = synth.make_inputs(512, "ded", seed=21, **meta["dbpedia"]).to(DEV)
    runs = []
    for _ in range(2):
        torch.manual_seed(78)
        out = m.diffusion_generate(inputs=ids, generation_config=_gc(), return_dict_in_generate=True,
                                   output_history=True, steps=32, temperature=0.1, alg="entropy", alg_temp=0,
                                   top_p=0.9, skip_noop_steps=False)
        runs.append(torch.stack(out.history))
    assert torch.equal(runs[0], runs[1]), "same seed, same inputs: trajectories must be identical"
    prev = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
    left = 33
    for step, k in enumerate(O.transfer_schedule(33, 32)):
        cur = runs[0][step]
        was_mask = prev == synth.MASK
        assert torch.equal(cur[~was_mask], prev[~was_mask]), f"step {step}: an unmasked position changed"
        left -= k
        cnt = (cur == synth.MASK).sum(1)
        assert bool(((cnt == left) | (cnt == left + 1)).all()), f"step {step}"
        assert bool(((cur >= 0) & (cur < 25427)).all())
        prev = cur


# --------------------------------------------------------------------------- WN18RR (V = 41066 > the LDS-resident sampler)
def test_wn18rr_vocab_loop_vs_oracle(real):
    """The reference's third dataset (configs/config-sampling.yml:3): V = 100 + 40 943 + 22 + 1 = 41 066, whose fp32
    logits row (164 KB) does not fit in LDS -- the sampler runs from the registered row workspace.  Whole loop in fp32
    on a 2-layer cut of the real width against the oracle on injected noise, T = 0.1 (head path) and 0.5 (flat rows)."""
    meta, _ = real
    kw = dict(meta["cfg3"], vocab_size=synth.vocab_size(**synth.WN18RR))
    assert kw["vocab_size"] == 41066 > _hip.SAMPLE_LDS_MAX_V
    m = _model(kw)
    w = synth.make_weights(m.config, seed=0, rich=False)
    for kind, T, B in (("abd", 0.1, 12), ("abd", 0.5, 12)):   # (as test_config3_vocab_loop_vs_oracle: abduction rows)
        ids = synth.make_inputs(B, kind, seed=47 + int(10 * T), **synth.WN18RR)
        g = torch.Generator().manual_seed(19)
        noises = {}

        def noise_fn(step, M, V):
            if step not in noises:
                noises[step] = torch.empty(M, V).exponential_(1, generator=g)
            return noises[step]

        trace = {}
        out = m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(), return_dict_in_generate=True,
                                   output_history=True, steps=12, temperature=T, alg="entropy", alg_temp=0, top_p=0.9,
                                   _noise_fn=noise_fn, _trace=trace)
        otr = O.SampleTrace()
        with O.exact_nucleus():   # at this V the fp32 prefix sums of the reference are rounding-decided (oracle header)
            _, hist = O.sample_loop(w, _ocfg(kw), ids, max_length=46, steps=12, mask_token_id=synth.MASK,
                                    alg="entropy", alg_temp=0, temperature=T, top_p=0.9,
                                    noise_fn=lambda s, M, V: noises[s], trace=otr, tie_break="index")
        assert trace["k"] == otr.k
        for i in range(12):
            assert torch.equal(trace["x0"][i].cpu(), otr.x0[i]), f"{kind}: x0 differs at step {i}"
            np.testing.assert_allclose(trace["confidence"][i].cpu().numpy(), otr.confidence[i].numpy(), rtol=1e-3, atol=1e-5)
            assert torch.equal(out.history[i].cpu(), hist[i]), f"{kind}: history differs at step {i}"


def test_wn18rr_full_depth_bf16_lazy_noise_properties(real):
    """WN18RR's vocabulary at the full 10-layer shape in bf16 with the lazy Philox noise (the product's default path:
    head GEMM at N = 41 066, row-workspace sampler, dark_sample_rows_philox): same seed -> same trajectory, the lazy
    noise equals the materialised fill, and the loop's invariants hold."""
    meta, _ = real
    kw = dict(meta["real"], vocab_size=41066)
    m = _model(kw, torch.bfloat16)
    ids = synth.make_inputs(96, "ded", seed=22, **synth.WN18RR).to(DEV)
    runs = []
    for lazy in (True, True, False):
        torch.manual_seed(79)
        out = m.diffusion_generate(inputs=ids, generation_config=_gc(), return_dict_in_generate=True,
                                   output_history=True, steps=32, temperature=0.1, alg="entropy", alg_temp=0,
                                   top_p=0.9, skip_noop_steps=False, lazy_noise=lazy)
        runs.append(torch.stack(out.history))
    assert torch.equal(runs[0], runs[1]), "same seed, same inputs: trajectories must be identical"
    assert torch.equal(runs[0], runs[2]), "lazy Philox noise must equal torch's materialised exponential_ fill"
    prev = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
    left = 33
    for step, k in enumerate(O.transfer_schedule(33, 32)):
        cur = runs[0][step]
        was_mask = prev == synth.MASK
        assert torch.equal(cur[~was_mask], prev[~was_mask]), f"step {step}: an unmasked position changed"
        left -= k
        cnt = (cur == synth.MASK).sum(1)
        assert bool(((cnt == left) | (cnt == left + 1)).all()), f"step {step}"
        assert bool(((cur >= 0) & (cur < 41066)).all())
        prev = cur


# --------------------------------------------------------------------------- real-width reference goldens
def test_real_shape_deduction_and_config1_goldens(real):
    """Full 10-layer default shape, fp32: a deduction batch (33 masks) at T = 0.1 and T = 0.5 over 32 steps and
    BASELINE config 1 itself (64 abduction rows, 16 steps) against the reference's own histories (equal up to the
    order of exact confidence ties, see _check_against_golden)."""
    meta, _ = real
    m = _model(meta["real"])
    exact = [c["name"] for c, ids, hist in _golden_cases(real, "real_") if _check_against_golden(m, c, ids, hist)]
    print("real width, free run bit-equal to the reference golden:", exact)


# --------------------------------------------------------------------------- config 5 (RL rollout)
def test_config5_repeated_rows_lazy_philox(real):
    """BASELINE config 5's settings: every query repeated x4 (`num_generations`, configs/config-grpo.yml:6), T = 0.5,
    top_p 0.9, 32 steps, the categorical noise evaluated lazily inside the sampler from torch's Philox stream.  The
    four copies of a query must follow different trajectories, and the whole batch must equal the oracle fed the SAME
    stream, materialised with torch's own `exponential_` on the device from the same seed (2-layer cut, fp32)."""
    meta, _ = real
    kw = dict(meta["real"], num_hidden_layers=2)
    m = _model(kw)
    w = synth.make_weights(m.config, seed=0, rich=False)
    q = synth.make_inputs(6, "abd", seed=9)
    ids = q.repeat_interleave(4, dim=0)
    torch.manual_seed(31)
    out = m.diffusion_generate(inputs=q.to(DEV), generation_config=_gc(), num_return_sequences=4,
                               return_dict_in_generate=True, output_history=True, steps=32, temperature=0.5,
                               alg="entropy", alg_temp=0, top_p=0.9, lazy_noise=True)
    got = torch.stack(out.history).cpu()
    torch.manual_seed(31)  # the same device stream, drawn the way the reference draws it

    def noise_fn(step, M, V):
        return torch.empty((M, V), device=DEV, dtype=torch.float32).exponential_().cpu()

    _, hist = O.sample_loop(w, _ocfg(kw), ids, max_length=46, steps=32, mask_token_id=synth.MASK, alg="entropy",
                            alg_temp=0, temperature=0.5, top_p=0.9, noise_fn=noise_fn, tie_break="index")
    exp = torch.stack(hist)
    bad = (got != exp).any(-1).any(-1).nonzero().flatten().tolist()
    assert not bad, f"history differs from the oracle on the same Philox stream at steps {bad}"
    final = got[-1].view(6, 4, 46)
    distinct = sum(len({tuple(r.tolist()) for r in final[i]}) > 1 for i in range(6))
    assert distinct >= 5, "the copies of a query must sample different trajectories"


# --------------------------------------------------------------------------- dark_denoise_step
def test_denoise_step_entry_point(real):
    """`dark_denoise_step` (one C call per step for hosts that track the masked-row count themselves) against the
    oracle and against `diffusion_generate` (the sequenced entry points): sequences after every step, x0 and
    confidence read back through `dark_step_workspace_offsets`, bit for bit."""
    meta, _ = real
    kw = dict(meta["real"], num_hidden_layers=2)
    m = _model(kw)
    w = synth.make_weights(m.config, seed=0, rich=False)
    lib, pk = _hip.load(), m.packed()
    ids = synth.make_inputs(8, "ded", seed=13)
    B, N, V, steps = 8, 46, kw["vocab_size"], 6
    g = torch.Generator().manual_seed(3)
    noises = [torch.empty(B * 33, V).exponential_(1, generator=g) for _ in range(steps)]
    otr = O.SampleTrace()
    _, hist = O.sample_loop(w, _ocfg(kw), ids, max_length=N, steps=steps, mask_token_id=synth.MASK, alg="entropy",
                            alg_temp=0, temperature=0.3, top_p=0.9, noise_fn=lambda s, M, V_: noises[s][:M], trace=otr,
                            tie_break="index")
    seq = m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(), return_dict_in_generate=True,
                               output_history=True, steps=steps, temperature=0.3, alg="entropy", alg_temp=0, top_p=0.9,
                               _noise_fn=lambda s, M, V_: noises[s][:M])
    x = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK).to(DEV).contiguous()
    ts = torch.linspace(1, 1e-3, steps + 1)
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(steps):
        M = int((x == synth.MASK).sum())
        k = transfer_count(M, B, ts[i], ts[i + 1], i == steps - 1)
        assert k == otr.k[i]
        nbytes = lib.dark_step_workspace_bytes(pk.ref, B, N, M)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        off = (C.c_size_t * 7)()
        _hip.check(lib.dark_step_workspace_offsets(pk.ref, B, N, M, off))
        noise = noises[i][:M].to(DEV).contiguous()
        _hip.check(lib.dark_denoise_step(pk.ref, _hip.ptr(x), None, None, B, N, M, synth.MASK, 0.3, 0.9, 0,
                                         _hip.CONF_NEG_ENTROPY, _hip.ptr(noise), k, _hip.ptr(ws), nbytes, stream))
        torch.cuda.synchronize()
        conf = ws[off[5]: off[5] + 4 * M].view(torch.float32).cpu()
        x0 = ws[off[6]: off[6] + 8 * M].view(torch.int64).cpu()
        cnt = ws[off[3]: off[3] + 4].view(torch.int32).cpu()
        assert int(cnt[0]) == M
        assert torch.equal(x0, otr.x0[i]), f"step {i}: x0"
        np.testing.assert_allclose(conf.numpy(), otr.confidence[i].numpy(), rtol=1e-3, atol=1e-5)
        assert torch.equal(x.cpu(), hist[i]), f"step {i}: sequences vs oracle"
        assert torch.equal(x, seq.history[i]), f"step {i}: sequences vs the sequenced entry points"
    with pytest.raises(_hip.DarkHipError):  # workspace too small is refused, not overrun
        _hip.check(lib.dark_denoise_step(pk.ref, _hip.ptr(x), None, None, B, N, 8, synth.MASK, 0.0, -1.0, 0, 2, None, 1,
                                         _hip.ptr(ws), 1024, stream))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_rows_equals_full_forward(real, dtype):
    """dark_forward_hidden_begin + dark_forward_hidden_rows (the last layer's o_proj / MLP for the rows the head reads
    only) against dark_forward_hidden: the selected rows of the residual stream bit for bit, padded attention mask and
    duplicate / unsorted row indices included; 0 and 1 layers as edge cases."""
    meta, _ = real
    lib = _hip.load()
    for L in (3, 1, 0):
        m = _model(dict(meta["real"], num_hidden_layers=L), dtype)
        pk = m.packed()
        B, N, d = 24, 46, 768
        x = torch.nn.functional.pad(synth.make_inputs(B, "ded", seed=6), (0, 1), value=synth.MASK).to(DEV).contiguous()
        km = torch.ones(B, N, dtype=torch.uint8, device=DEV)
        km[1, 30:40] = 0
        pos = (km.long().cumsum(-1) - 1).masked_fill(km == 0, 1).to(torch.int32).contiguous()
        g = torch.Generator().manual_seed(L)
        rows = torch.randint(0, B * N, (389,), generator=g).to(torch.int32)
        rows[5] = rows[9]
        rows_d = rows.to(DEV)
        nb = lib.dark_forward_workspace_bytes(pk.ref, B, N)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        stream = torch.cuda.current_stream().cuda_stream
        full = torch.empty(B * N, d, device=DEV, dtype=pk.resid_dtype)   # (bf16 models carry a bf16 residual stream)
        _hip.check(lib.dark_forward_hidden(pk.ref, _hip.ptr(x), _hip.ptr(pos), _hip.ptr(km), B, N, _hip.ptr(ws), nb,
                                           _hip.ptr(full), stream))
        part = torch.full((B * N, d), float("nan"), device=DEV, dtype=pk.resid_dtype)
        out = torch.full((rows.numel(), d), float("nan"), device=DEV, dtype=pk.resid_dtype)
        _hip.check(lib.dark_forward_hidden_begin(pk.ref, _hip.ptr(x), _hip.ptr(pos), _hip.ptr(km), B, N, _hip.ptr(ws),
                                                 nb, _hip.ptr(part), stream))
        _hip.check(lib.dark_forward_hidden_rows(pk.ref, B, N, _hip.ptr(rows_d), rows.numel(), _hip.ptr(ws), nb,
                                                _hip.ptr(part), _hip.ptr(out), stream))
        torch.cuda.synchronize()
        assert torch.equal(out, full[rows_d.long()]), (L, dtype)


@pytest.mark.parametrize("alg,alg_temp,T", [("origin", None, 0.4), ("origin", None, 0.0), ("entropy", 0.5, 0.3),
                                            ("maskgit_plus", 0.7, 0.5), ("topk_margin", 0.2, 0.0)])
def test_native_variants_equal_materialised_path(real, alg, alg_temp, T):
    """alg='origin' and alg_temp > 0 run on device kernels (dark_scatter_tokens / dark_commit_sampled, the draws taken
    from torch's Philox stream); a no-op logits hook forces the materialised [B,N,V] path that calls torch.rand /
    torch.multinomial itself.  Same seed: the histories and the generator's final offset must be identical."""
    meta, _ = real
    m = _model(dict(meta["real"], num_hidden_layers=2))
    ids = synth.make_inputs(24, "ded", seed=12).to(DEV)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    res = []
    for hook in (None, (lambda step, x, logits: logits)):
        torch.manual_seed(41)
        kw = {} if hook is None else {"generation_logits_hook_func": hook}
        out = m.diffusion_generate(ids, generation_config=_gc(), return_dict_in_generate=True, output_history=True,
                                   steps=10, temperature=T, alg=alg, alg_temp=alg_temp, top_p=0.9, **kw)
        res.append((torch.stack(out.history).cpu(), gen.get_offset()))
    bad = (res[0][0] != res[1][0]).any(-1).any(-1).nonzero().flatten().tolist()
    assert not bad, f"native and materialised paths differ at steps {bad}"
    assert res[0][1] == res[1][1], "the device generator must end at the same offset"
    assert int((res[0][0][-1] == synth.MASK).sum()) <= 2


# --------------------------------------------------------------------------- boundary behaviour
def test_int32_ids_are_accepted(real):
    """nn.Embedding in the reference takes int32 ids too (modeling_dream_length.py:665): the boundary casts once."""
    meta, _ = real
    m = _model(dict(meta["real"], num_hidden_layers=1))
    ids = synth.make_inputs(4, "abd", seed=3).to(DEV)
    a = m.diffusion_generate(ids, generation_config=_gc(), steps=4, alg="entropy", temperature=0.0)
    b = m.diffusion_generate(ids.to(torch.int32), generation_config=_gc(), steps=4, alg="entropy", temperature=0.0)
    assert b.dtype == torch.int64 and torch.equal(a, b)
    x = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
    assert torch.equal(m(x).logits, m(x.to(torch.int32)).logits)


def test_think_sample_generator_accounting_and_index_error(real):
    """check_and_select (generation_utils.py:411-432) draws two discarded Categorical samples over [4B, N, V] per
    think step: the device generator must advance by exactly those two fills on top of the sampler's own draws, and a
    row 0 without SEP raises the reference's IndexError (:419)."""
    meta, _ = real
    m = _model(dict(meta["real"], num_hidden_layers=1))
    ids = synth.make_inputs(4, "abd", seed=3).to(DEV)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    B, N, V, steps, interval = 4, 46, meta["real"]["vocab_size"], 6, 2
    offs = []
    for think in (False, True):
        torch.manual_seed(5)
        # greedy sampling: the only generator traffic is check_and_select's own (its temperature defaults to 1.0)
        m.diffusion_generate(ids, generation_config=_gc(), steps=steps, alg="entropy", temperature=0.0, top_p=0.9,
                             think_sample=think, think_sample_interval=interval)
        offs.append(gen.get_offset())
    torch.manual_seed(5)
    base = gen.get_offset()
    n_think = len(range(0, steps, interval))
    for _ in range(2 * n_think):
        philox_fill_plan(torch.device(DEV), 4 * B * N * V)
    assert offs[1] - offs[0] == gen.get_offset() - base > 0
    no_sep = ids.clone()
    no_sep[0, 11] = synth.END
    with pytest.raises(IndexError):
        m.diffusion_generate(no_sep, generation_config=_gc(), steps=2, alg="entropy", temperature=0.0,
                             think_sample=True)


def test_linspace_on_device_equals_host():
    """The k-schedule's timesteps are a host linspace here and `torch.linspace(..., device=x.device)` in the reference
    (generation_utils.py:484): the two must hold the same fp32 values, or k could differ by one at a truncation edge."""
    for steps in (1, 2, 8, 16, 32, 64, 100, 512):
        for eps in (1e-3, 1e-2):
            a = torch.linspace(1, eps, steps + 1)
            b = torch.linspace(1, eps, steps + 1, device=DEV).cpu()
            assert torch.equal(a, b), (steps, eps)
            ka = [int(torch.tensor(33.0) * (1 - a[i + 1] / a[i])) for i in range(steps)]
            kb = [int(torch.tensor(33.0, device=DEV) * (1 - b[i + 1].to(DEV) / b[i].to(DEV))) for i in range(steps)]
            assert ka == kb


def test_model_on_non_current_device():
    """Kernel attributes and streams are per device: a model on cuda:1 must work while cuda:0 is current."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    kw = dict(vocab_size=15080, hidden_size=768, intermediate_size=3072, num_hidden_layers=1, num_attention_heads=8,
              num_key_value_heads=8, max_position_embeddings=64, mask_token_id=synth.MASK, pad_token_id=synth.END)
    cfg = DreamConfig(**kw)
    w = synth.make_weights(cfg, seed=0)
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        m = DreamModel_length(cfg)
        m.load_state_dict(w, strict=True)
        m = m.to(dev)
        torch.cuda.set_device(0)
        ids = synth.make_inputs(4, "abd", seed=3).to(dev)
        outs.append(m.diffusion_generate(ids, generation_config=_gc(), steps=4, alg="entropy", temperature=0.0).cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_lm_head_rows_padded_stride(dtype):
    """dark_lm_head_rows on a vocabulary that is not a multiple of four (DBpedia50: 25 427): on a stride that leaves room
    the GEMM runs over the next multiple of four (the persistent kernel's 16-byte stores) -- the first V columns are the
    same bits as on the tight stride, the columns up to that multiple hold the zero logits of lm_head's padding rows,
    everything beyond is untouched (include/dark_hip.h)."""
    lib = _hip.load()
    V = 25427
    m = _model(dict(vocab_size=V, hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=4,
                    num_key_value_heads=4, max_position_embeddings=64, mask_token_id=synth.MASK, pad_token_id=synth.END),
               dtype)
    pk = m.packed()
    M = 2100
    hidden = (torch.randn(M, 256, generator=torch.Generator().manual_seed(3)) * 2).to(DEV).to(pk.resid_dtype)
    s = _hip.stream_ptr(torch.device(DEV))
    ws_b = lib.dark_head_workspace_bytes(pk.ref, M)
    ws = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
    outs = {}
    for ldl in (V, V + 1, V + 5):
        buf = torch.full((M, ldl), 7.0, dtype=torch.float32, device=DEV)
        _hip.check(lib.dark_lm_head_rows(pk.ref, _hip.ptr(hidden), None, M, _hip.ptr(ws), ws_b, _hip.ptr(buf), ldl, s))
        torch.cuda.synchronize()
        outs[ldl] = buf.cpu()
    assert bool((outs[V] != 7.0).all())
    for ldl in (V + 1, V + 5):
        assert torch.equal(outs[ldl][:, :V], outs[V]), ldl
        assert bool((outs[ldl][:, V:V + 1] == 0.0).all()), "the column up to the multiple of four: zero logits"
        assert bool((outs[ldl][:, V + 1:] == 7.0).all()), "columns beyond stay untouched"
