"""GPU parity tests of the whole path behind the reference's Python surface
(DreamModel_length.forward / diffusion_generate) against the committed reference goldens and the oracle."""
import numpy as np
import pytest
import torch

from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, DreamModelOutput, synth
from oracle import dream_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(meta_cfg, rich, dtype=torch.float32):
    cfg = DreamConfig(**meta_cfg)
    m = DreamModel_length(cfg)
    m.load_state_dict(synth.make_weights(cfg, seed=0, rich=rich), strict=True)
    return m.to(DEV).to(dtype)


@pytest.fixture(scope="module")
def mini(golden_meta):
    return _model(golden_meta["mini"], True)


def _gc(n_in):
    return DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                                 eos_token_id=synth.END, expand_token_id=synth.EXPAND, delete_token_id=synth.END,
                                 sep_token_id=synth.SEP, max_length=n_in + 1)


def _cpu_noise(seed):
    """Replays the reference's RNG stream: it draws exponential_([M,V]) from the global CPU generator once
    per step (and rand(M) first for alg='origin')."""
    torch.manual_seed(seed)
    return ((lambda step, M, V: torch.empty(M, V).exponential_(1)), (lambda step, M: torch.rand(M)),
            (lambda step, B, N: torch.empty(B, N).exponential_(1)))  # torch.multinomial's own draw (alg_temp > 0)


def test_forward_logits_fp32(mini, goldens):
    ids = torch.from_numpy(goldens["fwd_ids"]).to(DEV)
    B, N = ids.shape
    out = mini(ids, torch.ones(B, 1, N, N, dtype=torch.bool, device=DEV), torch.arange(N, device=DEV).expand(B, -1))
    ref = goldens["fwd_logits"]
    got = out.logits.cpu().numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-5)   # north-star bound: 1e-3 relative
    assert np.abs(got - ref).max() < 2e-5                          # what fp32 MFMA actually achieves


def test_forward_padded_mask_fp32(mini, goldens):
    ids = torch.from_numpy(goldens["fwd_ids"]).to(DEV)
    am = torch.from_numpy(goldens["fwdpad_mask"]).to(DEV)
    tok = am.cumsum(-1) - 1
    tok.masked_fill_(am == 0, 1)
    m4 = torch.logical_and(am.unsqueeze(1).unsqueeze(-2), am.unsqueeze(1).unsqueeze(-1))
    got = mini(ids, m4, tok).logits.cpu().numpy()
    ref = goldens["fwdpad_logits"]
    live = goldens["fwdpad_mask"].astype(bool)
    # rows of masked-out queries: the reference's value there depends on SDPA's all-masked-row handling and is
    # never consumed (those positions are padding); compare the live rows
    np.testing.assert_allclose(got[live], ref[live], rtol=1e-3, atol=2e-5)


def test_forward_bf16(golden_meta, goldens):
    m = _model(golden_meta["mini"], True, torch.bfloat16)
    ids = torch.from_numpy(goldens["fwd_ids"]).to(DEV)
    got = m(ids).logits.float().cpu().numpy()
    ref = goldens["fwd_logits"]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 3e-2, err  # stated bf16 bound: 3% of the logit range


def _loop_ids(case, goldens):
    if "ids" in case:
        return torch.from_numpy(goldens[case["ids"]])
    return torch.from_numpy(goldens["loop_abd_ids" if case["name"].startswith("abd") else "loop_ded_ids"])


TIE_FREE = {"abd_t01", "abd_t05_s32", "ded_t01", "abd_maskgit", "abd_margin", "abd_origin", "abd_padmask", "ded_algtemp"}
# Pinned from the MI355X runs (round 2): free runs bit-equal to the reference's goldens, and the only cases allowed to
# fall back to the teacher-forced near-tie check (flat mini-model distributions at T >= 0.5 / top-k / ragged greedy).
GOLDEN_EXACT = {"abd_margin", "abd_maskgit", "abd_origin", "abd_padmask", "abd_t01", "ded_t01", "ded_algtemp"}
ORACLE_EXACT = GOLDEN_EXACT | {"abd_greedy"}   # exact ties at the k-th place: equal to the oracle's device tie order
NEAR_TIE_CASES = {"abd_t05_s32", "ded_t05_s8", "ded_topk", "rag_t0"}
NEAR_TIE = 2e-6  # relative: ~16 ulp of fp32; confidences are sums over V terms, so implementations differ by a few ulp


def _oracle_loop(w, cfg, ids, am, kw, seed, trace=None):
    torch.manual_seed(seed)
    _, oh = O.sample_loop(w, cfg, ids, max_length=ids.shape[1] + 1, mask_token_id=synth.MASK, attention_mask=am,
                          steps=kw["steps"], alg=kw["alg"], alg_temp=kw.get("alg_temp"), temperature=kw["temperature"],
                          top_p=kw.get("top_p"), top_k=kw.get("top_k"), tie_break="index", trace=trace)
    return oh


def _explained_by_near_tie(prev, hip, ora, conf, k):
    """A step where HIP and the oracle commit different positions is acceptable only if the oracle's own
    confidences of the positions they disagree on are within NEAR_TIE of each other (an fp32 rounding tie)."""
    mi = prev == synth.MASK
    off = torch.cumsum(mi.sum(1), 0) - mi.sum(1)
    for b in (hip != ora).any(1).nonzero().view(-1).tolist():
        pos = mi[b].nonzero().view(-1)
        c = conf[off[b]: off[b] + pos.numel()]
        only_h = [i for i, j in enumerate(pos.tolist()) if hip[b, j] != prev[b, j] and ora[b, j] == prev[b, j]]
        only_o = [i for i, j in enumerate(pos.tolist()) if ora[b, j] != prev[b, j] and hip[b, j] == prev[b, j]]
        if not only_h or len(only_h) != len(only_o):
            return False
        lo, hi = min(c[only_h + only_o]), max(c[only_h + only_o])
        if float(hi - lo) > NEAR_TIE * max(1.0, float(hi.abs())):
            return False
    return True


def test_loops_match_reference_goldens(mini, goldens, golden_meta):
    """Token ids and unmask order after every step, fp32, recorded seed.

    Free-running histories must equal the oracle run with device-topk tie semantics (lowest index among exactly
    equal confidences, see oracle.sample_loop `tie_break`), and, for the cases where no exact tie straddles the
    k-th place (tests/test_oracle_golden.py::test_index_tie_break_differs_only_at_ties), the reference's own CPU
    goldens bit for bit.  If a free run diverges, the case is replayed under teacher forcing (every step starts
    from the oracle's state) and each disagreeing step must be an fp32 near-tie between the candidates
    (NEAR_TIE).  The mini model (V=512, 2 layers, random weights) gives near-flat, near-identical distributions at
    neighbouring positions, so ulp-level ties are common there; WHICH cases are bit-exact against the reference
    goldens / the oracle and which may take the near-tie route is pinned by name (GOLDEN_EXACT, ORACLE_EXACT,
    NEAR_TIE_CASES), and the real-shape tests below must be exact."""
    cfg = O.OracleConfig(**{k: v for k, v in golden_meta["mini"].items() if k in O.OracleConfig.__dataclass_fields__})
    w = synth.make_weights(mini.config, seed=0, rich=True)
    checked = 0
    golden_exact, oracle_exact, near_tie = [], [], []
    for case in golden_meta["cases"]["loops"]:
        name, kw = case["name"], dict(case["kw"])
        ids_cpu = _loop_ids(case, goldens)
        am_cpu = torch.from_numpy(goldens[case["attention_mask"]]) if "attention_mask" in case else None
        extra = dict(attention_mask=am_cpu.to(DEV)) if am_cpu is not None else {}

        def run(**priv):
            noise_fn, rand_fn, rank_fn = _cpu_noise(case["seed"])
            return mini.diffusion_generate(inputs=ids_cpu.to(DEV), generation_config=_gc(ids_cpu.shape[1]),
                                           return_dict_in_generate=True, output_history=True,
                                           number_transfer_tokens=1, _noise_fn=noise_fn, _rand_fn=rand_fn,
                                           _rank_noise_fn=rank_fn, **priv, **extra, **kw)

        out = run()
        assert isinstance(out, DreamModelOutput)
        hist = torch.stack(out.history).cpu()
        otr = O.SampleTrace()
        oh = torch.stack(_oracle_loop(w, cfg, ids_cpu, am_cpu, kw, case["seed"], otr))
        checked += 1
        if torch.equal(hist, oh):
            oracle_exact.append(name)
            if name in TIE_FREE:
                assert np.array_equal(hist.numpy(), goldens[name + "_hist"]), f"{name} vs reference golden"
                golden_exact.append(name)
            continue
        assert kw["alg"] != "origin", name
        near_tie.append(name)
        print(f"{name}: free run diverges from the oracle; checking every step under teacher forcing")
        forced = []
        run(_state_hook=lambda i, x: (forced.append(x.cpu().clone()), oh[i])[1])
        x_init = torch.nn.functional.pad(ids_cpu, (0, 1), value=synth.MASK)
        for i, hx in enumerate(forced):
            if not torch.equal(hx, oh[i]):
                prev = oh[i - 1] if i else x_init
                assert _explained_by_near_tie(prev, hx, oh[i], otr.confidence[i], otr.k[i]), f"{name} step {i}"
    print(f"loops: {checked} cases; bit-exact vs reference goldens {sorted(golden_exact)}; bit-exact vs the oracle "
          f"{sorted(oracle_exact)}; fp32 near-ties {sorted(near_tie)}")
    # Pinned: a case that stops being bit-exact is a regression even if its steps are still near-ties.
    assert checked == 12
    assert set(golden_exact) >= GOLDEN_EXACT, sorted(GOLDEN_EXACT - set(golden_exact))
    assert set(oracle_exact) >= ORACLE_EXACT, sorted(ORACLE_EXACT - set(oracle_exact))
    assert set(near_tie) <= NEAR_TIE_CASES, sorted(set(near_tie) - NEAR_TIE_CASES)


def test_alg_temp_property(mini, goldens):
    ids = torch.from_numpy(goldens["loop_ded_ids"]).to(DEV)
    torch.manual_seed(0)
    out = mini.diffusion_generate(inputs=ids, generation_config=_gc(45), return_dict_in_generate=True,
                                  output_history=True, steps=10, temperature=0.3, alg="entropy", alg_temp=0.5,
                                  top_p=0.9)
    sched = O.transfer_schedule(33, 10)
    left = 33
    for h, k in zip(out.history, sched):
        left -= k
        cnt = (h == synth.MASK).sum(1).cpu()
        assert bool((cnt >= left).all()) and bool((cnt <= left + 2).all())  # +: a sampled id may equal MASK


def test_real_shape_history_fp32(goldens, golden_meta):
    m = _model(golden_meta["real"], False)
    ids = torch.from_numpy(goldens["real_abd_ids"]).to(DEV)
    x = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
    lg = m(x[:1]).logits[0, :3].cpu().numpy()
    ref = goldens["real_logits_row0_pos0_3"]
    np.testing.assert_allclose(lg, ref, rtol=1e-3, atol=5e-5)
    for name, seed, kw in [("real_abd_t01_hist", 2, dict(steps=16, temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9)),
                           ("real_abd_greedy_hist", 2, dict(steps=8, temperature=0.0, alg="entropy", alg_temp=0, top_p=0.9))]:
        noise_fn = _cpu_noise(seed)[0]
        out = m.diffusion_generate(inputs=ids, generation_config=_gc(45), return_dict_in_generate=True,
                                   output_history=True, _noise_fn=noise_fn, **kw)
        assert np.array_equal(torch.stack(out.history).cpu().numpy(), goldens[name]), name


def test_config1_batch64_vs_oracle(golden_meta):
    """BASELINE config 1 shape (B=64, 16 steps, T=0.1, top_p=0.9) on a 2-layer cut of the real width, against the
    oracle run on the same weights and the same noise: histories identical, per-step confidences close."""
    kw = dict(golden_meta["real"], num_hidden_layers=2)
    m = _model(kw, False)
    cfg = O.OracleConfig(**{k: v for k, v in kw.items() if k in O.OracleConfig.__dataclass_fields__})
    w = synth.make_weights(m.config, seed=0, rich=False)
    ids = synth.make_inputs(64, "abd", seed=1)
    g = torch.Generator().manual_seed(99)
    noises = {}

    def noise_fn(step, M, V):
        if step not in noises:
            noises[step] = torch.empty(M, V).exponential_(1, generator=g)
        return noises[step]

    trace = {}
    out = m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(45), return_dict_in_generate=True,
                               output_history=True, steps=16, temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9,
                               _noise_fn=noise_fn, _trace=trace)
    otr = O.SampleTrace()
    seq, hist = O.sample_loop(w, cfg, ids, max_length=46, steps=16, mask_token_id=synth.MASK, alg="entropy", alg_temp=0,
                              temperature=0.1, top_p=0.9, noise_fn=lambda s, M, V: noises[s], trace=otr,
                              tie_break="index")
    assert trace["k"] == otr.k == O.transfer_schedule(11, 16)
    for i in range(16):
        assert torch.equal(trace["x0"][i].cpu(), otr.x0[i]), f"x0 differs at step {i}"
        np.testing.assert_allclose(trace["confidence"][i].cpu().numpy(), otr.confidence[i].numpy(), rtol=1e-3, atol=1e-5)
        assert torch.equal(out.history[i].cpu(), hist[i]), f"history differs at step {i}"


def test_alg_temp_real_width_vs_oracle(golden_meta):
    """`alg_temp > 0` (multinomial ranking, generation_utils.py:522-530) at the real width, fp32, with the categorical
    draws AND the ranking draws replayed: histories equal the oracle's.  The kernel forms softmax(conf / alg_temp) in
    fp32 with its own summation order, so a step may differ from the oracle only where two quotients p / q of the
    positions in dispute are within an fp32 rounding of each other (checked under teacher forcing).  bf16
    models rank in fp32 as well, where the reference would rank in bf16: no reference fixture covers that, parity
    unpinned (Commit)."""
    kw = dict(golden_meta["real"], num_hidden_layers=2)
    m = _model(kw, False)
    cfg = O.OracleConfig(**{k: v for k, v in kw.items() if k in O.OracleConfig.__dataclass_fields__})
    w = synth.make_weights(m.config, seed=0, rich=False)
    ids = synth.make_inputs(32, "ded", seed=4)
    g = torch.Generator().manual_seed(123)
    noises, ranks = {}, {}

    def noise_fn(step, M, V):
        if step not in noises:
            noises[step] = torch.empty(M, V).exponential_(1, generator=g)
        return noises[step]

    def rank_fn(step, B, N):
        if step not in ranks:
            ranks[step] = torch.empty(B, N).exponential_(1, generator=g)
        return ranks[step]

    steps, alg_temp = 12, 0.5
    gen = dict(steps=steps, temperature=0.1, alg="entropy", alg_temp=alg_temp, top_p=0.9)
    otr = O.SampleTrace()
    # the oracle first: it draws noise and ranks in the order the steps need them, the device run replays the same tensors
    _, oh = O.sample_loop(w, cfg, ids, max_length=46, mask_token_id=synth.MASK, noise_fn=noise_fn, rank_noise_fn=rank_fn,
                          trace=otr, tie_break="index", **gen)
    out = m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(45), return_dict_in_generate=True,
                               output_history=True, _noise_fn=noise_fn, _rank_noise_fn=rank_fn, **gen)
    hist = [h.cpu() for h in out.history]
    if all(torch.equal(a, b) for a, b in zip(hist, oh)):
        return
    forced = []
    m.diffusion_generate(inputs=ids.to(DEV), generation_config=_gc(45), return_dict_in_generate=True, output_history=True,
                         _noise_fn=noise_fn, _rank_noise_fn=rank_fn,
                         _state_hook=lambda i, x: (forced.append(x.cpu().clone()), oh[i])[1], **gen)
    x_init = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
    for i, hx in enumerate(forced):
        if torch.equal(hx, oh[i]):
            continue
        prev = oh[i - 1] if i else x_init
        mi = prev == synth.MASK
        full = torch.full(prev.shape, float("-inf"))
        full[mi] = otr.confidence[i]
        quot = torch.softmax(full / alg_temp, dim=-1) / ranks[i]
        for b in (hx != oh[i]).any(1).nonzero().view(-1).tolist():
            dis = ((hx[b] != prev[b]) ^ (oh[i][b] != prev[b])).nonzero().view(-1)
            assert dis.numel() >= 2, (i, b)
            qd = quot[b, dis]
            assert float(qd.max() - qd.min()) <= 1e-5 * float(qd.max()), (i, b, qd.tolist())


def test_bf16_decisions_close_to_fp32(golden_meta):
    """Stated bf16 bound (the north star asks for one): from the SAME state and the SAME noise, one denoise step in
    bf16 must pick the same token as fp32 for >= 97% of the masked positions and confidences within 0.05 absolute
    for >= 95% of them.  (Whole trajectories are not compared: one flipped unmask order changes every later
    context, so sequence-level agreement measures chaos, not kernel error.)"""
    kw = dict(golden_meta["real"], num_hidden_layers=4)
    ids = synth.make_inputs(64, "ded", seed=3).to(DEV)
    g = torch.Generator().manual_seed(5)
    noise = torch.empty(64 * 33, kw["vocab_size"]).exponential_(1, generator=g)
    res = []
    for dt in (torch.float32, torch.bfloat16):
        m = _model(kw, False, dt)
        tr = {}
        m.diffusion_generate(inputs=ids, generation_config=_gc(45), steps=1, temperature=0.1, alg="entropy",
                             alg_temp=0, top_p=0.9, _noise_fn=lambda s, M, V: noise, _trace=tr)
        res.append((tr["x0"][0].cpu(), tr["confidence"][0].cpu()))
        del m
    same = (res[0][0] == res[1][0]).float().mean().item()
    close = ((res[0][1] - res[1][1]).abs() < 0.05).float().mean().item()
    print(f"bf16 vs fp32, one step, {res[0][0].numel()} rows: token agreement {same:.4f}, |dconf|<0.05 {close:.4f}")
    assert same >= 0.97, same
    assert close >= 0.95, close


def test_bf16_residual_stream_modes(golden_meta):
    """bf16 models carry the residual stream in bf16 by default (what the reference's bf16 model does,
    modeling_dream_length.py:508, :514) and in fp32 with `residual_in_fp32 = True`.  Both against the fp32 model of the
    same weights at the real width: each within the stated bf16 bound (3 % of the logit range), and the bf16 stream
    within 2x the error of the fp32 stream (the other bf16 roundings -- operands, activations -- dominate)."""
    kw = dict(golden_meta["real"], num_hidden_layers=4)
    ids = torch.nn.functional.pad(synth.make_inputs(16, "ded", seed=4), (0, 1), value=synth.MASK).to(DEV)
    ref = _model(kw, False, torch.float32)(ids).logits.float().cpu()
    errs = {}
    for fp32_stream in (False, True):
        m = _model(kw, False, torch.bfloat16)
        m.residual_in_fp32 = fp32_stream
        got = m(ids).logits.float().cpu()
        assert m.packed().resid_dtype == (torch.float32 if fp32_stream else torch.bfloat16)
        errs[fp32_stream] = ((got - ref).abs().max() / ref.abs().max()).item()
        del m
    print(f"bf16 logits vs fp32, max error / logit range: bf16 stream {errs[False]:.4f}, fp32 stream {errs[True]:.4f}")
    assert errs[False] < 3e-2 and errs[True] < 3e-2, errs
    assert errs[False] < 2.0 * errs[True] + 1e-3, errs


def test_api_contract(mini, goldens):
    ids = torch.from_numpy(goldens["loop_abd_ids"]).to(DEV)
    before = ids.clone()
    # bare tensor by default (generation_utils.py:547-548), input not mutated, unknown kwargs ignored
    seq = mini.diffusion_generate(ids, generation_config=_gc(45), steps=4, alg="entropy", number_transfer_tokens=1)
    assert isinstance(seq, torch.Tensor) and seq.shape == (8, 46) and seq.dtype == torch.long and seq.is_cuda
    assert torch.equal(ids, before)
    with pytest.raises(ValueError):      # input length >= max_length (:186-192)
        mini.diffusion_generate(ids, generation_config=_gc(44), steps=2, alg="entropy")
    with pytest.raises(RuntimeError):    # unknown alg (:518-519)
        mini.diffusion_generate(ids, generation_config=_gc(45), steps=2, alg="nope")
    with pytest.raises(AssertionError):  # inputs is None (:313)
        mini.diffusion_generate(None, generation_config=_gc(45))
    # num_return_sequences expands the batch (:352-356); hooks are called with (None, x, None) then per step
    calls = []

    def tok_hook(step, x, logits):
        calls.append((step, None if logits is None else tuple(logits.shape)))
        return x

    out = mini.diffusion_generate(ids[:2], generation_config=_gc(45), steps=3, alg="entropy", temperature=0.2,
                                  num_return_sequences=2, return_dict_in_generate=True, output_history=True,
                                  generation_tokens_hook_func=tok_hook,
                                  generation_logits_hook_func=lambda step, x, logits: logits)
    assert out.sequences.shape == (4, 46) and len(out.history) == 3
    assert calls[0] == (None, None) and calls[1] == (0, (4, 46, 512)) and len(calls) == 4
    # max_new_tokens path
    seq = mini.diffusion_generate(ids[:2], generation_config=DreamGenerationConfig(mask_token_id=synth.MASK),
                                  max_new_tokens=3, steps=2, alg="maskgit_plus")
    assert seq.shape == (2, 48)


def test_save_and_from_pretrained_roundtrip(mini, tmp_path, goldens):
    mini.generation_config = _gc(45)
    mini.save_pretrained(str(tmp_path))
    m2 = DreamModel_length.from_pretrained(str(tmp_path)).to(DEV)
    assert m2.generation_config.mask_token_id == synth.MASK and m2.generation_config.max_length == 46
    ids = torch.from_numpy(goldens["fwd_ids"]).to(DEV)
    assert torch.equal(m2(ids).logits, mini(ids).logits)


def test_skip_noop_steps_is_invisible(golden_meta):
    """skip_noop_steps (default on) must not change tokens, history or the torch RNG stream: with 11 masks and
    32 steps 22 steps commit nothing; their forward is reused, their noise draw is kept."""
    kw = dict(golden_meta["real"], num_hidden_layers=2)
    m = _model(kw, False)
    ids = synth.make_inputs(16, "abd", seed=7).to(DEV)
    res = []
    for skip in (False, True):
        torch.manual_seed(123)
        out = m.diffusion_generate(inputs=ids, generation_config=_gc(45), return_dict_in_generate=True,
                                   output_history=True, steps=32, temperature=0.5, alg="entropy", alg_temp=0,
                                   top_p=0.9, skip_noop_steps=skip)
        res.append((torch.stack(out.history).cpu(), torch.rand(4, device=DEV).cpu()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]), "the device generator must end in the same state"
    changed = (res[0][0][1:] != res[0][0][:-1]).any(-1).any(-1)
    assert int(changed.sum()) <= 10  # only the 10 committing steps change x


def test_forward_is_deterministic_at_full_size(golden_meta):
    """The forward at the bench shape (512 x 46 rows: every kernel runs several rounds of workgroups per CU) gives
    the same bits on repeated calls with other allocations in between -- an uninitialised register or a missing wait
    shows up here long before it moves a tolerance-based comparison (tools/probe/determinism.py bisects a failure)."""
    m = _model(golden_meta["real"], False, torch.bfloat16)
    ids = synth.make_inputs(512, "ded", seed=5).to(DEV)
    x = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
    ref = m(x).logits.clone()
    for i in range(3):
        junk = torch.randn((i + 1) * 1000003, device=DEV)  # moves the caching allocator and the L2 contents
        assert torch.equal(m(x).logits, ref), f"forward differs on repeat {i}"
        del junk


def test_full_size_properties_bf16(golden_meta):
    """BASELINE config 2 at full size (512 deduction rows, 33 masks, 32 steps, bf16, T=0.1/top_p=0.9): the oracle
    cannot run this in seconds, so check the size-independent invariants of the loop (generation_utils.py:488-540):
    unmasked positions never change, the number of masks per row follows the k-schedule exactly, every committed id
    is a valid vocabulary id, and the same seed gives the same trajectory twice (deterministic kernels)."""
    m = _model(golden_meta["real"], False, torch.bfloat16)
    ids = synth.make_inputs(512, "ded", seed=21).to(DEV)
    runs = []
    for _ in range(2):
        torch.manual_seed(77)
        out = m.diffusion_generate(inputs=ids, generation_config=_gc(45), return_dict_in_generate=True,
                                   output_history=True, steps=32, temperature=0.1, alg="entropy", alg_temp=0,
                                   top_p=0.9, skip_noop_steps=False)
        runs.append(torch.stack(out.history))
    hist = runs[0]
    assert torch.equal(runs[0], runs[1]), "same seed, same inputs: trajectories must be identical"
    x0 = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
    prev = x0
    left = 33
    for step, k in enumerate(O.transfer_schedule(33, 32)):
        cur = hist[step]
        was_mask = prev == synth.MASK
        assert torch.equal(cur[~was_mask], prev[~was_mask]), f"step {step}: an unmasked position changed"
        newly = was_mask & (cur != synth.MASK)
        left -= k
        # a committed id may itself be the MASK id (a legal vocabulary entry): such a row keeps one more mask
        cnt = (cur == synth.MASK).sum(1)
        assert bool(((cnt == left) | (cnt == left + 1)).all()), f"step {step}"
        assert bool((newly.sum(1) <= k).all())
        assert bool(((cur >= 0) & (cur < golden_meta["real"]["vocab_size"])).all())
        prev = cur
    assert int((hist[-1] == synth.MASK).sum()) <= 512 * 0.01


def test_config4_shape_vs_oracle():
    """BASELINE config 4 shape (N = 128, 12 heads of 64, long sequences) on a 2-layer cut: histories against the
    oracle on the same noise, fp32."""
    kw = dict(vocab_size=2048, hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12,
              num_key_value_heads=12, max_position_embeddings=128, mask_token_id=synth.MASK, pad_token_id=synth.END)
    m = _model(kw, True)
    cfg = O.OracleConfig(**{k: v for k, v in kw.items() if k in O.OracleConfig.__dataclass_fields__})
    w = synth.make_weights(m.config, seed=0, rich=True)
    g = torch.Generator().manual_seed(4)
    B, n_in = 6, 127
    ids = torch.randint(100, 2047, (B, n_in), generator=g)
    ids[:, 0] = synth.BOS
    ids[:, 40:103] = synth.MASK  # 63 masks + the padded one = 64 per row
    noises = {}

    def noise_fn(step, M, V):
        if step not in noises:
            noises[step] = torch.empty(M, V).exponential_(1, generator=g)
        return noises[step]

    gc = DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, max_length=128)
    out = m.diffusion_generate(inputs=ids.to(DEV), generation_config=gc, return_dict_in_generate=True,
                               output_history=True, steps=8, temperature=0.5, alg="entropy", alg_temp=0, top_p=0.9,
                               _noise_fn=noise_fn)
    _, hist = O.sample_loop(w, cfg, ids, max_length=128, steps=8, mask_token_id=synth.MASK, alg="entropy", alg_temp=0,
                            temperature=0.5, top_p=0.9, noise_fn=lambda s, M, V: noises[s], tie_break="index")
    got = torch.stack(out.history).cpu()
    exp = torch.stack(hist)
    mism = (got != exp).any(-1).any(-1)
    assert not bool(mism.any()), f"history differs from the oracle at steps {mism.nonzero().view(-1).tolist()}"


def test_think_sample_matches_reference(mini, golden_meta):
    """`think_sample=True` (generation_utils.py:367-432, :490-491) against goldens generated from the reference
    (tests/golden/make_goldens_think.py).  The categorical noise of the committing draws is recorded from an oracle
    run on the reference's CPU RNG stream (which also consumes the two discarded think draws) and replayed.
    Free runs must equal the oracle; a diverging case is replayed under teacher forcing and every differing step
    must be an fp32 near-tie, either between positions (confidences) or between tokens (the oracle's own logits of
    the two tokens within 1e-5: on think steps the reference forwards a 4x repeated batch, whose CPU GEMM blocking
    differs in the last bits from the plain forward)."""
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "reference_goldens_think.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(here, "reference_goldens_think.npz"))
    cfg = O.OracleConfig(**{k: v for k, v in golden_meta["mini"].items() if k in O.OracleConfig.__dataclass_fields__})
    w = synth.make_weights(cfg, seed=0, rich=True)
    exact = 0
    for c in meta["cases"]:
        kw = dict(c["kw"])
        ids = torch.from_numpy(g[c["ids"]])
        am = torch.from_numpy(g[c["attention_mask"]]) if c["attention_mask"] else None
        noise = {}

        def rec(step, M, V):
            noise[step] = torch.empty(M, V).exponential_(1)
            return noise[step]

        torch.manual_seed(c["seed"])
        otr = O.SampleTrace()
        _, oh = O.sample_loop(w, cfg, ids, max_length=ids.shape[1] + 1, mask_token_id=synth.MASK, attention_mask=am,
                              steps=kw["steps"], alg=kw["alg"], alg_temp=kw["alg_temp"], temperature=kw["temperature"],
                              top_p=kw["top_p"], think_sample=True, think_sample_interval=kw["think_sample_interval"],
                              sep_token_id=synth.SEP, noise_fn=rec, tie_break="index", trace=otr, keep_logits=True)
        oh = torch.stack(oh)
        extra = {} if am is None else {"attention_mask": am.to(DEV)}

        def run(**priv):
            out = mini.diffusion_generate(ids.to(DEV), generation_config=_gc(ids.shape[1]), output_history=True,
                                          return_dict_in_generate=True, think_sample=True,
                                          _noise_fn=lambda s_, M, V: noise[s_], **priv, **extra, **kw)
            return torch.stack(out.history).cpu()

        got = run()
        if torch.equal(got, oh):
            exact += int(np.array_equal(got.numpy(), g[c["name"] + "_hist"]))
            continue
        forced = []
        run(_state_hook=lambda i, x: (forced.append(x.cpu().clone()), oh[i])[1])
        x_init = torch.nn.functional.pad(ids, (0, 1), value=synth.MASK)
        for i, hx in enumerate(forced):
            if torch.equal(hx, oh[i]):
                continue
            prev = oh[i - 1] if i else x_init
            if _explained_by_near_tie(prev, hx, oh[i], otr.confidence[i], otr.k[i]):
                continue
            mi = prev == synth.MASK
            flat = mi.flatten().nonzero().flatten().tolist()
            for b, j in (hx != oh[i]).nonzero().tolist():
                assert hx[b, j] != synth.MASK and oh[i][b, j] != synth.MASK, (c["name"], i, b, j)
                lg = otr.mask_logits[i][flat.index(b * mi.shape[1] + j)]
                assert abs(float(lg[hx[b, j]] - lg[oh[i][b, j]])) <= 1e-5, (c["name"], i, b, j)
    assert exact >= 2, exact                                          # == the reference's goldens end to end


def test_flat_rows_switch_the_noise_form_invisibly(golden_meta):
    """Round 3: on flat rows (T = 0.5 on random-init logits: thousands of kept tokens a row) the loop materialises the
    Exp(1) noise instead of evaluating Philox per kept token (HipState.note_entropy).  The switch happens mid-call, one
    or two steps behind the statistic; sequences, histories and the generator offset are those of either pure form."""
    cfg = DreamConfig(**dict(golden_meta["real"], num_hidden_layers=2))
    m = DreamModel_length(cfg)
    m.load_state_dict(synth.make_weights(cfg, seed=0), strict=True)
    m = m.to(DEV).to(torch.bfloat16)
    gen = torch.cuda.default_generators[0]
    ids = synth.make_inputs(64, "ded", seed=4).to(DEV)
    kw = dict(steps=12, temperature=0.5, alg="entropy", alg_temp=0, top_p=0.9, skip_noop_steps=False,
              return_dict_in_generate=True, output_history=True)
    outs, offs, flat = [], [], []
    for lazy in (True, False, True):
        for st in m._states.values():
            st.flat_rows, st.ent_event = False, None
        torch.manual_seed(21)
        o = m.diffusion_generate(ids, generation_config=_gc(45), lazy_noise=lazy, **kw)
        torch.cuda.synchronize()
        outs.append(torch.stack(o.history))
        offs.append(gen.get_offset())
        flat.append(any(st.flat_rows for st in m._states.values()))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and offs[0] == offs[1] == offs[2]
    assert flat[0] and flat[2], "the flat-row statistic never fired on rows this flat"
    # peaked rows keep the lazy form
    for st in m._states.values():
        st.flat_rows, st.ent_event = False, None
    m.diffusion_generate(ids, generation_config=_gc(45), **dict(kw, temperature=0.02))
    torch.cuda.synchronize()
    assert not any(st.flat_rows for st in m._states.values())


def test_lazy_noise_is_invisible(golden_meta):
    """`lazy_noise=True` (draws computed in the sampler from torch's Philox stream) and `lazy_noise=False` (the
    [M, V] exponential_ fill the reference makes) give the same sequences and leave the generator at the same
    offset; checked on the real vocabulary in bf16, fixed-length and variable-length loops."""
    from dark_amd import DreamModel
    from dark_amd.generation_expand_utils import DreamGenerationConfig as ExpandConfig
    kw_cfg = dict(golden_meta["real"], num_hidden_layers=2)
    cfg = DreamConfig(**kw_cfg)
    m = DreamModel(cfg)
    m.load_state_dict(synth.boost_edit_tokens(synth.make_weights(cfg, seed=0), 8.0, 7.0), strict=True)
    m = m.to(DEV).to(torch.bfloat16)
    gen = torch.cuda.default_generators[0]
    ids = synth.make_inputs(48, "ded", seed=3).to(DEV)
    kw = dict(steps=8, temperature=0.5, alg="entropy", alg_temp=0, top_p=0.9)
    outs, offs = [], []
    for lazy in (True, False, True):
        torch.manual_seed(11)
        outs.append(m.diffusion_generate_length(ids, generation_config=_gc(45), lazy_noise=lazy, **kw))
        offs.append(gen.get_offset())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and offs[0] == offs[1] == offs[2]
    prompts = [p.to(DEV) for p in synth.make_var_inputs(24, "ded", 6, seed=5)]
    gce = ExpandConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS, eos_token_id=synth.END,
                       expand_token_id=synth.EXPAND, delete_token_id=synth.END, max_length=50)
    outs, offs = [], []
    for lazy in (True, False):
        torch.manual_seed(12)
        outs.append(m.diffusion_generate_batch(prompts, generation_config=gce, temperature=0.5, top_p=0.9, alg_temp=0,
                                               lazy_noise=lazy))
        offs.append(gen.get_offset())
    assert torch.equal(outs[0], outs[1]) and offs[0] == offs[1]
