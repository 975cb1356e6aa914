"""dark_head_sample_rows (LM head + sampler without materialised logits: gemm256.hip EPI_HEAD records, sample.hip
head_finish_kernel, device-sized fallback) must give the SAME conf / x0 bits as dark_lm_head_rows + dark_sample_rows.
Rows are built to hit every branch: peaked rows, exact ties inside one 64-column slice and across slices (duplicated
lm_head rows), all-equal rows (hidden = 0: flat, certified by nothing -> fallback list), rows whose candidates exceed
64, all three confidence modes, Philox and materialised noise, the bench vocabulary and one of config 3's size (25428: the fused path needs a
multiple of 4; 25427 itself takes the two-call form, covered by the last test)."""
import pytest
import torch

from dark_amd import DreamConfig, DreamModel_length, _hip, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(V, peak=1.0):
    cfg = DreamConfig(vocab_size=V, hidden_size=768, intermediate_size=3072, num_hidden_layers=1,
                      num_attention_heads=8, mask_token_id=synth.MASK, pad_token_id=synth.END)
    sd = synth.make_weights(cfg, seed=0)
    w = sd["lm_head.weight"]
    # exact ties: equal weight rows give equal logits in every row of the batch
    w[501] = w[500]            # same 64-column slice
    w[9000] = w[500]           # another slice
    w[V - 1] = w[V - 3]        # in the last, partly filled slice
    sd["model.norm.weight"] = sd["model.norm.weight"] * peak  # > 1: logits as peaked as a trained model's
    m = DreamModel_length(cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).to(torch.bfloat16)


def _hidden(rows, seed):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(rows, 768, generator=g)
    h[5] = 0.0                 # all logits equal: as flat as a row gets
    h[7] *= 0.01               # nearly flat: hundreds of candidates in the head range
    h[11] *= 4.0               # one dominant token
    return h.to(DEV)


def _both(st, M, T, top_p, top_k, mode, noise, philox):
    out = []
    for fused in (True, False):
        st.fused_head = fused
        st.head_stats.zero_()
        st.conf.fill_(float("nan"))
        st.x0.fill_(-1)
        conf, x0 = st.head_and_sample(M, T, top_p, top_k, mode, noise, philox)
        torch.cuda.synchronize()
        out.append((conf.clone(), x0.clone(), st.head_stats.cpu().tolist()))
    st.fused_head = "auto"
    return out


@pytest.mark.parametrize("V,peak", [(15080, 1.0), (15080, 6.0), (25428, 6.0)])
def test_fused_head_equals_two_call_form(V, peak):
    """peak = 1: the flat rows of random-init weights (almost everything goes to the fallback list); peak = 6: rows as
    peaked as a trained model's (almost everything finishes from the records)."""
    m = _model(V, peak)
    st = m._hip_state(64, 46)
    M = 64 * 46 - 3            # a ragged row count >= 1024
    st.hidden_rows.copy_(_hidden(64 * 46, 1))
    st.rows_valid = True
    seen_fallback, seen_fast = False, False
    for T, top_p, mode in ((0.1, 0.9, _hip.CONF_NEG_ENTROPY), (0.1, 0.9, _hip.CONF_PROB), (0.5, 0.95, _hip.CONF_MARGIN),
                           (1.0, 0.9, _hip.CONF_NEG_ENTROPY), (0.05, 0.0, _hip.CONF_PROB), (0.2, 0.5, _hip.CONF_NEG_ENTROPY)):
        philox = (1234 + int(T * 100), 8, 1 << 20)
        (c1, x1, s1), (c2, x2, s2) = _both(st, M, T, top_p, 0, mode, None, philox)
        assert s2 == [0, 0], "the two-call form keeps no statistics"
        assert s1[0] == M and 0 <= s1[1] <= M
        assert torch.equal(x1, x2), f"T={T} top_p={top_p} mode={mode}: {int((x1 != x2).sum())} tokens differ"
        assert torch.equal(c1.view(torch.int32), c2.view(torch.int32)), f"T={T} top_p={top_p} mode={mode}: confidences differ"
        assert s1[1] >= 1, "the all-equal row cannot be finished from the records"
        seen_fallback |= s1[1] > M // 2
        seen_fast |= s1[1] < M // 10
        print(f"V={V} peak={peak} T={T} top_p={top_p} mode={mode}: {s1[1]} of {M} rows needed their logits")
    assert seen_fast, "peaked rows must finish from the records"
    assert seen_fallback, "flat rows must take the fallback"
    # materialised noise instead of the Philox stream
    noise = torch.empty(M, V, device=DEV).exponential_(1.0)
    (c1, x1, _), (c2, x2, _) = _both(st, M, 0.1, 0.9, 0, _hip.CONF_NEG_ENTROPY, noise, None)
    assert torch.equal(x1, x2) and torch.equal(c1.view(torch.int32), c2.view(torch.int32))


def test_cases_outside_the_fused_path_run_the_two_call_form():
    """temperature 0, top-k, no nucleus filter, few rows: same entry point, same results, no records involved."""
    m = _model(15080)
    st = m._hip_state(64, 46)
    st.hidden_rows.copy_(_hidden(64 * 46, 2))
    st.rows_valid = True
    V = 15080
    for M, T, top_p, top_k in ((2000, 0.0, 0.9, 0), (2000, 0.3, -1.0, 0), (2000, 0.3, 0.9, 20), (700, 0.1, 0.9, 0)):  # noqa
        philox = (77, 0, 1 << 20) if T > 0 else None
        noise = None
        (c1, x1, s1), (c2, x2, _) = _both(st, M, T, top_p, top_k, _hip.CONF_NEG_ENTROPY, noise, philox)
        assert s1 == [0, 0]
        assert torch.equal(x1, x2) and torch.equal(c1.view(torch.int32), c2.view(torch.int32))
