"""`dark_lm_head_rows_noalloc` (the one entry point that allocates nothing, include/dark_hip.h) through the C ABI
against `dark_lm_head_rows`: bit-identical logits on two calls in a row, the second with fewer rows; sizes beyond the
static scratch buffer are refused.  Collected last (file name) so that a failure here hides no other test."""
import pytest
import torch

from dark_amd import DreamConfig, DreamModel_length, _hip, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(vocab_size=15080, hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=8,
          num_key_value_heads=8, max_position_embeddings=64, mask_token_id=synth.MASK, pad_token_id=synth.END)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_lm_head_rows_noalloc_matches_workspace_variant(dtype):
    lib = _hip.load()
    cfg = DreamConfig(**KW)
    m = DreamModel_length(cfg)
    m.load_state_dict(synth.make_weights(cfg, seed=0), strict=True)
    m = m.to(DEV).to(dtype)
    pk = m.packed()
    B, N, V, d = 64, 46, cfg.vocab_size, cfg.hidden_size
    x = torch.nn.functional.pad(synth.make_inputs(B, "ded", seed=4), (0, 1), value=synth.MASK).to(DEV).contiguous()
    st = m._hip_state(B, N)
    st.forward_hidden(x, None, None)
    st.mask_rows(x, synth.MASK)
    M_all = st.wait_count()
    assert M_all == B * 33
    stream = torch.cuda.current_stream().cuda_stream
    max_rows = lib.dark_noalloc_max_rows(pk.ref)
    assert max_rows == (8 << 20) // (d * pk.elem)
    _hip.check(lib.dark_noalloc_init())
    for M in (M_all, 517):  # two calls in a row, the second with a smaller input
        assert M <= max_rows
        ws_bytes = lib.dark_head_workspace_bytes(pk.ref, M)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        ref = torch.full((M, V), float("nan"), device=DEV)
        got = torch.full((M, V), float("nan"), device=DEV)
        _hip.check(lib.dark_lm_head_rows(pk.ref, _hip.ptr(st.hidden), _hip.ptr(st.src_rows), M, _hip.ptr(ws), ws_bytes,
                                         _hip.ptr(ref), V, stream))
        _hip.check(lib.dark_lm_head_rows_noalloc(pk.ref, _hip.ptr(st.hidden), _hip.ptr(st.src_rows), M, _hip.ptr(got),
                                                 V, stream))
        torch.cuda.synchronize()
        assert not bool(torch.isnan(got).any())
        assert torch.equal(got, ref), f"M={M}: {(got != ref).sum().item()} logits differ"
    with pytest.raises(_hip.DarkHipError, match="scratch"):
        big = torch.empty(1, device=DEV)
        _hip.check(lib.dark_lm_head_rows_noalloc(pk.ref, _hip.ptr(st.hidden), None, int(max_rows) + 1, _hip.ptr(big), V,
                                                 stream))
