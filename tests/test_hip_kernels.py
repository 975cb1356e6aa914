"""GPU parity tests, kernel by kernel, through the C ABI (libdark_hip.so) against the CPU oracle.

Bars: integer outputs (token ids, row indices, committed sequences) bit-exact; fp32 kernels within
1e-4 relative of an fp64 reference (north-star: logits within 1e-3); bf16 kernels within 2e-2 of the
fp32 oracle (stated looser bound: bf16 has 8 mantissa bits, inputs are rounded before the MFMA).
"""
import numpy as np
import pytest
import torch

from oracle import dream_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def lib():
    from dark_amd import _hip
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return _hip.load()


def _hip():
    from dark_amd import _hip
    return _hip


def _s():
    return _hip().stream_ptr(torch.device(DEV))


def _pad_rows(w, mult=128):
    r = (-w.shape[0]) % mult
    return w if r == 0 else torch.cat([w, w.new_zeros((r,) + tuple(w.shape[1:]))], 0)


def _tol(dtype):
    return dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)


def _dt(dtype):
    return _hip().DARK_BF16 if dtype == torch.bfloat16 else _hip().DARK_F32


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(300, 576, 192), (2944, 2304, 768), (129, 768, 3072), (64, 15080, 768)])
def test_gemm_plain_and_bias(lib, dtype, M, N, K):
    H = _hip()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t()
    Ad, Wd = A.to(DEV), _pad_rows(W).to(DEV)
    # epi 3: plain fp32 store with an ldo wider than N and a guard column
    ldo = N + 5
    out = torch.full((M, ldo), 7.0, device=DEV)
    H.check(lib.dark_dbg_gemm(_dt(dtype), 3, H.ptr(Ad), H.ptr(Wd), None, H.ptr(out), M, N, K, ldo, _s()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[:, :N].cpu().double().numpy(), ref.numpy(), **_tol(dtype))
    assert bool((out[:, N:] == 7.0).all()), "columns beyond N must stay untouched"
    # epi 0: + bias, output in the model dtype
    out2 = torch.zeros((M, N), device=DEV, dtype=dtype)
    bd = _pad_rows(bias).to(DEV)
    H.check(lib.dark_dbg_gemm(_dt(dtype), 0, H.ptr(Ad), H.ptr(Wd), H.ptr(bd), H.ptr(out2), M, N, K, N, _s()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out2.cpu().double().numpy(), (ref + bias.double()).numpy(), **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_resid_and_swiglu(lib, dtype):
    H = _hip()
    from dark_amd.modeling_dream_length import gate_up_group, interleave_gate_up
    g = torch.Generator().manual_seed(5)
    M, d, ff = 333, 192, 384
    A = torch.randn(M, ff, generator=g).to(dtype)
    Wd_ = (torch.randn(d, ff, generator=g) * 0.05).to(dtype)
    res = torch.randn(M, d, generator=g)
    x = res.clone().to(DEV)
    Ad, Wdd = A.to(DEV), _pad_rows(Wd_).to(DEV)  # keep device tensors alive across the async launch
    H.check(lib.dark_dbg_gemm(_dt(dtype), 1, H.ptr(Ad), H.ptr(Wdd), None, H.ptr(x), M, d, ff, d, _s()))
    torch.cuda.synchronize()
    ref = res.double() + A.double() @ Wd_.double().t()
    np.testing.assert_allclose(x.cpu().double().numpy(), ref.numpy(), **_tol(dtype))
    # SwiGLU over interleaved gate/up
    X = torch.randn(M, d, generator=g).to(dtype)
    Wg = (torch.randn(ff, d, generator=g) * 0.1).to(dtype)
    Wu = (torch.randn(ff, d, generator=g) * 0.1).to(dtype)
    Wgu = _pad_rows(interleave_gate_up(Wg, Wu, gate_up_group(dtype))).to(DEV)
    act = torch.zeros(M, ff, device=DEV, dtype=dtype)
    Xd = X.to(DEV)
    H.check(lib.dark_dbg_gemm(_dt(dtype), 2, H.ptr(Xd), H.ptr(Wgu), None, H.ptr(act), M, 2 * ff, d, ff, _s()))
    torch.cuda.synchronize()
    gg, uu = X.double() @ Wg.double().t(), X.double() @ Wu.double().t()
    ref = torch.nn.functional.silu(gg) * uu
    np.testing.assert_allclose(act.cpu().double().numpy(), ref.numpy(), **_tol(dtype))


@pytest.mark.parametrize("epi,M,N,K", [(8, 24001, 1536, 192), (8, 777, 768, 3072), (8, 23552, 768, 256),
                                       (0, 24001, 2304, 128), (2, 5000, 6144, 768), (3, 2049, 15080, 768),
                                       (8, 1, 768, 128), (0, 288, 512, 128), (2, 289, 512, 128), (3, 321, 264, 128)])
def test_gemm_persistent_tile_heights_bit_identical(lib, epi, M, N, K):
    """gemm_bf16_s256 (gemm256.hip) with its tile height forced to 256 / 288 / 320 rows (variants 8 / 9 / 10: different
    DMA piece splits and counted waits per height) against the 128-row kernel (variant 1) on the same operands:
    every flavour the forward routes there -- bias, SwiGLU, fp32 head and the bf16-residual read-modify-write --
    bit for bit, ragged last tiles, one round and several tiles per CU; rows beyond M untouched.  On the taller tiles
    the bias / SwiGLU / residual flavours run the interleaved LOAD interval (round 3); variant 14 is the library's own
    choice of height with the sequential interval, variant 0 the default: both are held to the same bits."""
    H = _hip()
    from dark_amd.modeling_dream_length import interleave_gate_up
    g = torch.Generator().manual_seed(epi * 131 + M + N + K)
    A = (torch.randn(M, K, generator=g)).bfloat16().to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    if epi == 2:
        W = interleave_gate_up(W[: N // 2], W[N // 2:], 16)
    W = _pad_rows(W).to(DEV)
    bias = _pad_rows(torch.randn(N, generator=g)).to(DEV)
    ldo = N // 2 if epi == 2 else N
    odt = torch.float32 if epi == 3 else torch.bfloat16
    base = torch.randn(M + 2, ldo, generator=g).to(odt).to(DEV)
    outs = {}
    try:
        for v in (1, 8, 9, 10, 14, 0):
            H.check(lib.dark_dbg_set_gemm_variant(v))
            out = base.clone()
            H.check(lib.dark_dbg_gemm(H.DARK_BF16, epi, H.ptr(A), H.ptr(W), H.ptr(bias) if epi == 0 else None, H.ptr(out),
                                      M, N, K, ldo, _s()))
            torch.cuda.synchronize()
            outs[v] = out
    finally:
        H.check(lib.dark_dbg_set_gemm_variant(0))
    assert not torch.equal(outs[1][:M], base[:M])
    for v in (8, 9, 10, 14, 0):
        assert torch.equal(outs[v][:M].view(torch.int16 if odt == torch.bfloat16 else torch.int32),
                           outs[1][:M].view(torch.int16 if odt == torch.bfloat16 else torch.int32)), (epi, v)
        assert torch.equal(outs[v][M:], base[M:]), "rows beyond M must stay untouched"


@pytest.mark.parametrize("M,K", [(1, 192), (95, 192), (96, 384), (333, 768), (2049, 3072)])
def test_gemm_rowpanel_resid_norm(lib, M, K):
    """gemm_rowpanel.hip (variant 5; o_proj / down_proj + residual + RMSNorm in one launch): the fp32 residual is
    bit-identical to the residual GEMM's, the normalised rows are the norm of that residual (fp64 reference within
    bf16 rounding) and within 1 bf16 ulp of the separate rmsnorm kernel's; rows beyond M are never written."""
    H = _hip()
    d, eps = 768, 1e-6
    g = torch.Generator().manual_seed(M * 7 + K)
    A = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    W = (torch.randn(d, K, generator=g) * 0.05).bfloat16().to(DEV)
    h0 = torch.randn(M + 2, d, generator=g).to(DEV)
    gam = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    h1, h2 = h0.clone(), h0.clone()
    x1 = torch.full((M + 2, d), 3.0, device=DEV, dtype=torch.bfloat16)
    x2 = torch.full((M, d), 3.0, device=DEV, dtype=torch.bfloat16)
    H.check(lib.dark_dbg_gemm_resid_norm(H.ptr(A), H.ptr(W), H.ptr(h1), H.ptr(gam), H.ptr(x1), M, K, eps, _s()))
    H.check(lib.dark_dbg_gemm(H.DARK_BF16, 1, H.ptr(A), H.ptr(W), None, H.ptr(h2), M, d, K, d, _s()))
    H.check(lib.dark_dbg_rmsnorm(H.DARK_BF16, H.DARK_F32, H.ptr(h2), H.ptr(gam), H.ptr(x2), None, M, M, d, eps, _s()))
    h3 = h0.clone()
    H.check(lib.dark_dbg_gemm_resid_norm(H.ptr(A), H.ptr(W), H.ptr(h3), None, None, M, K, eps, _s()))
    torch.cuda.synchronize()
    assert torch.equal(h1, h2) and torch.equal(h3, h2)
    assert torch.equal(h1[M:], h0[M:]) and bool((x1[M:] == 3.0).all()), "rows beyond M must stay untouched"
    ref = O.rms_norm(h1[:M].cpu(), gam.cpu(), eps)
    np.testing.assert_allclose(x1[:M].cpu().float().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)
    ulp = (x1[:M].view(torch.int16).int() - x2.view(torch.int16).int()).abs().max().item()
    assert ulp <= 1
    # shapes the kernel does not take are refused, not mis-run
    assert lib.dark_dbg_gemm_resid_norm(H.ptr(A), H.ptr(W), H.ptr(h3), None, None, M, 64, eps, _s()) == H.DARK_ERR_ARG
    assert lib.dark_dbg_gemm_resid_norm(H.ptr(A), H.ptr(W), H.ptr(h3), H.ptr(gam), None, M, K, eps, _s()) == H.DARK_ERR_ARG


@pytest.mark.parametrize("M,K", [(1, 192), (95, 192), (96, 384), (333, 768), (2049, 3072), (23552, 768)])
def test_gemm_rowpanel_bf16_stream_is_bit_identical(lib, M, K):
    """Round 3: the row-panel kernel on a bf16 residual stream gives the SAME BITS as the residual GEMM (epi 8) followed
    by rmsnorm_h16_kernel -- hidden and xn -- because it rebuilds that kernel's sum of squares in its order; rows beyond
    M are never written."""
    H = _hip()
    d, eps = 768, 1e-6
    g = torch.Generator().manual_seed(M * 5 + K)
    A = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    W = (torch.randn(d, K, generator=g) * 0.05).bfloat16().to(DEV)
    h0 = torch.randn(M + 2, d, generator=g).bfloat16().to(DEV)
    gam = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    h1, h2, h3 = h0.clone(), h0.clone(), h0.clone()
    x1 = torch.full((M + 2, d), 3.0, device=DEV, dtype=torch.bfloat16)
    x2 = torch.full((M, d), 3.0, device=DEV, dtype=torch.bfloat16)
    H.check(lib.dark_dbg_gemm_resid_norm_h(H.ptr(A), H.ptr(W), H.ptr(h1), H.ptr(gam), H.ptr(x1), M, K, eps, _s()))
    H.check(lib.dark_dbg_set_gemm_variant(1))   # the 128-row residual GEMM (the persistent one gives the same bits)
    H.check(lib.dark_dbg_gemm(H.DARK_BF16, 8, H.ptr(A), H.ptr(W), None, H.ptr(h2), M, d, K, d, _s()))
    H.check(lib.dark_dbg_set_gemm_variant(0))
    H.check(lib.dark_dbg_rmsnorm(H.DARK_BF16, H.DARK_BF16, H.ptr(h2), H.ptr(gam), H.ptr(x2), None, M, M, d, eps, _s()))
    H.check(lib.dark_dbg_gemm_resid_norm_h(H.ptr(A), H.ptr(W), H.ptr(h3), None, None, M, K, eps, _s()))
    torch.cuda.synchronize()
    assert torch.equal(h1.view(torch.int16), h2.view(torch.int16)) and torch.equal(h3.view(torch.int16), h2.view(torch.int16))
    assert torch.equal(h1[M:], h0[M:]) and bool((x1[M:] == 3.0).all()), "rows beyond M must stay untouched"
    assert torch.equal(x1[:M].view(torch.int16), x2.view(torch.int16)), "xn must equal the separate norm kernel's bit for bit"
    ref = O.rms_norm(h1[:M].float().cpu(), gam.cpu(), eps)
    np.testing.assert_allclose(x1[:M].cpu().float().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)


# --------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_and_rmsnorm(lib, dtype):
    H = _hip()
    g = torch.Generator().manual_seed(3)
    V, d, M = 512, 192, 77
    table = torch.randn(V, d, generator=g).to(dtype)
    ids = torch.randint(0, V, (M,), generator=g)
    out = torch.zeros(M, d, device=DEV)
    idd, tabd = ids.to(DEV), table.to(DEV)
    H.check(lib.dark_dbg_embed(_dt(dtype), H.DARK_F32, H.ptr(idd), H.ptr(tabd), H.ptr(out), M, d, V, _s()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), table.float()[ids])
    x = torch.randn(M, d, generator=g) * 3
    w = 1 + 0.1 * torch.randn(d, generator=g)
    src = torch.randint(0, M, (40,), generator=g).int()
    y = torch.zeros(40, d, device=DEV, dtype=dtype)
    xd, wd, srcd = x.to(DEV), w.to(DEV), src.to(DEV)
    H.check(lib.dark_dbg_rmsnorm(_dt(dtype), H.DARK_F32, H.ptr(xd), H.ptr(wd), H.ptr(y), H.ptr(srcd), M, 40, d, 1e-6, _s()))
    torch.cuda.synchronize()
    ref = O.rms_norm(x[src.long()], w, 1e-6)
    np.testing.assert_allclose(y.cpu().float().numpy(), ref.numpy(), **(_tol(dtype) if dtype == torch.bfloat16
                                                                         else dict(rtol=2e-6, atol=1e-6)))
    # a bf16 residual stream exists for bf16 models only
    assert lib.dark_dbg_rmsnorm(H.DARK_F32, H.DARK_BF16, H.ptr(xd), H.ptr(wd), H.ptr(y), None, M, 40, d, 1e-6, _s()) \
        == H.DARK_ERR_ARG


def test_bf16_residual_stream_kernels(lib):
    """The elementwise kernels on a bf16 residual stream (dark_model_desc.resid_dtype = DARK_BF16): the embedding copies
    the bf16 table rows exactly; RMSNorm of a bf16 stream against the same values held in fp32: within one bf16 ulp
    (the 16-byte kernel of the bf16 stream reduces the sum of squares in another order) and against the oracle; a
    width that is not a multiple of 8 takes the generic kernel, bit-identical to the fp32-stream one."""
    H = _hip()
    g = torch.Generator().manual_seed(5)
    V, d, M = 512, 768, 203
    table = torch.randn(V, d, generator=g).bfloat16()
    ids = torch.randint(0, V, (M,), generator=g)
    out = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
    idd, tabd = ids.to(DEV), table.to(DEV)
    H.check(lib.dark_dbg_embed(H.DARK_BF16, H.DARK_BF16, H.ptr(idd), H.ptr(tabd), H.ptr(out), M, d, V, _s()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), table[ids])
    x = (torch.randn(M, d, generator=g) * 3).bfloat16().to(DEV)
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    src = torch.randint(0, M, (90,), generator=g).int().to(DEV)
    y_h = torch.zeros(90, d, device=DEV, dtype=torch.bfloat16)
    y_f = torch.zeros(90, d, device=DEV, dtype=torch.bfloat16)
    xf = x.float()
    H.check(lib.dark_dbg_rmsnorm(H.DARK_BF16, H.DARK_BF16, H.ptr(x), H.ptr(w), H.ptr(y_h), H.ptr(src), M, 90, d, 1e-6, _s()))
    H.check(lib.dark_dbg_rmsnorm(H.DARK_BF16, H.DARK_F32, H.ptr(xf), H.ptr(w), H.ptr(y_f), H.ptr(src), M, 90, d, 1e-6, _s()))
    torch.cuda.synchronize()
    ulp = (y_h.view(torch.int16).int() - y_f.view(torch.int16).int()).abs().max().item()
    assert ulp <= 1, ulp
    ref = O.rms_norm(xf.cpu()[src.cpu().long()], w.cpu(), 1e-6)
    np.testing.assert_allclose(y_h.cpu().float().numpy(), ref.numpy(), **_tol(torch.bfloat16))
    d2 = 196   # (d % 8 != 0: generic kernel)
    x2 = (torch.randn(M, d2, generator=g) * 3).bfloat16().to(DEV)
    w2 = (1 + 0.1 * torch.randn(d2, generator=g)).to(DEV)
    a = torch.zeros(M, d2, device=DEV, dtype=torch.bfloat16)
    b = torch.zeros(M, d2, device=DEV, dtype=torch.bfloat16)
    x2f = x2.float()
    H.check(lib.dark_dbg_rmsnorm(H.DARK_BF16, H.DARK_BF16, H.ptr(x2), H.ptr(w2), H.ptr(a), None, M, M, d2, 1e-6, _s()))
    H.check(lib.dark_dbg_rmsnorm(H.DARK_BF16, H.DARK_F32, H.ptr(x2f), H.ptr(w2), H.ptr(b), None, M, M, d2, 1e-6, _s()))
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (777, 768, 3072), (130, 384, 128), (45, 200, 64), (2049, 768, 192),
                                   (77, 196, 128)])
def test_gemm_bf16_residual_epilogue(lib, M, N, K):
    """EPI_RESID_H (epi 8): out (bf16) = bf16(float(out) + A W^T), ONE rounding of the fp32 sum -- compared with the
    fp32-stream flavour (epi 1) started from the same values: rounding its fp32 result gives the bf16 result bit for
    bit (same accumulators, same add).  Rows beyond M stay untouched; unaligned N takes the scalar tail."""
    H = _hip()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    W = torch.zeros((N + 127) // 128 * 128, K, dtype=torch.bfloat16)
    W[:N] = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    W = W.to(DEV)
    h0 = torch.randn(M + 3, N, generator=g).bfloat16().to(DEV)
    hb, hf = h0.clone(), h0.float()
    H.check(lib.dark_dbg_gemm(H.DARK_BF16, 8, H.ptr(A), H.ptr(W), None, H.ptr(hb), M, N, K, N, _s()))
    H.check(lib.dark_dbg_gemm(H.DARK_BF16, 1, H.ptr(A), H.ptr(W), None, H.ptr(hf), M, N, K, N, _s()))
    torch.cuda.synchronize()
    assert torch.equal(hb[:M], hf[:M].bfloat16())
    assert torch.equal(hb[M:], h0[M:]), "rows beyond M must stay untouched"
    assert lib.dark_dbg_gemm(H.DARK_F32, 8, H.ptr(hf), H.ptr(hf), None, H.ptr(hf), M, N, K, N, _s()) == H.DARK_ERR_UNSUPPORTED


# --------------------------------------------------------------------------- attention
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H_,HD,masked", [(46, 2, 96, False), (46, 8, 96, True), (128, 3, 64, True),
                                             (17, 2, 64, False), (100, 2, 96, True)])
def test_attention(lib, dtype, N, H_, HD, masked):
    Hh = _hip()
    from dark_amd.modeling_dream_length import rope_tables
    g = torch.Generator().manual_seed(N * 7 + HD)
    B, d = (300 if (N, H_) == (46, 8) else 3), H_ * HD  # 2400 workgroups: several rounds per CU, not just the first
    qkv = torch.randn(B, N, 3 * d, generator=g).to(dtype)
    cos, sin = rope_tables(HD, 1000.0, 128)
    key_mask = pos = None
    am4 = None
    pos_ids = torch.arange(N).unsqueeze(0).expand(B, -1)
    if masked:
        am = torch.ones(B, N, dtype=torch.long)
        am[0, N // 3: N // 2] = 0
        am[2, 1:4] = 0
        pos_ids = am.cumsum(-1) - 1
        pos_ids.masked_fill_(am == 0, 1)
        am4 = torch.logical_and(am.unsqueeze(1).unsqueeze(-2), am.unsqueeze(1).unsqueeze(-1))
        key_mask = am.to(torch.uint8).to(DEV)
        pos = pos_ids.int().contiguous().to(DEV)
    out = torch.zeros(B * N, d, device=DEV, dtype=dtype)
    qd, cd, sd = qkv.to(DEV), cos.to(DEV), sin.to(DEV)
    Hh.check(lib.dark_dbg_attention(_dt(dtype), Hh.ptr(qd), Hh.ptr(cd), Hh.ptr(sd),
                                    Hh.ptr(pos), Hh.ptr(key_mask), Hh.ptr(out), B, N, H_, HD, 128, _s()))
    torch.cuda.synchronize()
    # oracle: same ops as modeling_dream_length.py:385-431 on the fp32 image of the inputs
    q, k, v = [t.float().view(B, N, H_, HD).transpose(1, 2) for t in qkv.float().split(d, dim=-1)]
    c, s = O.rope_cos_sin(pos_ids, HD, 1000.0)
    q, k = O.apply_rope(q, k, c, s)
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=am4, is_causal=False)
    ref = ref.transpose(1, 2).reshape(B * N, d)
    ref = torch.nan_to_num(ref, nan=0.0)  # fully-masked query rows: 0
    np.testing.assert_allclose(out.cpu().float().numpy(), ref.numpy(), **(_tol(dtype) if dtype == torch.bfloat16
                                                                           else dict(rtol=1e-4, atol=1e-5)))


# --------------------------------------------------------------------------- sampling
def _sample_hip(lib, logits, T, top_p, top_k, mode, noise):
    H = _hip()
    M, V = logits.shape
    conf = torch.zeros(M, device=DEV)
    x0 = torch.zeros(M, dtype=torch.long, device=DEV)
    ld = logits.to(DEV).contiguous()
    nd = noise.to(DEV).contiguous() if noise is not None else None
    H.check(lib.dark_sample_rows(H.ptr(ld), V, M, V, float(T), float(-1.0 if top_p is None else top_p), int(top_k or 0), mode, H.ptr(nd),
                                 V if nd is not None else 0, H.ptr(conf), H.ptr(x0), _s()))
    torch.cuda.synchronize()
    return conf.cpu(), x0.cpu()


def _mode(margin, negent):
    H = _hip()
    return H.CONF_NEG_ENTROPY if negent else (H.CONF_MARGIN if margin else H.CONF_PROB)


def test_sample_rows_golden_cases(lib, goldens, golden_meta):
    """The reference's own sample_tokens outputs (tests/golden), noise replayed from the recorded seed."""
    logits = torch.from_numpy(goldens["st_logits"])
    for ci, c in enumerate(golden_meta["cases"]["sample_tokens"]):
        torch.manual_seed(c["seed"])
        noise = torch.empty_like(logits).exponential_(1) if c["T"] > 0 else None
        conf, x0 = _sample_hip(lib, logits, c["T"], c["top_p"], c["top_k"], _mode(c["margin"], c["neg_entropy"]), noise)
        # rows 5 (all equal) and 7 (200 exact duplicates): torch.sort's order among equal logits is
        # unspecified, so which duplicates survive the nucleus filter is not pinned by the reference
        keep = [r for r in range(logits.shape[0]) if r not in (5, 7)]
        assert np.array_equal(x0.numpy()[keep], goldens[f"st{ci}_x0"][keep]), f"case {ci}"
        np.testing.assert_allclose(conf.numpy()[keep], goldens[f"st{ci}_conf"][keep], rtol=2e-5, atol=1e-6,
                                   err_msg=f"case {ci}")


@pytest.mark.parametrize("T,top_p,top_k,margin,negent", [
    (0.0, None, None, False, False), (0.1, 0.9, None, False, True), (0.5, 0.9, None, False, True),
    (1.0, 0.95, None, True, False), (0.7, None, 50, False, True), (0.3, 0.8, 7, False, False),
    (0.0, 0.9, None, False, True), (2.0, 0.999, None, False, True)])
def test_sample_rows_large_vocab(lib, T, top_p, top_k, margin, negent):
    g = torch.Generator().manual_seed(11)
    M, V = 96, 15080
    logits = torch.randn(M, V, generator=g) * 0.55
    logits[::7] *= 4
    noise = torch.empty(M, V).exponential_(1, generator=g) if T > 0 else None
    rc, rx = O.sample_tokens(logits, temperature=T, top_p=top_p, top_k=top_k, margin_confidence=margin,
                             neg_entropy=negent, noise=noise)
    conf, x0 = _sample_hip(lib, logits, T, top_p, top_k, _mode(margin, negent), noise)
    assert torch.equal(x0, rx)
    np.testing.assert_allclose(conf.numpy(), rc.numpy(), rtol=3e-5, atol=1e-6)


def test_sample_rows_dbpedia_vocab(lib):
    g = torch.Generator().manual_seed(12)
    M, V = 40, 25427
    logits = torch.randn(M, V, generator=g)
    noise = torch.empty(M, V).exponential_(1, generator=g)
    rc, rx = O.sample_tokens(logits, temperature=0.5, top_p=0.9, neg_entropy=True, noise=noise)
    conf, x0 = _sample_hip(lib, logits, 0.5, 0.9, None, _hip().CONF_NEG_ENTROPY, noise)
    assert torch.equal(x0, rx)
    np.testing.assert_allclose(conf.numpy(), rc.numpy(), rtol=3e-5, atol=1e-6)


WN18RR_V = 100 + 40943 + 22 + 1   # akgr/tokenizer.py:19-27 with WN18RR's counts: 41 066 > DARK_SAMPLE_LDS_MAX_V


@pytest.mark.parametrize("T,top_p,top_k,margin,negent", [
    (0.0, None, None, False, False), (0.1, 0.9, None, False, True), (0.5, 0.9, None, False, True),
    (1.0, 0.95, None, True, False), (0.7, None, 50, False, True), (0.3, 0.8, 7, False, False)])
def test_sample_rows_wn18rr_vocab(lib, T, top_p, top_k, margin, negent):
    """The reference's third dataset: a 164 KB fp32 row does not fit in LDS, the sampler works on the registered row
    workspace (sample_bigv_kernel).  Against the oracle, as for the other vocabularies."""
    g = torch.Generator().manual_seed(31)
    M, V = 48, WN18RR_V
    logits = torch.randn(M, V, generator=g) * 0.55
    logits[::5] *= 4
    logits[3, 100:140] = logits[3].max() + 0.25       # a run of equal logits across the nucleus boundary
    noise = torch.empty(M, V).exponential_(1, generator=g) if T > 0 else None
    conf, x0 = _sample_hip(lib, logits, T, top_p, top_k, _mode(margin, negent), noise)
    # (1) against the nucleus rule evaluated exactly (oracle.exact_nucleus: prefix sums in float64): every row
    with O.exact_nucleus():
        rc, rx = O.sample_tokens(logits, temperature=T, top_p=top_p, top_k=top_k, margin_confidence=margin,
                                 neg_entropy=negent, noise=noise, stable_ties=True)
    assert torch.equal(x0, rx)
    np.testing.assert_allclose(conf.numpy(), rc.numpy(), rtol=3e-5, atol=1e-6)
    # (2) against the reference's fp32 arithmetic: every row except those whose exact cumulative mass passes within 1e-5
    # of top_p at a token that matters (p > 1e-4) -- there the REFERENCE's kept set is decided by the rounding of its
    # fp32 softmax normaliser over 41 066 terms (row 16 at T = 0.1: the first 39 tokens hold 0.8999976 of the mass,
    # torch's CPU fp32 softmax makes it 0.9000022 and keeps 39; the exact answer, and the kernel's, is 40).
    rc, rx = O.sample_tokens(logits, temperature=T, top_p=top_p, top_k=top_k, margin_confidence=margin,
                             neg_entropy=negent, noise=noise, stable_ties=True)
    ok = torch.ones(M, dtype=torch.bool)
    if top_p is not None and T > 0:
        ps = torch.sort(torch.softmax(logits.double() / T, -1), descending=True).values
        gap, at = (torch.cumsum(ps, -1) - top_p).abs().min(-1)
        ok = ~((gap < 1e-5) & (ps.gather(1, at[:, None])[:, 0] > 1e-4))
        assert int((~ok).sum()) <= 2
    assert torch.equal(x0[ok], rx[ok])
    np.testing.assert_allclose(conf.numpy()[ok], rc.numpy()[ok], rtol=3e-5, atol=1e-6)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("T,top_p,top_k", [(0.1, 0.9, None), (0.5, 0.9, None), (1.0, None, 40), (0.0, 0.9, None)])
def test_sample_bigv_equals_lds_kernel_bit_for_bit(lib, mode, T, top_p, top_k):
    """Same program, other row storage: a 30 000-wide row padded to 41 066 columns with logits that carry no mass
    (-1e30) must give the LDS-resident kernel's conf / x0 on the unpadded row, bit for bit."""
    g = torch.Generator().manual_seed(32)
    M, Vs, V = 64, 30000, WN18RR_V
    small = torch.randn(M, Vs, generator=g) * 0.6
    small[::4] *= 5
    small[1, 7:19] = small[1].max() + 0.5
    big = torch.full((M, V), -1e30)
    big[:, :Vs] = small
    ns = torch.empty(M, Vs).exponential_(1, generator=g) if T > 0 else None
    nb = None
    if ns is not None:
        nb = torch.ones(M, V)
        nb[:, :Vs] = ns
    c0, x0 = _sample_hip(lib, small, T, top_p, top_k, mode, ns)
    c1, x1 = _sample_hip(lib, big, T, top_p, top_k, mode, nb)
    assert torch.equal(x0, x1)
    assert torch.equal(c0, c1)


def test_sample_bigv_needs_its_workspace(lib):
    """No entry point allocates: without a registered row workspace a V > 32768 call is refused with the size it needs
    (the ctypes binding registers one by itself; this calls underneath it)."""
    H = _hip()
    raw = lib.dark_sample_rows.__wrapped__
    M, V = 4, WN18RR_V
    logits = torch.randn(M, V, device=DEV)
    conf = torch.zeros(M, device=DEV)
    x0 = torch.zeros(M, dtype=torch.long, device=DEV)
    H.check(lib.dark_sampler_set_row_workspace(None, 0))
    H._row_ws.clear()
    rc = raw(H.ptr(logits), V, M, V, 0.0, -1.0, 0, 0, None, 0, H.ptr(conf), H.ptr(x0), _s())
    assert rc == H.DARK_ERR_ARG and b"dark_sampler_set_row_workspace" in lib.dark_hip_last_error()
    small = torch.empty(1024, dtype=torch.uint8, device=DEV)
    H.check(lib.dark_sampler_set_row_workspace(H.ptr(small), small.numel()))
    rc = raw(H.ptr(logits), V, M, V, 0.0, -1.0, 0, 0, None, 0, H.ptr(conf), H.ptr(x0), _s())
    assert rc == H.DARK_ERR_ARG
    assert lib.dark_sampler_set_row_workspace(H.ptr(small), 0) == H.DARK_ERR_ARG
    H.check(lib.dark_sampler_set_row_workspace(None, 0))
    H.check(lib.dark_sample_rows(H.ptr(logits), V, M, V, 0.0, -1.0, 0, 0, None, 0, H.ptr(conf), H.ptr(x0), _s()))
    torch.cuda.synchronize()
    assert torch.equal(x0.cpu(), logits.cpu().argmax(-1))


@pytest.mark.parametrize("V,ld_pad", [(15080, 0), (15082, 2), (25427, 1), (16384, 0), (26624, 0), (1000, 0)])
def test_two_stage_sampler_equals_single_stage_bit_for_bit(lib, V, ld_pad):
    """sample_head_kernel (row in registers) + sample_kernel on the rows it marks, against sample_kernel alone: conf
    and x0 bit for bit.  Rows of every kind in one launch: peaked (<= 64 candidates: wave tail), 65..256 candidates
    (block list mode of the first stage), more (second stage), flat (early exit), runs of equal logits across the
    nucleus boundary, all-equal rows, a NaN, -inf entries; every confidence mode; injected and Philox noise; T = 0;
    top_p = 0; vocabularies with V % 4 != 0 on a padded stride (the tail elements) and the 26-float4 instance."""
    H = _hip()
    g = torch.Generator().manual_seed(41 + V)
    M = 300
    ld = V + ld_pad
    assert ld % 4 == 0
    base = torch.randn(M, V, generator=g)
    scale = torch.tensor([0.3, 1.0, 3.0, 6.0, 12.0, 25.0])[torch.arange(M) % 6][:, None]
    rows = base * scale
    rows[7, 10:50] = rows[7].max() + 0.5          # 40 equal maxima
    rows[8, 100:400] = rows[8].max() + 0.5        # 300 equal maxima (beyond the first stage's capacity)
    rows[9, 3:150] = rows[9].max() + 2.0          # 147 equal maxima (block list mode)
    rows[10] = 1.25                               # all equal
    rows[11, 5] = float("nan")
    rows[12, ::3] = float("-inf")
    rows[13, V - 3:] = rows[13].max() + 4.0       # the winners sit in the tail elements
    buf = torch.zeros(M, ld)
    buf[:, :V] = rows
    logits = buf.to(DEV)
    noise = torch.empty(M, V).exponential_(1, generator=g).to(DEV)
    conf = torch.zeros(M, device=DEV)
    x0 = torch.zeros(M, dtype=torch.long, device=DEV)

    def run(stages, T, top_p, mode, philox):
        H.check(lib.dark_dbg_set_sampler_stages(stages))
        conf.fill_(-7.0)
        x0.fill_(-7)
        if philox:
            H.check(lib.dark_sample_rows_philox(H.ptr(logits), ld, M, V, T, top_p, 0, mode, 77, 4, 256 * 512, H.ptr(conf),
                                                H.ptr(x0), _s()))
        else:
            H.check(lib.dark_sample_rows(H.ptr(logits), ld, M, V, T, top_p, 0, mode, H.ptr(noise) if T > 0 else None,
                                         V if T > 0 else 0, H.ptr(conf), H.ptr(x0), _s()))
        torch.cuda.synchronize()
        return conf.cpu().view(torch.int32).clone(), x0.cpu().clone()

    try:
        for T, top_p in ((0.1, 0.9), (0.5, 0.9), (1.0, 0.95), (0.0, 0.9), (0.3, 0.0), (2.0, 0.5)):
            for mode in (0, 1, 2):
                for philox in ((False, True) if T > 0 else (False,)):
                    c1, x1 = run(1, T, top_p, mode, philox)
                    c2, x2 = run(0, T, top_p, mode, philox)
                    assert torch.equal(x1, x2), (T, top_p, mode, philox, (x1 != x2).nonzero().flatten()[:8].tolist())
                    assert torch.equal(c1, c2), (T, top_p, mode, philox, (c1 != c2).nonzero().flatten()[:8].tolist())
                    assert int(x2.min()) >= 0 and int(x2.max()) < V
    finally:
        H.check(lib.dark_dbg_set_sampler_stages(0))


@pytest.mark.parametrize("V", [25427, 15081])
def test_sampler_unpadded_stride_equals_padded_two_stage(lib, V):
    """A caller that hands over logits on a stride of V with V % 4 != 0 (sample_into, the expand loop) is served by the
    single-stage kernel's scalar row load; head_and_sample pads the stride and takes the two stages.  Same rows, same
    noise: conf and x0 bit for bit (the dispatch depends on ldl % 4 and the base alignment)."""
    H = _hip()
    g = torch.Generator().manual_seed(5 + V)
    M = 200
    rows = torch.randn(M, V, generator=g) * torch.tensor([0.3, 1.0, 6.0, 25.0])[torch.arange(M) % 4][:, None]
    rows[3, 7:60] = rows[3].max() + 0.5
    rows[4, V - 2:] = rows[4].max() + 3.0
    noise = torch.empty(M, V).exponential_(1, generator=g).to(DEV)
    ldp = (V + 3) & ~3
    padded = torch.zeros(M, ldp)
    padded[:, :V] = rows
    padded = padded.to(DEV)
    plain = rows.to(DEV).contiguous()
    assert plain.stride(0) % 4 != 0
    out = {}
    for name, lg, ld in (("plain", plain, V), ("padded", padded, ldp)):
        for T, top_p, mode in ((0.1, 0.9, 2), (0.5, 0.9, 0), (1.0, 0.95, 1), (0.0, 0.9, 2)):
            conf = torch.full((M,), -7.0, device=DEV)
            x0 = torch.full((M,), -7, dtype=torch.long, device=DEV)
            H.check(lib.dark_sample_rows(H.ptr(lg), ld, M, V, T, top_p, 0, mode, H.ptr(noise) if T > 0 else None,
                                         V if T > 0 else 0, H.ptr(conf), H.ptr(x0), _s()))
            torch.cuda.synchronize()
            out[(name, T, top_p, mode)] = (conf.cpu().view(torch.int32), x0.cpu())
    for (name, T, top_p, mode), (c, x) in out.items():
        if name == "plain":
            cp, xp = out[("padded", T, top_p, mode)]
            assert torch.equal(x, xp) and torch.equal(c, cp), (T, top_p, mode)
            assert int(x.min()) >= 0 and int(x.max()) < V


def test_sample_rows_kept_set_matches_top_p(lib):
    """Property at full size: with noise concentrated on one token, x0 must be that token iff it survives
    the nucleus filter of the oracle."""
    g = torch.Generator().manual_seed(13)
    M, V = 64, 15080
    logits = torch.randn(M, V, generator=g)
    filt = O.top_p_logits(logits / 0.7, 0.9)
    kept = filt > torch.finfo(torch.float32).min
    probe = torch.randint(0, V, (M,), generator=g)
    noise = torch.full((M, V), 1e30)
    noise[torch.arange(M), probe] = 1e-30
    conf, x0 = _sample_hip(lib, logits, 0.7, 0.9, None, _hip().CONF_PROB, noise)
    exp_hit = kept[torch.arange(M), probe]
    assert torch.equal(x0 == probe, exp_hit)
    assert 0 < int(exp_hit.sum()) < M  # the probe set exercises both outcomes


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("T", [1.0, 0.5, 0.0])
def test_sample_rows_head_set_sizes(lib, mode, T):
    """Rows whose mass sits in exactly c elements near the maximum, c around the boundaries between the sampler's
    paths: <= 64 candidates (wave 0 finishes the row with shuffles), 65..512 (block-wide list mode), > 512 (the head
    gather overflows: histogram path).  Every third row makes the head elements equal, so that the nucleus boundary
    falls inside a run of equal logits (only the first few by index survive).  All against oracle.sample_tokens."""
    g = torch.Generator().manual_seed(21)
    V = 15080
    sizes = [1, 2, 3, 17, 63, 64, 65, 66, 128, 301, 511, 512, 513, 701]   # (none with 0.9 * c an integer)
    rows = []
    for r, c in enumerate(sizes * 3):
        l = torch.randn(V, generator=g) * 0.3 - 30.0                      # tail: no mass to speak of
        idx = torch.randperm(V, generator=g)[:c]
        l[idx] = (2.0 if r % 3 == 1 else 2.0 + torch.rand(c, generator=g) * 4.0)  # head: within 6 of the maximum
        if r % 3 == 2 and c > 2:
            l[idx[: c // 2]] = 5.5                                         # half of the head exactly equal, on top
        rows.append(l)
    logits = torch.stack(rows)
    M = logits.shape[0]
    noise = torch.empty(M, V).exponential_(1, generator=g) if T > 0 else None
    margin, negent = mode == _hip().CONF_MARGIN, mode == _hip().CONF_NEG_ENTROPY
    rc, rx = O.sample_tokens(logits, temperature=T, top_p=0.9, top_k=None, margin_confidence=margin,
                             neg_entropy=negent, noise=noise, stable_ties=True)
    conf, x0 = _sample_hip(lib, logits, T, 0.9, None, mode, noise)
    assert torch.equal(x0, rx), (x0 != rx).nonzero().view(-1).tolist()
    np.testing.assert_allclose(conf.numpy(), rc.numpy(), rtol=3e-5, atol=1e-6)


# --------------------------------------------------------------------------- mask rows + commit
def _commit_ref(x, mask_id, conf, x0, k):
    """generation_utils.py:522-534 with ties -> lower index."""
    x = x.clone()
    mask_index = x == mask_id
    full = torch.full(x.shape, -torch.inf)
    full[mask_index] = conf
    # stable descending sort = (confidence desc, index asc)
    order = torch.sort(full, dim=1, descending=True, stable=True).indices[:, :k]
    x_ = torch.zeros_like(x) + mask_id
    x_[mask_index] = x0
    rows = torch.arange(x.size(0)).unsqueeze(1).expand_as(order)
    x[rows, order] = x_[rows, order]
    return x


@pytest.mark.parametrize("B,N,k", [(8, 46, 1), (8, 46, 3), (64, 46, 11), (5, 128, 7), (3, 200, 4), (1024, 46, 2),
                                   (2100, 128, 3)])  # the last one exceeds 256 tiles: single-workgroup kernel
def test_mask_rows_and_commit(lib, B, N, k):
    H = _hip()
    g = torch.Generator().manual_seed(B * N + k)
    mask_id = 6
    x = torch.randint(100, 500, (B, N), generator=g)
    x[torch.rand(B, N, generator=g) < 0.3] = mask_id
    x[0] = 7          # a row with no mask at all (re-mask quirk: k positions get MASK)
    x[1, :] = mask_id  # a fully masked row
    xd = x.to(DEV)
    ri = torch.zeros(B * N, dtype=torch.int32, device=DEV)
    sr = torch.zeros(B * N, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(4, dtype=torch.int32, device=DEV)
    H.check(lib.dark_mask_rows(H.ptr(xd), B, N, mask_id, H.ptr(ri), H.ptr(sr), H.ptr(cnt), _s()))
    torch.cuda.synchronize()
    flat = (x.view(-1) == mask_id).nonzero().view(-1)
    M = flat.numel()
    assert int(cnt[0]) == M
    assert torch.equal(ri[:M].cpu().long(), flat)
    assert torch.equal(sr[:M].cpu().long(), torch.where(flat % N == 0, flat, flat - 1))
    conf = torch.randn(M, generator=g)
    conf = torch.round(conf * 4) / 4  # many exact ties
    x0 = torch.randint(100, 500, (M,), generator=g)
    cd, x0d = conf.to(DEV), x0.to(DEV)
    H.check(lib.dark_commit_topk(H.ptr(xd), B, N, mask_id, H.ptr(ri), M, H.ptr(cd), H.ptr(x0d), k, _s()))
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), _commit_ref(x, mask_id, conf, x0, k))


def test_commit_nan_confidence_counts_as_largest(lib):
    """torch.topk orders NaN as the largest value and still selects exactly k (generation_utils.py:526): a NaN
    confidence must be transferred first, not `every NaN regardless of k`; -0.0 ties with +0.0 (lower index wins)."""
    H = _hip()
    mask_id, B, N = 6, 2, 46
    x = torch.full((B, N), 9, dtype=torch.long)
    x[0, [3, 7, 11, 20, 30]] = mask_id
    x[1, [1, 2, 5]] = mask_id
    conf = torch.tensor([0.5, float("nan"), 0.9, float("nan"), float("nan"), 0.0, -0.0, -1.0])
    x0 = torch.arange(100, 108)
    for k in (1, 2, 4):
        xd = x.to(DEV)
        ri = torch.zeros(B * N, dtype=torch.int32, device=DEV)
        sr = torch.zeros(B * N, dtype=torch.int32, device=DEV)
        cnt = torch.zeros(4, dtype=torch.int32, device=DEV)
        H.check(lib.dark_mask_rows(H.ptr(xd), B, N, mask_id, H.ptr(ri), H.ptr(sr), H.ptr(cnt), _s()))
        cd, x0d = conf.to(DEV), x0.to(DEV)  # kept alive across the asynchronous launch
        H.check(lib.dark_commit_topk(H.ptr(xd), B, N, mask_id, H.ptr(ri), 8, H.ptr(cd), H.ptr(x0d), k, _s()))
        torch.cuda.synchronize()
        got = xd.cpu()
        # row 0: NaNs at masked slots 1, 3, 4 (positions 7, 20, 30) go first in index order, then 0.9, then 0.5
        order0 = [7, 20, 30, 11, 3]
        exp = x.clone()
        for j in order0[:k]:
            exp[0, j] = x0[[3, 7, 11, 20, 30].index(j)]
        # row 1: 0.0 (pos 1) ties with -0.0 (pos 2): the lower index first; -1.0 last; k = 4 re-masks position 0
        order1 = [1, 2, 5]
        for j in order1[:k]:
            exp[1, j] = x0[5 + order1.index(j)]
        if k == 4:
            exp[1, 0] = mask_id
        assert torch.equal(got, exp), (k, got[0].tolist(), got[1].tolist())


def test_top_p_zero_keeps_only_the_top_token(lib):
    """`top_p=0.0` is a live filter in the reference (`top_p is not None and top_p < 1`, generation_utils.py:62):
    `cum > 0` removes everything after the first sorted token, so sampling is greedy whatever the noise says."""
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(64, 15080, generator=g)
    noise = torch.empty_like(logits).exponential_(1, generator=g)
    rc, rx = O.sample_tokens(logits, temperature=0.7, top_p=0.0, neg_entropy=True, noise=noise)
    conf, x0 = _sample_hip(lib, logits, 0.7, 0.0, None, _mode(False, True), noise)
    assert torch.equal(x0, logits.argmax(-1)) and torch.equal(x0, rx)
    np.testing.assert_allclose(conf.numpy(), rc.numpy(), atol=1e-6)
    conf2, x02 = _sample_hip(lib, logits, 0.7, None, None, _mode(False, True), noise)  # None = no filter
    assert not torch.equal(x02, x0)


def test_error_paths(lib):
    H = _hip()
    with pytest.raises(H.DarkHipError):
        H.check(lib.dark_sample_rows(None, 10, 1, 10, 0.0, 0.0, 0, 0, None, 0, None, None, None))
    a = torch.zeros(4, 100, device=DEV)
    c = torch.zeros(4, device=DEV)
    i = torch.zeros(4, dtype=torch.long, device=DEV)
    with pytest.raises(H.DarkHipError):  # temperature > 0 without noise
        H.check(lib.dark_sample_rows(H.ptr(a), 100, 4, 100, 0.5, 0.9, 0, 2, None, 0, H.ptr(c), H.ptr(i), _s()))
    with pytest.raises(H.DarkHipError):  # K not a multiple of 64
        H.check(lib.dark_dbg_gemm(0, 3, H.ptr(a), H.ptr(a), None, H.ptr(a), 4, 4, 100, 4, _s()))


def _fill_plan(numel):
    props = torch.cuda.get_device_properties(0)
    grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)
    return grid * 256, ((numel - 1) // (grid * 256 * 4) + 1) * 4


@pytest.mark.parametrize("numel", [1, 7, 256, 1000, 15080 * 11, 2048 * 256 * 4 + 5, 15080 * 2000])
def test_philox_exponential_is_torchs_stream(lib, numel):
    """dark_philox_exponential == torch.empty(numel).exponential_() bit for bit, and the generator offset ATen
    reserves for that fill is the one dark_amd.generation_utils.philox_fill_plan reserves."""
    H = _hip()
    gen = torch.cuda.default_generators[0]
    torch.manual_seed(4321 + numel)
    torch.empty(37, device=DEV).exponential_()           # start from a non-zero offset
    seed, off0 = gen.initial_seed(), gen.get_offset()
    ref = torch.empty(numel, device=DEV).exponential_()
    nth, adv = _fill_plan(numel)
    assert gen.get_offset() - off0 == adv
    out = torch.empty(numel, device=DEV)
    H.check(lib.dark_philox_exponential(seed, off0, nth, numel, H.ptr(out), _s()))
    assert torch.equal(out, ref)
    from dark_amd.generation_utils import philox_fill_plan
    gen.set_offset(off0)
    assert philox_fill_plan(torch.device(DEV), numel) == (seed, off0, nth) and gen.get_offset() - off0 == adv


@pytest.mark.parametrize("T,top_p,top_k,mode", [(0.1, 0.9, 0, 2), (0.5, 0.9, 0, 2), (1.0, -1.0, 0, 0), (0.7, -1.0, 20, 1)])
def test_sample_rows_philox_equals_materialised_noise(lib, T, top_p, top_k, mode):
    """The sampler fed by the lazy Philox draws == the sampler fed by torch's exponential_ fill of the same state."""
    H = _hip()
    g = torch.Generator().manual_seed(5)
    M, V = 300, 15080
    logits = (torch.randn(M, V, generator=g) * 0.6).to(DEV)
    logits[3] *= 8
    logits[5] = 0.0
    gen = torch.cuda.default_generators[0]
    torch.manual_seed(99)
    seed, off0 = gen.initial_seed(), gen.get_offset()
    noise = torch.empty(M, V, device=DEV).exponential_()
    nth, _ = _fill_plan(M * V)
    c1, x1 = torch.empty(M, device=DEV), torch.empty(M, dtype=torch.long, device=DEV)
    c2, x2 = torch.empty(M, device=DEV), torch.empty(M, dtype=torch.long, device=DEV)
    H.check(lib.dark_sample_rows(H.ptr(logits), V, M, V, T, top_p, top_k, mode, H.ptr(noise), V, H.ptr(c1), H.ptr(x1), _s()))
    H.check(lib.dark_sample_rows_philox(H.ptr(logits), V, M, V, T, top_p, top_k, mode, seed, off0, nth, H.ptr(c2),
                                        H.ptr(x2), _s()))
    assert torch.equal(x1, x2) and torch.equal(c1, c2)
