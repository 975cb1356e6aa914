// GPT-2 backbone of the legacy diffusion denoiser: akgr/abduction_model/model/custom_gpt.py:25-114
// (`DiffusionModel(GPT2LMHeadModel)`, built by create_gpt2_transformer :446-472 from the stock `gpt2` config: 12 layers,
// 12 heads of 64, d = 768, learned positions, LayerNorm, GELU-new, LM head tied to the token embedding).
//
// What is new against the Dream body of api.hip: token + learned position embedding, LayerNorm (mean / variance, weight
// AND bias) in place of RMSNorm, biases on every projection, GELU-new in place of SwiGLU, no rotary embedding.  The
// GEMMs are the kernels of gemm.hip with two more epilogues (EPI_BIAS_GELU, EPI_RESID_BIAS); attention is attn.hip
// with identity rotation tables (cos = 1, sin = 0: x * 1 - y * 0 is exact) and, optionally, the causal mask.
//
// Attention semantics.  The class sets every block's `attn.bias` buffer to ones "to allow bidirectional attention"
// (:34-36) and passes a [B, N] key-padding mask (:356, :82-83).  Whether that made the pinned transformers 4.50.2 run
// bidirectionally depends on the attention implementation it picked (its sdpa path builds the causal mask without
// looking at the buffer), so both are provided: causal = 0 (the stated intent, default) and causal = 1.
#include "common.h"

namespace dark {

// ------------------------------------------------------------------ wte[id] + wpe[pos] -> residual stream (TH)
template <typename TW, typename TH>
__global__ __launch_bounds__(256) void gpt2_embed_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ pos,
                                                         const TW* __restrict__ wte, const TW* __restrict__ wpe,
                                                         TH* __restrict__ out, int M, int N, int d, int vocab,
                                                         int n_positions) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // (nn.Embedding device-asserts; no access leaves the table here)
  int p = pos ? pos[row] : row % N;
  p = p < 0 ? 0 : (p >= n_positions ? n_positions - 1 : p);
  const TW* a = wte + id * d;
  const TW* b = wpe + (int64_t)p * d;
  TH* o = out + (int64_t)row * d;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 x = load4<TW>(a + c), y = load4<TW>(b + c);
    store4<TH>(o + c, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
  }
}

// ------------------------------------------------------------------ LayerNorm (torch.nn.LayerNorm: biased variance)
// One wave per row, the row in registers for d <= 1024 (three passes over registers: mean, centred squares, output).
template <typename TOut, typename TH>
__global__ __launch_bounds__(256) void layernorm_kernel(const TH* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, TOut* __restrict__ out,
                                                        const int32_t* __restrict__ src_rows, int n_src, int M, int d,
                                                        float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  int64_t src = row;
  if (src_rows) {
    src = src_rows[row];
    src = src < 0 ? 0 : (src >= n_src ? n_src - 1 : src);
  }
  const TH* xr = x + src * d;
  TOut* o = out + (int64_t)row * d;
  auto emit = [&](int c, const float4 v, float mean, float r) {
    const float4 g = *(const float4*)(w + c), h = *(const float4*)(b + c);
    store4<TOut>(o + c, make_float4((v.x - mean) * r * g.x + h.x, (v.y - mean) * r * g.y + h.y,
                                    (v.z - mean) * r * g.z + h.z, (v.w - mean) * r * g.w + h.w));
  };
  if (d <= 1024) {
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c = lane * 4 + p * 256;
      if (c < d) {
        v[p] = load4<TH>(xr + c);
        s += (v[p].x + v[p].y) + (v[p].z + v[p].w);
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (lane * 4 + p * 256 < d) {
        const float a0 = v[p].x - mean, a1 = v[p].y - mean, a2 = v[p].z - mean, a3 = v[p].w - mean;
        ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
      }
    const float r = 1.0f / sqrtf(wave_sum(ss) / (float)d + eps);
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (lane * 4 + p * 256 < d) emit(lane * 4 + p * 256, v[p], mean, r);
    return;
  }
  float s = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = load4<TH>(xr + c);
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = load4<TH>(xr + c);
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
  }
  const float r = 1.0f / sqrtf(wave_sum(ss) / (float)d + eps);
  for (int c = lane * 4; c < d; c += 256) emit(c, load4<TH>(xr + c), mean, r);
}

int launch_layernorm(int dtype, int xdt, const void* x, const float* w, const float* b, void* out, const int32_t* src_rows,
                     int n_src, int M, int d, float eps, hipStream_t st) {
  DARK_CHECK_ARG(d % 4 == 0 && x && w && b && out, "layernorm: bad argument (d=%d)", d);
  DARK_CHECK_ARG(xdt == DARK_F32 || (xdt == DARK_BF16 && dtype == DARK_BF16), "layernorm: stream dtype %d with model dtype %d",
                 xdt, dtype);
  if (M <= 0) return DARK_OK;
  const dim3 grid(cdiv(M, 4)), block(256);
  if (dtype == DARK_BF16 && xdt == DARK_BF16)
    hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)x, w, b, (bf16_t*)out, src_rows,
                       n_src, M, d, eps);
  else if (dtype == DARK_BF16)
    hipLaunchKernelGGL((layernorm_kernel<bf16_t, float>), grid, block, 0, st, (const float*)x, w, b, (bf16_t*)out, src_rows,
                       n_src, M, d, eps);
  else
    hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, block, 0, st, (const float*)x, w, b, (float*)out, src_rows, n_src,
                       M, d, eps);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

static size_t esz(int dtype) { return dtype == DARK_BF16 ? 2 : 4; }

static int check_gpt2(const dark_gpt2_desc* m) {
  DARK_CHECK_ARG(m != nullptr, "null GPT-2 descriptor");
  DARK_CHECK_ARG(m->dtype == DARK_F32 || m->dtype == DARK_BF16, "bad dtype %d", m->dtype);
  DARK_CHECK_ARG(m->d > 0 && m->d % 64 == 0 && m->ff > 0 && m->ff % 64 == 0, "n_embd=%d / n_inner=%d must be multiples of 64",
                 m->d, m->ff);
  DARK_CHECK_ARG(m->n_heads > 0 && m->n_heads * m->head_dim == m->d, "heads %d x head_dim %d != n_embd %d", m->n_heads,
                 m->head_dim, m->d);
  DARK_CHECK_ARG(m->head_dim == 64 || m->head_dim == 96, "head_dim=%d must be 64 or 96", m->head_dim);
  DARK_CHECK_ARG(m->vocab >= 2 && m->vocab_pad >= m->vocab && m->vocab_pad % 128 == 0, "vocab=%d vocab_pad=%d", m->vocab,
                 m->vocab_pad);
  DARK_CHECK_ARG(m->n_layers >= 0 && (m->n_layers == 0 || m->layers) && m->n_positions >= 1, "layers / positions missing");
  DARK_CHECK_ARG(m->wte && m->wpe && m->lnf_w && m->lnf_b && m->ident_cos && m->ident_sin, "null weight pointer");
  DARK_CHECK_ARG(m->resid_dtype == DARK_F32 || (m->resid_dtype == DARK_BF16 && m->dtype == DARK_BF16),
                 "resid_dtype %d with model dtype %d (fp32 always, bf16 for bf16 models)", m->resid_dtype, m->dtype);
  return DARK_OK;
}

struct Gpt2Ws {
  char *xn, *qkv, *att, *act;
  size_t total;
};
static Gpt2Ws carve_gpt2(const dark_gpt2_desc* m, size_t rows, char* base) {
  const size_t es = esz(m->dtype);
  Gpt2Ws w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };  // (null base: sizes only)
  w.xn = take(rows * m->d * es);
  w.qkv = take(rows * 3 * m->d * es);
  w.att = take(rows * m->d * es);
  w.act = take(rows * m->ff * es);
  w.total = off;
  return w;
}

}  // namespace dark

using namespace dark;

extern "C" {

size_t dark_gpt2_forward_workspace_bytes(const dark_gpt2_desc* m, int B, int N) {
  if (!m || B <= 0 || N <= 0) return 0;
  return carve_gpt2(m, (size_t)B * N, nullptr).total;
}

int dark_gpt2_forward_hidden(const dark_gpt2_desc* m, const int64_t* ids, const int32_t* pos, const uint8_t* key_mask,
                             int causal, int B, int N, void* workspace, size_t workspace_bytes, void* hidden,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DARK_TRY(check_gpt2(m));
  DARK_CHECK_ARG(B > 0 && N > 0 && N <= 128, "gpt2_forward: B=%d N=%d (N must be in [1,128])", B, N);
  DARK_CHECK_ARG((int64_t)B * N < (1ll << 30), "gpt2_forward: B*N too large");
  DARK_CHECK_ARG(ids && hidden && workspace, "gpt2_forward: null pointer");
  DARK_CHECK_ARG(pos != nullptr || N <= m->n_positions, "gpt2_forward: N=%d exceeds n_positions=%d", N, m->n_positions);
  const int M = B * N, d = m->d;
  Gpt2Ws w = carve_gpt2(m, (size_t)M, (char*)workspace);
  DARK_CHECK_ARG(workspace_bytes >= w.total, "gpt2_forward: workspace %zu < required %zu bytes", workspace_bytes, w.total);
  {
    const dim3 grid(cdiv(M, 4)), block(256);
    if (m->dtype == DARK_BF16 && m->resid_dtype == DARK_BF16)
      hipLaunchKernelGGL((gpt2_embed_kernel<bf16_t, bf16_t>), grid, block, 0, st, ids, pos, (const bf16_t*)m->wte,
                         (const bf16_t*)m->wpe, (bf16_t*)hidden, M, N, d, m->vocab, m->n_positions);
    else if (m->dtype == DARK_BF16)
      hipLaunchKernelGGL((gpt2_embed_kernel<bf16_t, float>), grid, block, 0, st, ids, pos, (const bf16_t*)m->wte,
                         (const bf16_t*)m->wpe, (float*)hidden, M, N, d, m->vocab, m->n_positions);
    else
      hipLaunchKernelGGL((gpt2_embed_kernel<float, float>), grid, block, 0, st, ids, pos, (const float*)m->wte,
                         (const float*)m->wpe, (float*)hidden, M, N, d, m->vocab, m->n_positions);
    DARK_CHECK_LAUNCH();
  }
  const int xdt = m->resid_dtype;
  const int epi_resid = xdt == DARK_BF16 ? EPI_RESID_BIAS_H : EPI_RESID_BIAS;
  for (int l = 0; l < m->n_layers; ++l) {
    const dark_gpt2_layer_weights& L = m->layers[l];
    DARK_CHECK_ARG(L.w_qkv && L.b_qkv && L.w_o && L.b_o && L.w_fc && L.b_fc && L.w_proj && L.b_proj && L.ln1_w && L.ln1_b &&
                       L.ln2_w && L.ln2_b, "gpt2 layer %d: null weight", l);
    DARK_TRY(launch_layernorm(m->dtype, xdt, hidden, L.ln1_w, L.ln1_b, w.xn, nullptr, M, M, d, m->ln_eps, st));
    DARK_TRY(launch_gemm(m->dtype, EPI_BIAS, w.xn, L.w_qkv, L.b_qkv, w.qkv, M, 3 * d, d, 3 * d, st));
    // learned positions were added to the embedding: the rotation tables hold one row of cos = 1 / sin = 0
    DARK_TRY(launch_attention(m->dtype, w.qkv, m->ident_cos, m->ident_sin, nullptr, key_mask, w.att, B, N, m->n_heads,
                              m->head_dim, 1, st, causal != 0));
    DARK_TRY(launch_gemm(m->dtype, epi_resid, w.att, L.w_o, L.b_o, hidden, M, d, d, d, st));
    DARK_TRY(launch_layernorm(m->dtype, xdt, hidden, L.ln2_w, L.ln2_b, w.xn, nullptr, M, M, d, m->ln_eps, st));
    DARK_TRY(launch_gemm(m->dtype, EPI_BIAS_GELU, w.xn, L.w_fc, L.b_fc, w.act, M, m->ff, d, m->ff, st));
    DARK_TRY(launch_gemm(m->dtype, epi_resid, w.act, L.w_proj, L.b_proj, hidden, M, d, m->ff, d, st));
  }
  return DARK_OK;
}

int dark_gpt2_lm_head_rows(const dark_gpt2_desc* m, const void* hidden, const int32_t* src_rows, int M, void* workspace,
                           size_t workspace_bytes, float* logits, int64_t ldl, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DARK_TRY(check_gpt2(m));
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(hidden && workspace && logits, "gpt2_lm_head_rows: null pointer");
  DARK_CHECK_ARG(ldl >= m->vocab, "gpt2_lm_head_rows: ldl=%lld < vocab=%d", (long long)ldl, m->vocab);
  const size_t need = align_up((size_t)M * m->d * esz(m->dtype), 256);
  DARK_CHECK_ARG(workspace_bytes >= need, "gpt2_lm_head_rows: workspace %zu < required %zu bytes", workspace_bytes, need);
  DARK_TRY(launch_layernorm(m->dtype, m->resid_dtype, hidden, m->lnf_w, m->lnf_b, workspace, src_rows, 1 << 30, M, m->d, m->ln_eps, st));
  return launch_gemm(m->dtype, EPI_F32, workspace, m->wte, nullptr, logits, M, m->vocab, m->d, ldl, st);
}

int dark_dbg_layernorm(int dtype, int x_dtype, const void* x, const float* w, const float* b, void* out,
                       const int32_t* src_rows, int n_src, int M, int d, float eps, void* stream) {
  return launch_layernorm(dtype, x_dtype, x, w, b, out, src_rows, n_src, M, d, eps, (hipStream_t)stream);
}

}  // extern "C"
