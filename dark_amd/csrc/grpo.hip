// Forward-only log-probability path of the diffusion GRPO trainer:
// akgr/abduction_model/custom_grpo2.py:350-387 (forward_process), :409-418 (get_logits),
// :421-498 (_get_per_token_logps), :89-161 (selective_log_softmax), as used under no_grad for the old / reference
// policy log-probs (:681-709).  The [rows, V] logits are never written for bf16 models: the LM-head GEMM ends in a
// log-sum-exp epilogue (gemm.hip, EPI_LSE) and a one-wave-per-row reduction finishes log_softmax + gather.
#include "common.h"

namespace dark {

// ------------------------------------------------------------------ forward_process (:373-385)
// noisy[0] = where(target, mask, batch); noisy[1] = where(rand < t & target, ...); noisy[2] = where(rand > t & target, ...)
__global__ __launch_bounds__(256) void grpo_perturb_kernel(const int64_t* __restrict__ batch,
                                                           const uint8_t* __restrict__ target,
                                                           const float* __restrict__ rnd, float t_p, int64_t mask_id,
                                                           int total, int64_t* __restrict__ noisy,
                                                           uint8_t* __restrict__ completion_mask) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= total) return;
  const int64_t tok = batch[f];
  const bool q = target[f] != 0;
  const float r = rnd[f];
  const bool m1 = (r < t_p) && q, m2 = (r > t_p) && q;
  noisy[f] = q ? mask_id : tok;
  noisy[(int64_t)total + f] = m1 ? mask_id : tok;
  noisy[2ll * total + f] = m2 ? mask_id : tok;
  completion_mask[f] = m1 ? 1 : 0;
}

// src_rows[s*N + j] = s*N + max(j-1, 0): the logit shift of get_logits (:412-414)
__global__ __launch_bounds__(256) void shift_rows_kernel(int32_t* __restrict__ src_rows, int total, int N) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= total) return;
  src_rows[f] = (f % N == 0) ? f : f - 1;
}

// ------------------------------------------------------------------ log_softmax + gather, second stage
// fused: combine the P (max, sum) partials of a row; one wave per row
__global__ __launch_bounds__(256) void logprob_reduce_kernel(const float2* __restrict__ part,
                                                             const float* __restrict__ tgt, int M, int P,
                                                             float* __restrict__ logp) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  const float2* p = part + (int64_t)m * P;
  float mx = -INFINITY;
  for (int i = lane; i < P; i += 64) mx = fmaxf(mx, p[i].x);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < P; i += 64) {
    const float2 v = p[i];
    if (v.x > -INFINITY) s += v.y * __builtin_amdgcn_exp2f((v.x - mx) * 1.44269504088896341f);
  }
  s = wave_sum(s);
  if (lane == 0) logp[m] = tgt[m] - mx - logf(s);
}

// unfused: from materialised fp32 logits (fp32 models; also the A/B baseline of the fused path)
__global__ __launch_bounds__(256) void logprob_logits_kernel(const float* __restrict__ logits, int64_t ldl,
                                                             const int64_t* __restrict__ targets, int M, int V,
                                                             float* __restrict__ logp) {
  __shared__ float red[4];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)m * ldl;
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256) mx = fmaxf(mx, row[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < V; c += 256) s += expf(row[c] - mx);
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    int64_t t = targets[m];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    logp[m] = row[t] - mx - logf(red[0] + red[1] + red[2] + red[3]);
  }
}

// ------------------------------------------------------------------ selective_log_softmax's weighting (:136-152)
// out[i*B+b, j] = (lp[(3i)*B+b, j] + (mask ? lp[(3i+1)*B+b, j] * w[3i+1] : lp[(3i+2)*B+b, j] * w[3i+2])) / 2
__global__ __launch_bounds__(256) void grpo_combine_kernel(const float* __restrict__ lp,
                                                           const float* __restrict__ w,
                                                           const uint8_t* __restrict__ mask, int I, int B, int N,
                                                           float* __restrict__ out) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  const int per = B * N;
  if (f >= I * per) return;
  const int i = f / per, r = f - i * per;
  const float* base = lp + (int64_t)3 * i * per;
  const float p0 = base[r], p1 = base[per + r], p2 = base[2 * per + r];
  const float wl = mask[f] ? p1 * w[3 * i + 1] : p2 * w[3 * i + 2];
  out[f] = (p0 + wl) / 2.0f;
}

static constexpr int kLogitChunk = 4096;  // rows of fp32 logits materialised at a time on the unfused path

static size_t logprob_ws(const dark_model_desc* m, int M) {
  const size_t xn = align_up((size_t)M * m->d * (m->dtype == DARK_BF16 ? 2 : 4), 256);
  if (m->dtype == DARK_BF16) {
    const size_t P = 2 * (size_t)cdiv(m->vocab, 128);
    return xn + align_up((size_t)M * P * sizeof(float2) + (size_t)M * sizeof(float), 256);
  }
  const size_t rows = (size_t)(M < kLogitChunk ? M : kLogitChunk);
  return xn + align_up(rows * (size_t)m->vocab * sizeof(float), 256);
}

}  // namespace dark

using namespace dark;

extern "C" {

size_t dark_logprob_workspace_bytes(const dark_model_desc* m, int M) {
  if (!m || M <= 0) return 0;
  return logprob_ws(m, M);
}

int dark_lm_head_logprob(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int n_src,
                         const int64_t* targets, int M, void* workspace, size_t workspace_bytes, float* logp,
                         void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DARK_CHECK_ARG(m && hidden && targets && logp && workspace, "lm_head_logprob: null pointer");
  DARK_CHECK_ARG(M > 0 && n_src > 0, "lm_head_logprob: bad M=%d n_src=%d", M, n_src);
  DARK_CHECK_ARG(workspace_bytes >= logprob_ws(m, M), "lm_head_logprob: workspace %zu < %zu bytes", workspace_bytes,
                 logprob_ws(m, M));
  char* ws = (char*)workspace;
  void* xn = ws;
  ws += align_up((size_t)M * m->d * (m->dtype == DARK_BF16 ? 2 : 4), 256);
  DARK_TRY(launch_rmsnorm(m->dtype, m->resid_dtype, hidden, m->final_norm, xn, src_rows, n_src, M, m->d, m->rms_eps, st));
  if (m->dtype == DARK_BF16) {
    const int P = 2 * cdiv(m->vocab, 128);
    float2* part = (float2*)ws;
    const float* tgt = (const float*)(part + (size_t)M * P);
    DARK_TRY(launch_gemm(DARK_BF16, EPI_LSE, xn, m->lm_head, (const float*)targets, part, M, m->vocab, m->d, P, st));
    hipLaunchKernelGGL(logprob_reduce_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, part, tgt, M, P, logp);
    DARK_CHECK_LAUNCH();
    return DARK_OK;
  }
  float* logits = (float*)ws;
  for (int r0 = 0; r0 < M; r0 += kLogitChunk) {
    const int rows = M - r0 < kLogitChunk ? M - r0 : kLogitChunk;
    DARK_TRY(launch_gemm(DARK_F32, EPI_F32, (const float*)xn + (size_t)r0 * m->d, m->lm_head, nullptr, logits, rows,
                         m->vocab, m->d, m->vocab, st));
    hipLaunchKernelGGL(logprob_logits_kernel, dim3(rows), dim3(256), 0, st, logits, (int64_t)m->vocab, targets + r0,
                       rows, m->vocab, logp + r0);
    DARK_CHECK_LAUNCH();
  }
  return DARK_OK;
}

int dark_logprob_from_logits(const float* logits, int64_t ldl, const int64_t* targets, int M, int V, float* logp,
                             void* stream) {
  DARK_CHECK_ARG(logits && targets && logp, "logprob_from_logits: null pointer");
  DARK_CHECK_ARG(M > 0 && V > 0 && ldl >= V, "logprob_from_logits: bad M=%d V=%d ldl=%lld", M, V, (long long)ldl);
  hipLaunchKernelGGL(logprob_logits_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, ldl, targets, M, V,
                     logp);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int dark_grpo_perturb(const int64_t* batch, const uint8_t* mask_target, const float* rnd, float t_p, int64_t mask_id,
                      int b, int l, int64_t* noisy, uint8_t* completion_mask, void* stream) {
  DARK_CHECK_ARG(batch && mask_target && rnd && noisy && completion_mask, "grpo_perturb: null pointer");
  DARK_CHECK_ARG(b > 0 && l > 0 && (int64_t)b * l < (1ll << 29), "grpo_perturb: bad b=%d l=%d", b, l);
  const int total = b * l;
  hipLaunchKernelGGL(grpo_perturb_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, batch,
                     mask_target, rnd, t_p, mask_id, total, noisy, completion_mask);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int dark_shift_rows(int32_t* src_rows, int S, int N, void* stream) {
  DARK_CHECK_ARG(src_rows && S > 0 && N > 0 && (int64_t)S * N < (1ll << 30), "shift_rows: bad S=%d N=%d", S, N);
  hipLaunchKernelGGL(shift_rows_kernel, dim3(cdiv(S * N, 256)), dim3(256), 0, (hipStream_t)stream, src_rows, S * N,
                     N);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int dark_grpo_combine(const float* logp, const float* weights, const uint8_t* mask, int I, int B, int N, float* out,
                      void* stream) {
  DARK_CHECK_ARG(logp && weights && mask && out, "grpo_combine: null pointer");
  DARK_CHECK_ARG(I > 0 && B > 0 && N > 0 && (int64_t)I * B * N < (1ll << 29), "grpo_combine: bad I=%d B=%d N=%d", I, B,
                 N);
  hipLaunchKernelGGL(grpo_combine_kernel, dim3(cdiv(I * B * N, 256)), dim3(256), 0, (hipStream_t)stream, logp, weights,
                     mask, I, B, N, out);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

}  // extern "C"
