// HBM-bound stages of the denoise step: embedding gather, RMSNorm (optionally over gathered
// rows), mask-row compaction and the confidence-ranked commit.  One wave per row, 16-B accesses.
#include "common.h"

namespace dark {

// ------------------------------------------------------------------ embedding (modeling_dream_length.py:665)
template <typename TW, typename TH>
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids,
                                                    const TW* __restrict__ table,
                                                    TH* __restrict__ out, int M, int d, int vocab) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  int64_t id = ids[row];
  // nn.Embedding raises (device-asserts) on out-of-range ids; here they are clamped so that no access leaves
  // the table -- nothing validates them on the host, a caller that needs the error checks ids.max() itself
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const TW* src = table + id * d;
  TH* dst = out + (int64_t)row * d;
  for (int c = lane * 4; c < d; c += 256) store4<TH>(dst + c, load4<TW>(src + c));
}

// Embedding + the first layer's input RMSNorm in one pass (d <= 1024): the gathered row is already in registers, so
// xn = weight * row / rms(row) costs no second read of the 4*d-byte fp32 row.  Arithmetic and summation order are
// rmsnorm_kernel's.
template <typename TW, typename TH>
__global__ __launch_bounds__(256) void embed_norm_kernel(const int64_t* __restrict__ ids, const TW* __restrict__ table,
                                                         TH* __restrict__ out, const float* __restrict__ w,
                                                         TW* __restrict__ xn, int M, int d, int vocab, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const TW* src = table + id * d;
  float4 v[4], g[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = lane * 4 + p * 256;
    if (c < d) {
      v[p] = load4<TW>(src + c);
      g[p] = *(const float4*)(w + c);
      store4<TH>(out + (int64_t)row * d + c, v[p]);   // (a bf16 table row is exact in either stream type)
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    if (lane * 4 + p * 256 < d) ss += v[p].x * v[p].x + v[p].y * v[p].y + v[p].z * v[p].z + v[p].w * v[p].w;
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(ss / (float)d + eps);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = lane * 4 + p * 256;
    if (c < d) {
      store4<TW>(xn + (int64_t)row * d + c,
                 make_float4(g[p].x * (v[p].x * r), g[p].y * (v[p].y * r), g[p].z * (v[p].z * r), g[p].w * (v[p].w * r)));
    }
  }
}

// `xdt` = element type of the residual stream (DARK_F32, or DARK_BF16 for a bf16 model)
static int check_stream_dtype(int dtype, int xdt) {
  DARK_CHECK_ARG(xdt == DARK_F32 || (xdt == DARK_BF16 && dtype == DARK_BF16),
                 "residual stream dtype %d does not go with model dtype %d (fp32 always, bf16 for bf16 models)", xdt, dtype);
  return DARK_OK;
}

int launch_embed_norm(int dtype, int xdt, const int64_t* ids, const void* table, void* out, int M, int d, int vocab,
                      const float* ln_w, float eps, void* xn, hipStream_t st) {
  DARK_TRY(check_stream_dtype(dtype, xdt));
  if (d > 1024 || d % 4 != 0) {
    DARK_TRY(launch_embed(dtype, xdt, ids, table, out, M, d, vocab, st));
    return launch_rmsnorm(dtype, xdt, out, ln_w, xn, nullptr, M, M, d, eps, st);
  }
  DARK_CHECK_ARG(ids && table && out && ln_w && xn, "embed_norm: null pointer");
  const dim3 grid(cdiv(M, 4)), block(256);
  if (dtype == DARK_BF16 && xdt == DARK_BF16)
    hipLaunchKernelGGL((embed_norm_kernel<bf16_t, bf16_t>), grid, block, 0, st, ids, (const bf16_t*)table, (bf16_t*)out, ln_w,
                       (bf16_t*)xn, M, d, vocab, eps);
  else if (dtype == DARK_BF16)
    hipLaunchKernelGGL((embed_norm_kernel<bf16_t, float>), grid, block, 0, st, ids, (const bf16_t*)table, (float*)out, ln_w,
                       (bf16_t*)xn, M, d, vocab, eps);
  else
    hipLaunchKernelGGL((embed_norm_kernel<float, float>), grid, block, 0, st, ids, (const float*)table, (float*)out, ln_w,
                       (float*)xn, M, d, vocab, eps);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int launch_embed(int dtype, int xdt, const int64_t* ids, const void* table, void* out, int M, int d, int vocab,
                 hipStream_t st) {
  DARK_TRY(check_stream_dtype(dtype, xdt));
  DARK_CHECK_ARG(d % 4 == 0, "embed: d=%d must be a multiple of 4", d);
  const dim3 grid(cdiv(M, 4)), block(256);
  if (dtype == DARK_BF16 && xdt == DARK_BF16)
    hipLaunchKernelGGL((embed_kernel<bf16_t, bf16_t>), grid, block, 0, st, ids, (const bf16_t*)table, (bf16_t*)out, M, d, vocab);
  else if (dtype == DARK_BF16)
    hipLaunchKernelGGL((embed_kernel<bf16_t, float>), grid, block, 0, st, ids, (const bf16_t*)table, (float*)out, M, d, vocab);
  else
    hipLaunchKernelGGL((embed_kernel<float, float>), grid, block, 0, st, ids, (const float*)table, (float*)out, M, d, vocab);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ------------------------------------------------------------------ RMSNorm (modeling_dream_length.py:69-74)
template <typename TOut, typename TH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const TH* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      TOut* __restrict__ out,
                                                      const int32_t* __restrict__ src_rows, int n_src,
                                                      int M, int d, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  int64_t src = row;
  if (src_rows) {  // clamp: a stale or uninitialised index must never leave the buffer
    src = src_rows[row];
    src = src < 0 ? 0 : (src >= n_src ? n_src - 1 : src);
  }
  const TH* xr = x + src * d;
  TOut* o = out + (int64_t)row * d;
  auto emit = [&](int c, const float4 v, const float4 g, float r) {
    store4<TOut>(o + c, make_float4(g.x * (v.x * r), g.y * (v.y * r), g.z * (v.z * r), g.w * (v.w * r)));
  };
  if (d <= 1024) {
    // the row (and the weights) stay in registers between the two passes: one memory round trip per row, not two
    float4 v[4], g[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c = lane * 4 + p * 256;
      if (c < d) { v[p] = load4<TH>(xr + c); g[p] = *(const float4*)(w + c); }
    }
    float ss = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (lane * 4 + p * 256 < d) ss += v[p].x * v[p].x + v[p].y * v[p].y + v[p].z * v[p].z + v[p].w * v[p].w;
    ss = wave_sum(ss);
    const float r = 1.0f / sqrtf(ss / (float)d + eps);
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (lane * 4 + p * 256 < d) emit(lane * 4 + p * 256, v[p], g[p], r);
    return;
  }
  float ss = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = load4<TH>(xr + c);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(ss / (float)d + eps);
  for (int c = lane * 4; c < d; c += 256) emit(c, load4<TH>(xr + c), *(const float4*)(w + c), r);
}

// bf16 stream -> bf16 rows, d <= 1024 and d % 8 == 0: HALF a wave per row, 16-byte accesses on both sides (the
// one-wave-per-row kernel above moves 8 bytes a lane when the stream is bf16: 5.4 TB/s against 6.7 with fp32 input).
// Same arithmetic per element; the sum of squares is reduced over 32 lanes x 8-element chunks instead of 64 x 4, so
// the two kernels can differ by one bf16 ulp.
__global__ __launch_bounds__(256) void rmsnorm_h16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                          bf16_t* __restrict__ out, const int32_t* __restrict__ src_rows,
                                                          int n_src, int M, int d, float eps) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (row >= M) return;   // (whole half-waves leave: the shuffles below stay inside a half)
  int64_t src = row;
  if (src_rows) {
    src = src_rows[row];
    src = src < 0 ? 0 : (src >= n_src ? n_src - 1 : src);
  }
  const bf16_t* xr = x + src * d;
  // Summation order (part of the contract with gemm_rowpanel.hip, which rebuilds it to give the same bits): per 8-element
  // piece an fma chain from zero; this lane's pieces added in order; then the butterfly below.
  bf16x8 v[4];
  float ss = 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = l * 8 + p * 256;
    if (c < d) {
      v[p] = *(const bf16x8*)(xr + c);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) q = __builtin_fmaf((float)v[p][e], (float)v[p][e], q);
      ss = p == 0 ? q : ss + q;
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float r = 1.0f / sqrtf(ss / (float)d + eps);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = l * 8 + p * 256;
    if (c < d) {
      const float4 g0 = *(const float4*)(w + c), g1 = *(const float4*)(w + c + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      bf16x8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (bf16_t)(g[e] * ((float)v[p][e] * r));
      *(bf16x8*)(out + (int64_t)row * d + c) = y;
    }
  }
}

int launch_rmsnorm(int dtype, int xdt, const void* x, const float* w, void* out, const int32_t* src_rows, int n_src,
                   int M, int d, float eps, hipStream_t st) {
  DARK_TRY(check_stream_dtype(dtype, xdt));
  DARK_CHECK_ARG(d % 4 == 0, "rmsnorm: d=%d must be a multiple of 4", d);
  if (M <= 0) return DARK_OK;
  if (dtype == DARK_BF16 && xdt == DARK_BF16 && d <= 1024 && d % 8 == 0 &&
      (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w) & 15) == 0) {
    hipLaunchKernelGGL(rmsnorm_h16_kernel, dim3(cdiv(M, 8)), dim3(256), 0, st, (const bf16_t*)x, w, (bf16_t*)out, src_rows,
                       n_src, M, d, eps);
    DARK_CHECK_LAUNCH();
    return DARK_OK;
  }
  const dim3 grid(cdiv(M, 4)), block(256);
  if (dtype == DARK_BF16 && xdt == DARK_BF16)
    hipLaunchKernelGGL((rmsnorm_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)x, w, (bf16_t*)out, src_rows, n_src,
                       M, d, eps);
  else if (dtype == DARK_BF16)
    hipLaunchKernelGGL((rmsnorm_kernel<bf16_t, float>), grid, block, 0, st, (const float*)x, w, (bf16_t*)out, src_rows, n_src, M,
                       d, eps);
  else
    hipLaunchKernelGGL((rmsnorm_kernel<float, float>), grid, block, 0, st, (const float*)x, w, (float*)out, src_rows, n_src, M, d,
                       eps);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ------------------------------------------------------------------ row gather for the last layer's tail
// One wave per output row: the attention output row (model dtype) and the residual row (fp32 or bf16) of position
// rows[r].
template <typename T, typename TH>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ att, T* __restrict__ att_out,
                                                          const TH* __restrict__ x, TH* __restrict__ x_out,
                                                          const int32_t* __restrict__ rows, int n_src, int R, int d) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  int64_t src = rows[r];
  src = src < 0 ? 0 : (src >= n_src ? n_src - 1 : src);
  constexpr int EH = 16 / (int)sizeof(TH);
  for (int c = lane * EH; c < d; c += 64 * EH) *(uint4*)(x_out + (int64_t)r * d + c) = *(const uint4*)(x + src * d + c);
  if (att) {
    constexpr int E = 16 / (int)sizeof(T);  // elements per 16-byte piece
    for (int c = lane * E; c < d; c += 64 * E)
      *(uint4*)(att_out + (int64_t)r * d + c) = *(const uint4*)(att + src * d + c);
  }
}

int launch_gather_rows(int dtype, int xdt, const void* att, void* att_out, const void* x, void* x_out, const int32_t* rows,
                       int n_src, int R, int d, hipStream_t st) {
  DARK_TRY(check_stream_dtype(dtype, xdt));
  DARK_CHECK_ARG(d % 8 == 0, "gather_rows: d=%d must be a multiple of 8", d);
  DARK_CHECK_ARG(x && x_out && rows && (att == nullptr) == (att_out == nullptr), "gather_rows: null pointer");
  if (R <= 0) return DARK_OK;
  const dim3 grid(cdiv(R, 4)), block(256);
  if (dtype == DARK_BF16 && xdt == DARK_BF16)
    hipLaunchKernelGGL((gather_rows_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)att, (bf16_t*)att_out,
                       (const bf16_t*)x, (bf16_t*)x_out, rows, n_src, R, d);
  else if (dtype == DARK_BF16)
    hipLaunchKernelGGL((gather_rows_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)att, (bf16_t*)att_out,
                       (const float*)x, (float*)x_out, rows, n_src, R, d);
  else
    hipLaunchKernelGGL((gather_rows_kernel<float, float>), grid, block, 0, st, (const float*)att, (float*)att_out,
                       (const float*)x, (float*)x_out, rows, n_src, R, d);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ------------------------------------------------------------------ mask rows (generation_utils.py:489, :496, :501)
// Single workgroup stream compaction over the flattened [B*N] ids: row-major order is the order
// `logits[mask_index]` enumerates masked positions in.
__global__ __launch_bounds__(1024) void mask_rows_kernel(const int64_t* __restrict__ x, int total, int N,
                                                         int64_t mask_id, int32_t* __restrict__ row_index,
                                                         int32_t* __restrict__ src_rows,
                                                         int32_t* __restrict__ count) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = (total + 1023) / 1024;
  const int lo = min(tid * chunk, total), hi = min(lo + chunk, total);
  int cnt = 0;
  for (int f = lo; f < hi; ++f) cnt += (x[f] == mask_id);
  // inclusive scan inside the wave
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int base = 0, all = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) base += wave_tot[w];
    all += wave_tot[w];
  }
  int off = base + inc - cnt;
  for (int f = lo; f < hi; ++f) {
    if (x[f] == mask_id) {
      row_index[off] = f;
      src_rows[off] = (f % N == 0) ? f : f - 1;
      ++off;
    }
  }
  if (tid == 0) *count = all;
}

// Multi-workgroup variant for the usual sizes: workgroup w owns elements [1024w, 1024w + 1024).  Its output offset is
// the number of masks before it, which it counts itself from the (L2-resident, <= 2 MB) id array: no scratch buffer
// and no second launch; 38 -> 7 us at B*N = 23552.
__global__ __launch_bounds__(1024) void mask_rows_tiled_kernel(const int64_t* __restrict__ x, int total, int N,
                                                               int64_t mask_id, int32_t* __restrict__ row_index,
                                                               int32_t* __restrict__ src_rows,
                                                               int32_t* __restrict__ count) {
  __shared__ int wave_tot[16];
  __shared__ int before_sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int first = blockIdx.x * 1024;
  int before = 0;
  for (int f = tid; f < first; f += 1024) before += (x[f] == mask_id);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (tid == 0) before_sh = 0;
  __syncthreads();
  if (lane == 0 && before) atomicAdd(&before_sh, before);
  const int f = first + tid;
  const bool flag = f < total && x[f] == mask_id;
  const unsigned long long bal = __ballot(flag);
  if (lane == 0) wave_tot[wave] = __popcll(bal);
  __syncthreads();
  int off = before_sh, all = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) off += wave_tot[w];
    all += wave_tot[w];
  }
  if (flag) {
    off += __popcll(bal & ((1ull << lane) - 1ull));
    row_index[off] = f;
    src_rows[off] = (f % N == 0) ? f : f - 1;
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) *count = before_sh + all;
}

int launch_mask_rows(const int64_t* x, int B, int N, int64_t mask_id, int32_t* row_index, int32_t* src_rows,
                     int32_t* count, hipStream_t st) {
  DARK_CHECK_ARG(B > 0 && N > 0 && (int64_t)B * N < (1ll << 30), "mask_rows: bad B=%d N=%d", B, N);
  const int total = B * N, nb = cdiv(total, 1024);
  if (nb <= 256)
    hipLaunchKernelGGL(mask_rows_tiled_kernel, dim3(nb), dim3(1024), 0, st, x, total, N, mask_id, row_index, src_rows,
                       count);
  else
    hipLaunchKernelGGL(mask_rows_kernel, dim3(1), dim3(1024), 0, st, x, total, N, mask_id, row_index, src_rows,
                       count);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ------------------------------------------------------------------ commit (generation_utils.py:522-534)
// One wave per sequence.  full_confidence = -inf except at masked positions; the k largest
// (ties -> lower index) are transferred: x0 where masked, mask_id elsewhere (re-mask quirk).
template <int E>  // positions per lane, N <= 64*E
__global__ __launch_bounds__(256) void commit_kernel(int64_t* __restrict__ x, int B, int N, int64_t mask_id,
                                                     const int32_t* __restrict__ row_index, int M,
                                                     const float* __restrict__ conf,
                                                     const int64_t* __restrict__ x0, int k, float alg_temp,
                                                     const float* __restrict__ qnoise) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  // first m with row_index[m] >= b*N (binary search; row_index is ascending)
  const int key = b * N;
  int lo = 0, hi = M;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (row_index[mid] < key) lo = mid + 1; else hi = mid;
  }
  int m_next = lo;
  uint32_t c[E];  // order_key of the ranking value: NaN ranks first, as in torch.topk
  float cf[E];
  int64_t tok[E];
  bool msk[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = e * 64 + lane;
    const bool valid = j < N;
    const bool is_mask = valid && x[(int64_t)b * N + j] == mask_id;
    const unsigned long long bal = __ballot(is_mask);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    msk[e] = is_mask;
    cf[e] = -INFINITY;
    tok[e] = mask_id;
    if (is_mask) {
      cf[e] = conf[m_next + rank];
      tok[e] = x0[m_next + rank];
    }
    m_next += __popcll(bal);  // (positions j >= N are never selected and never counted below)
  }
  if (alg_temp > 0.f) {
    // generation_utils.py:527-530: p = softmax(full_confidence / alg_temp) over the row, then
    // torch.multinomial(p, k) WITHOUT replacement, which ATen evaluates as topk(p / q, k) with q ~ Exp(1) drawn as
    // one `exponential_` fill of [B, N] (aten/src/ATen/native/Distributions.cpp, multinomial fast path).  qnoise is
    // that fill.  Unmasked positions have p = 0 and rank last; a row with no masked position at all makes the
    // reference raise (NaN probabilities) -- here it re-masks its first k positions like the deterministic branch.
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      cf[e] = cf[e] / alg_temp;
      if (e * 64 + lane < N) mx = fmaxf(mx, cf[e]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool valid = e * 64 + lane < N;
      cf[e] = (valid && mx != -INFINITY) ? expf(cf[e] - mx) : 0.f;
      sum += cf[e];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = e * 64 + lane;
      const float p = sum > 0.f ? cf[e] / sum : 0.f;
      cf[e] = j < N ? p / qnoise[(int64_t)b * N + j] : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) c[e] = order_key(cf[e]);
  // rank of every position in (confidence desc, index asc) order
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = e * 64 + lane;
    int rank = 0;
#pragma unroll
    for (int e2 = 0; e2 < E; ++e2) {
      for (int l2 = 0; l2 < 64; ++l2) {
        const int j2 = e2 * 64 + l2;
        if (j2 >= N) break;
        const uint32_t c2 = (uint32_t)__shfl((int)c[e2], l2, 64);
        rank += (c2 > c[e]) || (c2 == c[e] && j2 < j);
      }
    }
    if (j < N && rank < k) x[(int64_t)b * N + j] = msk[e] ? tok[e] : mask_id;
  }
}

int launch_commit(int64_t* x, int B, int N, int64_t mask_id, const int32_t* row_index, int M,
                  const float* conf, const int64_t* x0, int k, hipStream_t st, float alg_temp, const float* qnoise) {
  DARK_CHECK_ARG(B > 0 && N > 0 && N <= 256, "commit: N=%d must be in [1,256]", N);
  DARK_CHECK_ARG(M >= 0 && k >= 0, "commit: bad M=%d k=%d", M, k);
  DARK_CHECK_ARG(!(alg_temp > 0.f) || qnoise, "commit: alg_temp > 0 needs the [B,N] Exp(1) draws");
  if (k == 0) return DARK_OK;
  if (k > N) k = N;
  const dim3 grid(cdiv(B, 4)), block(256);
  if (N <= 64)
    hipLaunchKernelGGL(commit_kernel<1>, grid, block, 0, st, x, B, N, mask_id, row_index, M, conf, x0, k, alg_temp, qnoise);
  else if (N <= 128)
    hipLaunchKernelGGL(commit_kernel<2>, grid, block, 0, st, x, B, N, mask_id, row_index, M, conf, x0, k, alg_temp, qnoise);
  else
    hipLaunchKernelGGL(commit_kernel<4>, grid, block, 0, st, x, B, N, mask_id, row_index, M, conf, x0, k, alg_temp, qnoise);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// alg = 'origin' (generation_utils.py:505-510): x[mask_index] = x0 where x0 is MASK except at the rows that drew
// `rand < p_transfer`; only those rows were sampled.  x[row_index[sel[i]]] = tokens[i].
__global__ __launch_bounds__(256) void scatter_tokens_kernel(int64_t* __restrict__ x, int64_t total,
                                                             const int32_t* __restrict__ row_index, int M,
                                                             const int64_t* __restrict__ sel,
                                                             const int64_t* __restrict__ tokens, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t m = sel[i];
  if (m < 0 || m >= M) return;
  const int64_t f = row_index[m];
  if (f >= 0 && f < total) x[f] = tokens[i];
}

int launch_scatter_tokens(int64_t* x, int64_t total, const int32_t* row_index, int M, const int64_t* sel,
                          const int64_t* tokens, int n, hipStream_t st) {
  if (n <= 0) return DARK_OK;
  DARK_CHECK_ARG(x && row_index && sel && tokens && M > 0, "scatter_tokens: bad argument");
  hipLaunchKernelGGL(scatter_tokens_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, total, row_index, M, sel, tokens, n);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

}  // namespace dark
