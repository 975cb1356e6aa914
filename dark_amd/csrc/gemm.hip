// MFMA GEMMs of the denoise forward: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues.
//
// Replaces the nn.Linear calls of modeling_dream_length.py:381-383 (q/k/v + bias), :436 (o_proj,
// + residual add :508), :221-222 (gate/up + SiLU*up, down + residual add :514) and :809 (lm_head).
//
// Both operands are K-contiguous ("B^T" layout = how nn.Linear stores weights), so A and W tiles
// are staged identically.  Block tile 128x128, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles.
//   bf16: v_mfma_f32_32x32x16_bf16, BK = 64, LDS rows of 128 B with a 16-B-chunk XOR swizzle
//         (chunk ^= row & 7) so the ds_read_b128 fragment reads spread over the bank row.
//   fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, k ascending), BK = 16, LDS rows padded to 17.
// Double-buffered LDS, global loads of tile t+1 issued before the MFMAs of tile t (register staging).
#include <utility>
#include <vector>

#include "common.h"

namespace dark {

constexpr int BM = 128, BN = 128;

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + expf(-g)); }

// acc[mi][ni][r]: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 inside a 32x32 tile
template <int EPI, typename T>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[2][2], const float* __restrict__ bias,
                                         void* __restrict__ out, int M, int N, int64_t ldo, int m0,
                                         int n0, int tn, int wm, int wn, int lane) {
  const int lr = 4 * (lane >> 5), lc = lane & 31;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + lr;
      if (m >= M) continue;
      if constexpr (EPI == EPI_SWIGLU) {
        const int c = tn * 64 + wn * 32 + lc;
        if (c < (N >> 1)) {
          const float g = acc[mi][0][r], u = acc[mi][1][r];
          ((T*)out)[(int64_t)m * ldo + c] = ElemTraits<T>::from_f32(silu_f(g) * u);
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int n = n0 + wn * 64 + ni * 32 + lc;
          if (n >= N) continue;
          const float v = acc[mi][ni][r];
          if constexpr (EPI == EPI_BIAS) {
            ((T*)out)[(int64_t)m * ldo + n] = ElemTraits<T>::from_f32(v + bias[n]);
          } else if constexpr (EPI == EPI_RESID) {
            float* p = (float*)out + (int64_t)m * ldo + n;
            *p = *p + v;
          } else {
            ((float*)out)[(int64_t)m * ldo + n] = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ bf16
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ A,
                                                           const bf16_t* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           void* __restrict__ out, int M, int N, int K,
                                                           int64_t ldo, int tilesN, int nwg) {
  constexpr int BK = 64;
  constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];  // A0 A1 B0 B1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int tn = wg % tilesN, tm = wg / tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging map: chunk c = tid + 256*p -> row = c>>3 (0..127), 16-B chunk kc = c&7
  const int srow = tid >> 3, kc = tid & 7;
  const int lds_off = srow * 128 + ((kc ^ (srow & 7)) << 4);  // + p*32*128
  const bf16_t* ga[4];
  const bf16_t* gw[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int ar = m0 + srow + 32 * p;
    ar = ar < M ? ar : M - 1;
    ga[p] = A + (int64_t)ar * K + kc * 8;
    gw[p] = W + (int64_t)(n0 + srow + 32 * p) * K + kc * 8;
  }
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define DARK_GLOAD(kt)                                  \
  do {                                                  \
    ra0 = *(const uint4*)(ga[0] + (kt) * BK);           \
    ra1 = *(const uint4*)(ga[1] + (kt) * BK);           \
    ra2 = *(const uint4*)(ga[2] + (kt) * BK);           \
    ra3 = *(const uint4*)(ga[3] + (kt) * BK);           \
    rb0 = *(const uint4*)(gw[0] + (kt) * BK);           \
    rb1 = *(const uint4*)(gw[1] + (kt) * BK);           \
    rb2 = *(const uint4*)(gw[2] + (kt) * BK);           \
    rb3 = *(const uint4*)(gw[3] + (kt) * BK);           \
  } while (0)
#define DARK_LSTORE(buf)                                           \
  do {                                                             \
    char* as_ = smem + (buf) * TILE_BYTES + lds_off;               \
    char* bs_ = smem + 2 * TILE_BYTES + (buf) * TILE_BYTES + lds_off; \
    *(uint4*)(as_) = ra0;                                          \
    *(uint4*)(as_ + 1 * 32 * 128) = ra1;                           \
    *(uint4*)(as_ + 2 * 32 * 128) = ra2;                           \
    *(uint4*)(as_ + 3 * 32 * 128) = ra3;                           \
    *(uint4*)(bs_) = rb0;                                          \
    *(uint4*)(bs_ + 1 * 32 * 128) = rb1;                           \
    *(uint4*)(bs_ + 2 * 32 * 128) = rb2;                           \
    *(uint4*)(bs_ + 3 * 32 * 128) = rb3;                           \
  } while (0)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = K / BK;
  const int frow = lane & 31, fh = lane >> 5;
  auto compute = [&](int cur) {
    const char* as = smem + cur * TILE_BYTES + (wm * 64 + frow) * 128;
    const char* bs = smem + 2 * TILE_BYTES + cur * TILE_BYTES + (wn * 64 + frow) * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = ((2 * kk + fh) ^ (frow & 7)) << 4;
      const bf16x8 a0 = *(const bf16x8*)(as + ch);
      const bf16x8 a1 = *(const bf16x8*)(as + 32 * 128 + ch);
      const bf16x8 b0 = *(const bf16x8*)(bs + ch);
      const bf16x8 b1 = *(const bf16x8*)(bs + 32 * 128 + ch);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  DARK_GLOAD(0);
  DARK_LSTORE(0);
  __syncthreads();
  for (int kt = 0; kt < KT - 1; ++kt) {
    const int cur = kt & 1;
    DARK_GLOAD(kt + 1);
    compute(cur);
    DARK_LSTORE(cur ^ 1);
    __syncthreads();
  }
  compute((KT - 1) & 1);
#undef DARK_GLOAD
#undef DARK_LSTORE
  epilogue<EPI, bf16_t>(acc, bias, out, M, N, ldo, m0, n0, tn, wm, wn, lane);
}

// ------------------------------------------------------------------ fp32
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float* __restrict__ A,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          void* __restrict__ out, int M, int N, int K,
                                                          int64_t ldo, int tilesN, int nwg) {
  constexpr int BK = 16, LDR = BK + 1;
  __shared__ float As[2][BM * LDR];
  __shared__ float Bs[2][BN * LDR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int tn = wg % tilesN, tm = wg / tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging map: chunk c = tid + 256*p -> row = c>>2 (0..127), float4 chunk kc = c&3
  const int srow = tid >> 2, kc = tid & 3;
  const float* ga[2];
  const float* gw[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int ar = m0 + srow + 64 * p;
    ar = ar < M ? ar : M - 1;
    ga[p] = A + (int64_t)ar * K + kc * 4;
    gw[p] = W + (int64_t)(n0 + srow + 64 * p) * K + kc * 4;
  }
  float4 ra[2], rb[2];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      ra[p] = *(const float4*)(ga[p] + kt * BK);
      rb[p] = *(const float4*)(gw[p] + kt * BK);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float* a = &As[buf][(srow + 64 * p) * LDR + kc * 4];
      a[0] = ra[p].x; a[1] = ra[p].y; a[2] = ra[p].z; a[3] = ra[p].w;
      float* b = &Bs[buf][(srow + 64 * p) * LDR + kc * 4];
      b[0] = rb[p].x; b[1] = rb[p].y; b[2] = rb[p].z; b[3] = rb[p].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int frow = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const float* as = &As[cur][(wm * 64 + frow) * LDR + fh];
    const float* bs = &Bs[cur][(wn * 64 + frow) * LDR + fh];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float a[2], b[2];
      a[0] = as[2 * ks];
      a[1] = as[32 * LDR + 2 * ks];
      b[0] = bs[2 * ks];
      b[1] = bs[32 * LDR + 2 * ks];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < KT) lstore(cur ^ 1);
    __syncthreads();
  }
  epilogue<EPI, float>(acc, bias, out, M, N, ldo, m0, n0, tn, wm, wn, lane);
}

// ------------------------------------------------------------------ launch
template <int EPI>
static int launch_epi(int dtype, const void* A, const void* W, const float* bias, void* out, int M, int N,
                      int K, int64_t ldo, hipStream_t st) {
  const int tilesM = cdiv(M, BM), tilesN = cdiv(N, BN);
  const int nwg = tilesM * tilesN;
  if (dtype == DARK_BF16) {
    constexpr int smem = 4 * BM * 64 * 2;  // 64 KiB
    static bool attr_done = false;
    if (!attr_done) {
      DARK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      attr_done = true;
    }
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(nwg), dim3(256), smem, st, (const bf16_t*)A,
                       (const bf16_t*)W, bias, out, M, N, K, ldo, tilesN, nwg);
  } else {
    hipLaunchKernelGGL(gemm_f32_kernel<EPI>, dim3(nwg), dim3(256), 0, st, (const float*)A,
                       (const float*)W, bias, out, M, N, K, ldo, tilesN, nwg);
  }
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ---- optional HIP-event timing of one GEMM flavour (bench.py roofline leg) ----
namespace {
struct Prof {
  int epi = -1;  // -1: off
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t used = 0;
  double flops = 0.0;
} g_prof;
}  // namespace

int prof_select(int epi) {
  g_prof.epi = epi;
  g_prof.used = 0;
  g_prof.flops = 0.0;
  return DARK_OK;
}

int prof_read(double* total_ms, double* total_flops, int* launches) {
  double ms = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    DARK_CHECK_HIP(hipEventSynchronize(g_prof.ev[i].second));
    float t = 0.f;
    DARK_CHECK_HIP(hipEventElapsedTime(&t, g_prof.ev[i].first, g_prof.ev[i].second));
    ms += t;
  }
  *total_ms = ms;
  *total_flops = g_prof.flops;
  *launches = (int)g_prof.used;
  g_prof.used = 0;
  g_prof.flops = 0.0;
  return DARK_OK;
}

static int launch_gemm_impl(int dtype, int epi, const void* A, const void* W, const float* bias, void* out, int M,
                            int N, int K, int64_t ldo, hipStream_t st);

int launch_gemm(int dtype, int epi, const void* A, const void* W, const float* bias, void* out, int M,
                int N, int K, int64_t ldo, hipStream_t st) {
  if (g_prof.epi != epi || g_prof.used >= 8192) return launch_gemm_impl(dtype, epi, A, W, bias, out, M, N, K, ldo, st);
  if (g_prof.used == g_prof.ev.size()) {
    hipEvent_t a, b;
    DARK_CHECK_HIP(hipEventCreate(&a));
    DARK_CHECK_HIP(hipEventCreate(&b));
    g_prof.ev.emplace_back(a, b);
  }
  auto& e = g_prof.ev[g_prof.used];
  DARK_CHECK_HIP(hipEventRecord(e.first, st));
  const int rc = launch_gemm_impl(dtype, epi, A, W, bias, out, M, N, K, ldo, st);
  DARK_CHECK_HIP(hipEventRecord(e.second, st));
  g_prof.used++;
  g_prof.flops += 2.0 * M * (double)N * K;
  return rc;
}

static int launch_gemm_impl(int dtype, int epi, const void* A, const void* W, const float* bias, void* out, int M,
                            int N, int K, int64_t ldo, hipStream_t st) {
  DARK_CHECK_ARG(dtype == DARK_F32 || dtype == DARK_BF16, "gemm: bad dtype %d", dtype);
  DARK_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  DARK_CHECK_ARG(K % 64 == 0, "gemm: K=%d must be a multiple of 64", K);
  DARK_CHECK_ARG(A && W && out, "gemm: null operand");
  switch (epi) {
    case EPI_BIAS:
      DARK_CHECK_ARG(bias != nullptr, "gemm: EPI_BIAS needs a bias");
      return launch_epi<EPI_BIAS>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_RESID: return launch_epi<EPI_RESID>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_SWIGLU:
      DARK_CHECK_ARG(N % 64 == 0, "gemm: SWIGLU needs N %% 64 == 0 (N=%d)", N);
      return launch_epi<EPI_SWIGLU>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_F32: return launch_epi<EPI_F32>(dtype, A, W, bias, out, M, N, K, ldo, st);
  }
  set_error("gemm: bad epilogue %d", epi);
  return DARK_ERR_ARG;
}

}  // namespace dark
