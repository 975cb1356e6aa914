// MFMA GEMMs of the denoise forward: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues.
//
// Replaces the nn.Linear calls of modeling_dream_length.py:381-383 (q/k/v + bias), :436 (o_proj,
// + residual add :508), :221-222 (gate/up + SiLU*up, down + residual add :514) and :809 (lm_head).
//
// Both operands are K-contiguous ("B^T" layout = how nn.Linear stores weights), so A and W tiles are staged
// identically.  Kernels of this file:
//   bf16: gemm_bf16_kernel16<EPI, BNt> -- 128 x BNt block tile (BNt = 192, or 128 for the fp32-out flavours), 4 waves
//         (2 x 2) of v_mfma_f32_16x16x32_bf16 tiles, BK = 64, two LDS stages filled by LDS-DMA, two workgroups per CU.
//         Serves o_proj + residual (and q/k/v + bias / down_proj + residual where the persistent kernel does not
//         take them: fp32 residual stream, few rows), the log-sum-exp head of the GRPO path, the GPT-2 epilogues, and every
//         launch too small for the persistent 256-wide kernels of gemm256.hip (pick_kernel below decides).  The
//         residual flavours exist for an fp32 stream (EPI_RESID, EPI_RESID_BIAS) and for a bf16 one (.._H).
//   fp32: gemm_f32_kernel<EPI> -- 128 x 128 tile, 4 waves of 2 x 2 v_mfma_f32_32x32x2_f32 tiles (an exact fp32 fma chain
//         per output, k ascending: what the bit-exact parity tests against the CPU oracle rely on), BK = 16, LDS rows
//         padded to 17 floats, register-staged double buffering.
// launch_gemm() is the single entry; the HIP-event timing hook of bench.py's roofline leg wraps it.

#include <utility>
#include <vector>

#include "common.h"
#include <mutex>

namespace dark {

constexpr int BM = 128, BN = 128;

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + expf(-g)); }
// bf16 epilogues: hardware exp2 / rcp (1 ulp each), far below the bf16 rounding of the result
__device__ __forceinline__ float silu_fast(float g) {
  return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.44269504088896340736f));
}

// GPT-2 backbone (custom_gpt.py:25-114 -> transformers' GPT2MLP): NewGELUActivation
__device__ __forceinline__ float gelu_new_f(float x) {
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
constexpr bool epi_has_bias(int epi) { return epi == EPI_BIAS || epi == EPI_BIAS_GELU; }
constexpr bool epi_is_resid(int epi) { return epi == EPI_RESID || epi == EPI_RESID_BIAS; }
// the same two with the residual stream held in bf16 (bf16 models, dark_model_desc.resid_dtype = DARK_BF16)
constexpr bool epi_is_resid_h(int epi) { return epi == EPI_RESID_H || epi == EPI_RESID_BIAS_H; }

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
          } else if constexpr (EPI == EPI_BIAS_GELU) {
            ((T*)out)[(int64_t)m * ldo + n] = ElemTraits<T>::from_f32(gelu_new_f(v + bias[n]));
          } else if constexpr (EPI == EPI_RESID) {
            float* p = (float*)out + (int64_t)m * ldo + n;
            *p = *p + v;
          } else if constexpr (EPI == EPI_RESID_BIAS) {
            float* p = (float*)out + (int64_t)m * ldo + n;
            *p = *p + (v + bias[n]);
          } else {
            ((float*)out)[(int64_t)m * ldo + n] = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ bf16
// Operands are SWAPPED in the MFMA (srcA = weight fragment, srcB = activation fragment), so the
// accumulator holds D[n][m]: a lane owns ONE output row m = lane & 15 and, per 16x16 tile, the four
// consecutive columns 4*(lane >> 4) + e, which makes the epilogue a row-per-lane store of 16-B pieces (bf16
// after one permlane16_swap per dword, fp32 directly) instead of 2-byte column-strided stores.
//
// Staging: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip).  The LDS image of an operand tile is
// lane-linear ([rows][8 chunks of 16 B]); the XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free (physical chunk = logical chunk ^ ((row >> 1) & 7): 16 rows of a read group land on the 16
// distinct 16-B slots of the 256-B bank row) is applied to the per-lane SOURCE address.
// Pipeline: 2 LDS stages, one barrier per K tile; the DMA of tile t+1 is in flight during the MFMAs of
// tile t; <= 80 KiB LDS -> 2 workgroups per CU, so the other workgroup's MFMAs cover this one's barrier.
//
// Tried and dropped (measured on MI355X): a 32x32x16-MFMA twin of this kernel (4-8 % slower
// under sustained load), a 256x256 4-stage LDS-DMA ring with counted vmcnt (no gain at K = 768), and folding
// the RMSNorms into the GEMM epilogues (the GEMMs slowed down by more than the norm passes cost).
#define DARK_GLDS16(gsrc, ldst)                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),         \
                                   (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}

// Two horizontally adjacent 16x16 tiles of final values of this lane (row m; tile t holds columns
// cb + 16t + 4q + e, q = lane >> 4) stored as bf16 in 16-B pieces: one permlane16_swap per dword moves the
// odd 16-lane rows' first-tile data to their even neighbours and the even rows' second-tile data to the odd
// ones, so even rows own columns 4q..4q+7 of the first tile and odd rows 4(q-1)..4(q-1)+7 of the second.
__device__ __forceinline__ void store_pair_bf16(const float (&v0)[4], const float (&v1)[4], bf16_t* __restrict__ orow,
                                                int cb, int ncols, int q, bool vec_ok) {
  const uint32_t a_lo = pack_bf16x2(v0[0], v0[1]), a_hi = pack_bf16x2(v0[2], v0[3]);
  const uint32_t b_lo = pack_bf16x2(v1[0], v1[1]), b_hi = pack_bf16x2(v1[2], v1[3]);
  if (vec_ok) {
    const auto sx = __builtin_amdgcn_permlane16_swap(a_lo, b_lo, false, false);
    const auto sy = __builtin_amdgcn_permlane16_swap(a_hi, b_hi, false, false);
    const int c = cb + ((q & 1) ? 16 + 4 * (q - 1) : 4 * q);
    if (c + 8 <= ncols) *(uint4*)(orow + c) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
  } else {
    const uint32_t w[4] = {a_lo, a_hi, b_lo, b_hi};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cb + 16 * t + 4 * q + e;
        if (c < ncols) ((uint16_t*)orow)[c] = (uint16_t)(w[2 * t + (e >> 1)] >> (16 * (e & 1)));
      }
  }
}

// ------------------------------------------------------------------ bf16, 16x16x32 MFMA shape
// 128 x BNt block tile, 4 waves (2x2), each wave 64 x BNt/2 = 4 x TNW tiles of v_mfma_f32_16x16x32_bf16.
// BNt = 128, or 192 for the N = 768 GEMMs: 184 x 4 tiles fill 512 workgroup slots 2.9 times instead of 4.3
// (96 % instead of 86 % tile quantisation at M = 23552) and cut the fragment reads per MFMA from 0.5 to 0.42.
template <int EPI, int BNt>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel16(const bf16_t* __restrict__ A,
                                                             const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias,
                                                             void* __restrict__ out, int M, int N, int K,
                                                             int64_t ldo, int tilesM, int tilesN, int nwg,
                                                             int vec_ok) {
  constexpr int BK = 64;
  constexpr int TILE = BM * BK * 2;              // activation tile bytes
  constexpr int WTILE = BNt * BK * 2;            // weight tile bytes
  constexpr int STAGE = TILE + WTILE;
  constexpr int WN = BNt / 2;                    // columns per wave
  constexpr int TNW = WN / 16;                   // 16-column MFMA tiles per wave
  constexpr int WP = BNt / 32;                   // 8-row weight pieces staged per wave
  constexpr int GM = 8;  // row panels per raster group (swept 2..32: flat within 2 % between 4 and 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int per_group = GM * tilesN;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gsz = min(GM, tilesM - first_m);
  const int rem = wg - group * per_group;
  const int tn = rem / gsz, tm = first_m + (rem - tn * gsz);
  const int m0 = tm * BM, n0 = tn * BNt;

  const bf16_t* ga[4];
  const bf16_t* gw[WP];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 32 + q * 8 + (lane >> 3);
    const int lch = (lane & 7) ^ ((row >> 1) & 7);
    ga[q] = A + (int64_t)min(m0 + row, M - 1) * K + lch * 8;
  }
#pragma unroll
  for (int q = 0; q < WP; ++q) {
    const int row = wave * (WP * 8) + q * 8 + (lane >> 3);
    const int lch = (lane & 7) ^ ((row >> 1) & 7);
    gw[q] = W + (int64_t)min(n0 + row, N - 1) * K + lch * 8;  // rows >= N only feed columns never stored
  }
  auto stage = [&](int buf, int kt) {
    char* a = smem + buf * STAGE + wave * 4096;
    char* w = smem + buf * STAGE + TILE + wave * (WP * 1024);
#pragma unroll
    for (int q = 0; q < 4; ++q) DARK_GLDS16(ga[q] + kt * BK, a + q * 1024);
#pragma unroll
    for (int q = 0; q < WP; ++q) DARK_GLDS16(gw[q] + kt * BK, w + q * 1024);
  };

  f32x4 acc[TNW][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < TNW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int KT = K / BK;
  const int frow = lane & 15, fq = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  auto compute = [&](int buf) {
    const char* as = smem + buf * STAGE + (wm * 64 + frow) * 128;
    const char* ws = smem + buf * STAGE + TILE + (wn * WN + frow) * 128;
    // All fragment reads of the stage (both 32-wide k halves) are issued before the first MFMA: one exposed LDS
    // latency per stage instead of one in front of every group of 8 MFMAs (+4 % on qkv and gate/up).  The residual
    // flavour keeps the compiler's interleaved order: with its prefetched residual pieces the 80 fragment registers
    // would spill.
    if constexpr (epi_is_resid(EPI) || EPI == EPI_RESID_BIAS_H) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((4 * ks + fq) ^ fsw) << 4;
        bf16x8 af[4], wf[TNW];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(as + i * 16 * 128 + ch);
#pragma unroll
        for (int j = 0; j < TNW; ++j) wf[j] = *(const bf16x8*)(ws + j * 16 * 128 + ch);
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
      }
      return;
    }
    bf16x8 af[2][4], wf[2][TNW];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((4 * ks + fq) ^ fsw) << 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = *(const bf16x8*)(as + i * 16 * 128 + ch);
#pragma unroll
      for (int j = 0; j < TNW; ++j) wf[ks][j] = *(const bf16x8*)(ws + j * 16 * 128 + ch);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[ks][i], acc[j][i], 0, 0, 0);
  };

  // Residual flavour, 16-byte path: the epilogue is a read-modify-write of the fp32 residual, 24 pieces of 16 bytes
  // a lane.  Left to the compiler every piece is load -> wait -> add -> store (it cannot tell the rows apart), 24
  // dependent memory round trips a lane, which at K = 768 outlast the main loop.  Here the old values of the
  // first two 16-row groups are fetched before the main loop (they ride under the first stage's wait) and those of the
  // other two in one batch at the top of the epilogue.
  float* ob = nullptr;
  float4 old0[2][TNW];
  auto resid_ok = [&](int mi, int ni) {
    return m0 + wm * 64 + mi * 16 + frow < M && n0 + wn * WN + 4 * fq + ni * 16 + 4 <= N;
  };
  if constexpr (epi_is_resid(EPI)) {
    if (vec_ok) {
      ob = (float*)out + (int64_t)(m0 + wm * 64 + frow) * ldo + n0 + wn * WN + 4 * fq;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < TNW; ++ni)
          if (resid_ok(mi, ni)) old0[mi][ni] = *(const float4*)(ob + (int64_t)mi * 16 * ldo + ni * 16);
    }
  }

  // bf16 residual stream: the same read-modify-write on 16-byte pieces of EIGHT bf16 values.  A piece is what
  // store_pair_bf16 writes: for a pair of column tiles (2p, 2p + 1) even 16-lane rows own columns 4q .. 4q+7 of the
  // first tile, odd rows 4(q-1) .. 4(q-1)+7 of the second; the fp32 accumulators are brought into that layout with one
  // permlane16_swap per register, the old values are added in fp32 and the sum is rounded once.  12 pieces a lane
  // (48 registers) instead of 24 (96): half the bytes of the fp32 stream each way.
  constexpr int TP = TNW / 2;
  bf16_t* obh = nullptr;
  uint4 oldh0[2][TP], oldh1[2][TP];
  const int hcol = (fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq;   // first column of this lane's piece inside a tile pair
  auto resid_h_ok = [&](int mi, int p) {
    return m0 + wm * 64 + mi * 16 + frow < M && n0 + wn * WN + 32 * p + hcol + 8 <= N;
  };
  if constexpr (epi_is_resid_h(EPI)) {
    static_assert(TNW % 2 == 0, "bf16 residual epilogue works on pairs of column tiles");
    if (vec_ok) {
      obh = (bf16_t*)out + (int64_t)(m0 + wm * 64 + frow) * ldo + n0 + wn * WN + hcol;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int p = 0; p < TP; ++p)
          if (resid_h_ok(mi, p)) oldh0[mi][p] = *(const uint4*)(obh + (int64_t)mi * 16 * ldo + 32 * p);
    }
  }

  // Bias flavour: the TNW bias pieces of this lane's columns are fetched here, once, and ride under the first
  // stage's wait; in the epilogue they would be a dependent load in front of every store (24 a lane).
  float4 b4[TNW];
  if constexpr (epi_has_bias(EPI) || EPI == EPI_RESID_BIAS || EPI == EPI_RESID_BIAS_H) {
#pragma unroll
    for (int ni = 0; ni < TNW; ++ni) {
      const int c = n0 + wn * WN + ni * 16 + 4 * fq;
      if (vec_ok && c + 4 <= N) {
        b4[ni] = *(const float4*)(bias + c);
      } else {
        b4[ni].x = c + 0 < N ? bias[c + 0] : 0.f; b4[ni].y = c + 1 < N ? bias[c + 1] : 0.f;
        b4[ni].z = c + 2 < N ? bias[c + 2] : 0.f; b4[ni].w = c + 3 < N ? bias[c + 3] : 0.f;
      }
    }
  }

  float4 old1[2][TNW];
  stage(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) stage((kt + 1) & 1, kt + 1);
    if constexpr (epi_is_resid(EPI)) {
      if (vec_ok && kt == KT - 1) {  // the other two row groups ride under the last compute step
#pragma unroll
        for (int mi = 2; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < TNW; ++ni)
            if (resid_ok(mi, ni)) old1[mi - 2][ni] = *(const float4*)(ob + (int64_t)mi * 16 * ldo + ni * 16);
      }
    }
    if constexpr (epi_is_resid_h(EPI)) {
      if (vec_ok && kt == KT - 1) {
#pragma unroll
        for (int mi = 2; mi < 4; ++mi)
#pragma unroll
          for (int p = 0; p < TP; ++p)
            if (resid_h_ok(mi, p)) oldh1[mi - 2][p] = *(const uint4*)(obh + (int64_t)mi * 16 * ldo + 32 * p);
      }
    }
    compute(kt & 1);
  }

  // bf16 residual stream, 16-byte path (see the prefetch above the main loop).  The swaps run in every lane.
  if constexpr (epi_is_resid_h(EPI)) {
    if (vec_ok) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          f32x4 a = acc[2 * p][mi], b = acc[2 * p + 1][mi];
          if constexpr (EPI == EPI_RESID_BIAS_H) {  // hidden + (acc + bias), as the fp32 flavour
            a[0] += b4[2 * p].x; a[1] += b4[2 * p].y; a[2] += b4[2 * p].z; a[3] += b4[2 * p].w;
            b[0] += b4[2 * p + 1].x; b[1] += b4[2 * p + 1].y; b[2] += b4[2 * p + 1].z; b[3] += b4[2 * p + 1].w;
          }
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
          }
          if (resid_h_ok(mi, p)) {
            const uint4 o = mi < 2 ? oldh0[mi & 1][p] : oldh1[mi & 1][p];
            const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
            uint32_t r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              r[j] = pack_bf16x2(v[2 * j] + __uint_as_float(ow[j] << 16), v[2 * j + 1] + __uint_as_float(ow[j] & 0xffff0000u));
            *(uint4*)(obh + (int64_t)mi * 16 * ldo + 32 * p) = make_uint4(r[0], r[1], r[2], r[3]);
          }
        }
    }
  }

  // Residual flavour, 16-byte path: second half of the read-modify-write (see the prefetch above the main loop).
  if constexpr (epi_is_resid(EPI)) {
    if (vec_ok) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < TNW; ++ni)
          if (resid_ok(mi, ni)) {
            const float4 p = mi < 2 ? old0[mi & 1][ni] : old1[mi & 1][ni];
            f32x4 a = acc[ni][mi];
            if constexpr (EPI == EPI_RESID_BIAS) {  // hidden + (acc + bias): the order of c_proj's addmm, then the residual add
              a[0] += b4[ni].x; a[1] += b4[ni].y; a[2] += b4[ni].z; a[3] += b4[ni].w;
            }
            *(float4*)(ob + (int64_t)mi * 16 * ldo + ni * 16) = make_float4(a[0] + p.x, a[1] + p.y, a[2] + p.z, a[3] + p.w);
          }
    }
  }

  // rows beyond M are computed on clamped operands and dropped here; the permlane swaps run in every lane
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + frow;
    const bool row_ok = m < M;
    const int64_t mrow = row_ok ? m : 0;
    if constexpr (EPI == EPI_SWIGLU) {
      // bf16 packing interleaves gate/up rows in groups of 16: MFMA tiles 2p / 2p+1 are gate / up of the
      // 16 output columns (n0 + wn*WN)/2 + 16p ..
      constexpr int TO = TNW / 2;
      float v[TO][4];
#pragma unroll
      for (int p = 0; p < TO; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[p][e] = silu_fast(acc[2 * p][mi][e]) * acc[2 * p + 1][mi][e];
      bf16_t* orow = (bf16_t*)out + mrow * ldo;
      const int cb = (n0 + wn * WN) >> 1, nout = row_ok ? (N >> 1) : 0;
#pragma unroll
      for (int p = 0; p + 1 < TO; p += 2) store_pair_bf16(v[p], v[p + 1], orow, cb + 16 * p, nout, fq, vec_ok);
      if constexpr (TO & 1) {
        const int c = cb + 16 * (TO - 1) + 4 * fq;
        if (vec_ok && c + 4 <= nout) {
          *(uint2*)(orow + c) = make_uint2(pack_bf16x2(v[TO - 1][0], v[TO - 1][1]), pack_bf16x2(v[TO - 1][2], v[TO - 1][3]));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < nout) orow[c + e] = (bf16_t)v[TO - 1][e];
        }
      }
    } else if constexpr (epi_has_bias(EPI)) {
#pragma unroll
      for (int np = 0; np < TNW / 2; ++np) {
        float v[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ni = 2 * np + t;
          v[t][0] = acc[ni][mi][0] + b4[ni].x; v[t][1] = acc[ni][mi][1] + b4[ni].y;
          v[t][2] = acc[ni][mi][2] + b4[ni].z; v[t][3] = acc[ni][mi][3] + b4[ni].w;
          if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t][e] = gelu_new_f(v[t][e]);
          }
        }
        store_pair_bf16(v[0], v[1], (bf16_t*)out + mrow * ldo, n0 + wn * WN + np * 32, row_ok ? N : 0, fq, vec_ok);
      }
    } else if constexpr (EPI == EPI_LSE) {
      // log-sum-exp partial of this wave's WN columns for row m: the 16 columns of an MFMA tile sit in the four
      // lanes frow, frow+16, frow+32, frow+48 (4 registers each).  out = float2 part[M][ldo] then float tgt[M];
      // `bias` carries the int64 target ids.
      const int cb = n0 + wn * WN + 4 * fq;
      float mx = -INFINITY;
#pragma unroll
      for (int ni = 0; ni < TNW; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb + ni * 16 + e < N) mx = fmaxf(mx, acc[ni][mi][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int ni = 0; ni < TNW; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb + ni * 16 + e < N) sum += __builtin_amdgcn_exp2f((acc[ni][mi][e] - mx) * 1.44269504088896341f);
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (row_ok) {
        if (fq == 0) ((float2*)out)[mrow * ldo + tn * 2 + wn] = make_float2(mx, sum);
        const int64_t t = ((const int64_t*)bias)[mrow];
        const int tc = (int)(t < 0 ? 0 : (t >= N ? N - 1 : t)) - cb;  // column offset inside this lane's pieces
        if (tc >= 0 && tc < TNW * 16 && (tc & 15) < 4) {
          float v = 0.f;
#pragma unroll
          for (int ni = 0; ni < TNW; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (tc == ni * 16 + e) v = acc[ni][mi][e];
          ((float*)((float2*)out + (int64_t)M * ldo))[mrow] = v;
        }
      }
    } else if constexpr (epi_is_resid_h(EPI)) {
      if (!row_ok || vec_ok) continue;  // (16-byte path: done above, in batches)
      bf16_t* orow = (bf16_t*)out + mrow * ldo;
#pragma unroll
      for (int ni = 0; ni < TNW; ++ni) {
        const int c = n0 + wn * WN + ni * 16 + 4 * fq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (c + e >= N) continue;
          float a = acc[ni][mi][e];
          if constexpr (EPI == EPI_RESID_BIAS_H) a += bias[c + e];
          orow[c + e] = (bf16_t)((float)orow[c + e] + a);
        }
      }
    } else {
      if (!row_ok) continue;
      if constexpr (epi_is_resid(EPI)) {
        if (vec_ok) continue;  // done above, in batches
      }
      float* orow = (float*)out + mrow * ldo;
#pragma unroll
      for (int ni = 0; ni < TNW; ++ni) {
        const int c = n0 + wn * WN + ni * 16 + 4 * fq;
        if (vec_ok && c + 4 <= N) {  // (EPI_F32; the residual flavour with vec_ok never gets here)
          *(float4*)(orow + c) = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (c + e >= N) continue;
            if constexpr (EPI == EPI_RESID) orow[c + e] += acc[ni][mi][e];
            else if constexpr (EPI == EPI_RESID_BIAS) orow[c + e] += acc[ni][mi][e] + bias[c + e];
            else orow[c + e] = acc[ni][mi][e];
          }
        }
      }
    }
  }
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
template <int EPI, int BNt>
static int launch_k16(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int64_t ldo,
                      int vec_ok, hipStream_t st) {
  constexpr int smem = 2 * (BM + BNt) * 64 * 2;
  const int tilesM = cdiv(M, BM), tilesN = cdiv(N, BNt);
  const int nwg = tilesM * tilesN;
  static SmemOptIn optin;
  DARK_TRY(optin.ensure((const void*)gemm_bf16_kernel16<EPI, BNt>, smem, st));
  hipLaunchKernelGGL((gemm_bf16_kernel16<EPI, BNt>), dim3(nwg), dim3(256), smem, st, (const bf16_t*)A,
                     (const bf16_t*)W, bias, out, M, N, K, ldo, tilesM, tilesN, nwg, vec_ok);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

template <int EPI>
static int launch_epi(int dtype, const void* A, const void* W, const float* bias, void* out, int M, int N,
                      int K, int64_t ldo, hipStream_t st) {
  const int tilesM = cdiv(M, BM), tilesN = cdiv(N, BN);
  const int nwg = tilesM * tilesN;
  if (dtype == DARK_BF16) {
    // 16-B row pieces need 16-B aligned rows: bf16 outputs 8 elements, fp32 outputs 4
    const int per16 = (epi_has_bias(EPI) || EPI == EPI_SWIGLU || epi_is_resid_h(EPI)) ? 8 : 4;
    const int ncols = EPI == EPI_SWIGLU ? N / 2 : N;
    const int vec_ok = (ldo % per16 == 0) && (ncols % per16 == 0) &&
                       ((uintptr_t)out % 16 == 0) && (bias == nullptr || (uintptr_t)bias % 16 == 0);
    // 128 x 192 tiles: fewer fragment reads per MFMA than 128 x 128 and, at N = 768, 736 instead of 1104
    // workgroups over the 512 slots; measured +7..9 % on the residual GEMMs, +5 % on gate/up and qkv
    // (tools/gemm_check.py --iters 2000).  Edge tiles clamp their operand rows, so any N works.
    // The fp32-out flavour (LM head, N = 15080) measures 6 % faster on 128-wide tiles.
    if constexpr (EPI == EPI_F32 || EPI == EPI_LSE) return launch_k16<EPI, 128>(A, W, bias, out, M, N, K, ldo, vec_ok, st);
    else return launch_k16<EPI, 192>(A, W, bias, out, M, N, K, ldo, vec_ok, st);
  } else {
    if constexpr (EPI == EPI_LSE || epi_is_resid_h(EPI)) {
      set_error("gemm: epilogue %d exists for bf16 only", EPI);
      return DARK_ERR_UNSUPPORTED;
    } else {
      hipLaunchKernelGGL(gemm_f32_kernel<EPI>, dim3(nwg), dim3(256), 0, st, (const float*)A,
                         (const float*)W, bias, out, M, N, K, ldo, tilesN, nwg);
    }
  }
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ---- optional HIP-event timing of one GEMM flavour (bench.py roofline leg) ----
namespace {
struct Prof {
  std::atomic<int> epi{-1};  // -1: off
  int rows = 0;  // > 0: only launches with exactly this many rows
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t used = 0;
  double flops = 0.0;
} g_prof;
std::mutex g_prof_mu;   // launches from several host threads may bracket themselves concurrently
}  // namespace

int prof_select(int epi, int rows) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.epi = epi;
  g_prof.rows = rows;
  g_prof.used = 0;
  g_prof.flops = 0.0;
  return DARK_OK;
}

int prof_read(double* total_ms, double* total_flops, int* launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
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
  if (g_prof.epi < 0) return launch_gemm_impl(dtype, epi, A, W, bias, out, M, N, K, ldo, st);   // hook off: no lock
  std::lock_guard<std::mutex> lk(g_prof_mu);
  // (at most PROF_MAX_LAUNCHES launches are bracketed per read; later ones run unbracketed and are not in the average)
  if (g_prof.epi != epi || g_prof.used >= 8192 || (g_prof.rows > 0 && g_prof.rows != M)) return launch_gemm_impl(dtype, epi, A, W, bias, out, M, N, K, ldo, st);
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

// 0 = automatic choice, 1 = always the 128-row kernels of this file, 2 = the persistent 256 x 256 kernel (gemm256.hip)
// wherever it supports the shape.  Test / measurement hook (dark_dbg_set_gemm_variant).
static std::atomic<int> g_gemm_variant{0};
int gemm_variant() { return g_gemm_variant.load(std::memory_order_relaxed); }
void set_p256_nodma(int v);
void set_s256_probe(int v);
void set_s256_rows(int v);
double s256_best_cost(int epi, int M, int N, int* cus_out);
// 3 = timing probe: variant 2 with the main loop's LDS-DMA issue removed (wrong results); 16 + p = timing probe p of
// gemm_bf16_s256 (gemm256.hip, wrong results): variant 4 with parts of its intervals removed; 8 / 10 = variant 4 with
// its tile height forced to 256 / 288 / 320 rows (A/B of the instantiations; other values leave the choice automatic)
// 1000 + t = the automatic choice with gemm_bf16_s256's TUNE = t wherever that kernel runs (results unchanged)
void set_s256_tune(int v);
void set_attn_variant(int v);   // (attn.hip; 12 = the library's GEMM choice + the attention kernel's A/B form)
void set_s256_interleave(int v);   // (gemm256.hip; 14 = the library's choice without the interleaved LOAD interval)
void set_gemm_variant(int v) {
  set_attn_variant(v == 12 ? 1 : 0);
  set_s256_interleave(v == 14 ? 0 : 1);
  const bool tuned = v >= 1000;
  set_s256_tune(tuned ? v - 1000 : 0);
  if (tuned) v = 0;
  set_p256_nodma(v == 3);
  set_s256_probe(v >= 16 ? v - 16 : 0);
  set_s256_rows(v == 8 ? 256 : (v == 9 ? 288 : (v == 10 ? 320 : 0)));
  g_gemm_variant.store(v == 3 ? 2 : ((v >= 16 || (v >= 8 && v <= 10)) ? 4 : v), std::memory_order_relaxed);
}

// Which kernel serves a bf16 GEMM: 0 = the 128-row kernels of this file, 2 = gemm_bf16_p256, 4 = gemm_bf16_s256.
// Automatic choice from the A/B of tools/gemm_ab.py (one process, interleaved rounds, M = 23552, random operands):
//   gate/up + SwiGLU (N 6144):  128x192 196 us | p256 205 us | s256 188 us     -> s256
//   LM head, fp32 out (N 15080): 128x128 706 us | p256 615 us | s256 611 us     -> s256
//   q/k/v + bias (N 2304):       128x192  88 us | p256 101 us | s256  94 us (256-row tiles: 828 tiles leave the fourth
//                                round 23 % full; 288-row tiles, added later: 85 us -> s256, see below)
// Few rows leave the 256 CUs short of 256 x 256 tiles: those launches stay on the 128-row kernels.
static int pick_kernel(int dtype, int epi, int M, int N, int K, int64_t ldo, const void* out, const void* bias) {
  if (dtype != DARK_BF16) return 0;
  int v = g_gemm_variant.load(std::memory_order_relaxed);
  // 6 / 7 / 11 = A/B of the automatic choice itself: 6 = without the one-round down_proj routing below, 7 = without
  // q/k/v on the 288-row persistent kernel, 11 = without o_proj there
  const bool no_resid_s256 = v == 6, qkv_s256 = v != 7, oproj_s256 = v != 11;
  if (v == 5 || v == 6 || v == 7 || v == 11 || v == 12 || v == 13 || v == 14) v = 0;  // 5 / 13 = automatic GEMM choice + the row-panel kernel for o_proj + down_proj / o_proj only (api.hip); 12 = attention A/B
  if (v == 1) return 0;
  const bool p_ok = p256_supports(epi, M, N, K, ldo, out, bias), s_ok = s256_supports(epi, M, N, K, ldo, out, bias);
  if (v == 2) return p_ok ? 2 : 0;
  if (v == 4) return s_ok ? 4 : 0;
  const bool enough_tiles = (int64_t)cdiv(M, 256) * cdiv(N, 256) >= 512;
  if (s_ok && enough_tiles && (epi == EPI_F32 || epi == EPI_SWIGLU)) return 4;
  // down_proj on a bf16 residual stream (N = 768, K = 3072): 736 tiles of 128 x 192 are 1.44 rounds of the 512
  // workgroup slots, 246 tiles of 288 x 256 are ONE 96 %-full round of the persistent kernel.  Costs in row-rounds of
  // 256 columns per CU (a round of the 128-row kernel = two 128 x 192 tiles per CU); the 128-row kernel's partial
  // rounds run faster than the model says (a lone workgroup has the CU to itself), hence the 10 % margin.  Measured
  // (tools/gemm_ab.py, M = 23552): 110 -> 88 us.  At K = 768 the one-round launch has no next tile to hide its
  // prologue and epilogue behind: 36.5 vs 37.8 us and +-0 in the loop in round 2; with the interleaved LOAD interval of
  // round 3 +0.2 % in the loop in three alternating pairs (profiles/r03zh_*), so o_proj goes there too.  q/k/v + bias (N = 2304): 738 tiles of 288 x 256
  // are three 96 %-full rounds against 4.3 rounds of 128 x 192: 85.2 vs 89.0 us, +0.2 % in the loop in three
  // alternating pairs (profiles/r02ab_qkv_routing_ab.txt) -- small, but the same bits from one kernel family.
  if (s_ok && ((epi == EPI_RESID_H && (K >= 1536 || oproj_s256) && !no_resid_s256) || (epi == EPI_BIAS && qkv_s256))) {
    int cus = 256;
    const double c256 = s256_best_cost(epi, M, N, &cus);
    const double c128 = (double)cdiv(cdiv(M, 128) * cdiv(N, 192), 2 * cus) * 192.0;   // two workgroups per CU
    // (K >= 1536: at equal row-rounds the persistent kernel's longer main loop wins -- M = 131072, BASELINE config 4:
    // 561 vs 588 us, profiles/r03o_gemm_ab_config4_rows.txt -- so no margin there)
    if (epi == EPI_RESID_H && K >= 1536 ? c256 <= c128 : c256 * 1.1 < c128) return 4;
  }
  return 0;
}

static int launch_gemm_impl(int dtype, int epi, const void* A, const void* W, const float* bias, void* out, int M,
                            int N, int K, int64_t ldo, hipStream_t st) {
  DARK_CHECK_ARG(dtype == DARK_F32 || dtype == DARK_BF16, "gemm: bad dtype %d", dtype);
  DARK_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  DARK_CHECK_ARG(K % 64 == 0, "gemm: K=%d must be a multiple of 64", K);
  DARK_CHECK_ARG(M < (1 << 24) && N < (1 << 24), "gemm: M=%d N=%d too large", M, N);
  DARK_CHECK_ARG(A && W && out, "gemm: null operand");
  const int kern = pick_kernel(dtype, epi, M, N, K, ldo, out, bias);
  if (kern != 0) {
    DARK_CHECK_ARG(epi != EPI_BIAS || bias != nullptr, "gemm: EPI_BIAS needs a bias");
    return kern == 4 ? launch_gemm_s256(epi, A, W, bias, out, M, N, K, ldo, cdiv(N, 128) * 128, st)
                     : launch_gemm_p256(epi, A, W, bias, out, M, N, K, ldo, cdiv(N, 128) * 128, st);
  }
  switch (epi) {
    case EPI_BIAS:
      DARK_CHECK_ARG(bias != nullptr, "gemm: EPI_BIAS needs a bias");
      return launch_epi<EPI_BIAS>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_RESID: return launch_epi<EPI_RESID>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_BIAS_GELU:
      DARK_CHECK_ARG(bias != nullptr, "gemm: EPI_BIAS_GELU needs a bias");
      return launch_epi<EPI_BIAS_GELU>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_RESID_BIAS:
      DARK_CHECK_ARG(bias != nullptr, "gemm: EPI_RESID_BIAS needs a bias");
      return launch_epi<EPI_RESID_BIAS>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_RESID_H: return launch_epi<EPI_RESID_H>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_RESID_BIAS_H:
      DARK_CHECK_ARG(bias != nullptr, "gemm: EPI_RESID_BIAS_H needs a bias");
      return launch_epi<EPI_RESID_BIAS_H>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_SWIGLU:
      DARK_CHECK_ARG(N % 32 == 0, "gemm: SWIGLU needs N %% 32 == 0 (N=%d)", N);
      return launch_epi<EPI_SWIGLU>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_F32: return launch_epi<EPI_F32>(dtype, A, W, bias, out, M, N, K, ldo, st);
    case EPI_LSE:
      DARK_CHECK_ARG(bias != nullptr && ldo == 2 * cdiv(N, 128), "gemm: EPI_LSE needs targets and ldo = 2*ceil(N/128)");
      return launch_epi<EPI_LSE>(dtype, A, W, bias, out, M, N, K, ldo, st);
  }
  set_error("gemm: bad epilogue %d", epi);
  return DARK_ERR_ARG;
}

}  // namespace dark
