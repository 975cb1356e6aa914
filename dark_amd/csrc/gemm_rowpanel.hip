// Row-panel bf16 GEMM for the two residual projections of a layer (o_proj, down_proj; modeling_dream_length.py:384,
// :222) with the residual add AND the following RMSNorm (modeling_dream_length.py:69-74) in its epilogue:
//     hidden[M,768] (fp32) += A[M,K] * W[768,K]^T ;   xn[M,768] (bf16) = ln_w * hidden / rms(hidden)
//
// Why a third GEMM shape.  N = d_model = 768 is narrow: 128 x 192 tiles give 736 workgroups over 512 slots (1.44
// rounds, the second 44 % full) and the output row is cut into four tiles, so the norm that follows needs its own pass
// over the fp32 residual (72 MB read + 36 MB written per norm at M = 23552, 16 us, twice a layer).  Here a workgroup
// owns 96 COMPLETE rows: cdiv(23552, 96) = 246 workgroups = one round on 256 CUs, and the epilogue already holds
// every element of its rows, so the row's sum of squares, the scale and the bf16 normalised row come out of the same
// registers -- the separate norm kernels of layers 1.. disappear.
//
// Structure (one workgroup of 8 waves per CU, 160 KiB of LDS):
//   * wave w owns columns [96w, 96w + 96) of all 96 rows: 6 x 6 MFMA tiles (v_mfma_f32_16x16x32_bf16), 144
//     accumulator registers, 12 fragment reads per 36 MFMAs;
//   * W is streamed by LDS-DMA in 32-deep K-tiles; a wave's slice of W is read by that wave only, so every wave
//     stages its OWN 6 pieces (3 slots of 48 KiB, two K-tiles ahead) and needs no barrier for them;
//   * the 96 x 32 A tile is shared by all waves and is read exactly once from memory (no L2 reuse, HBM / Infinity-
//     Cache latency): it travels through registers, fetched three K-tiles ahead by the first wave group and written
//     to a 2-slot LDS ring one K-tile ahead;
//   * the two wave groups run one barrier apart (as gemm_bf16_s256): while one reads fragments and issues loads the
//     other holds the MFMA pipe;
//   * every wave issues the same memory instructions every K-tile (past the last K-tile they carry out-of-range
//     offsets and fetch nothing), so the counted `vmcnt` waits are exact and the loop has no branches.
// Summation order of the norm: per row, wave w adds the squares of its 24 columns per lane in a fixed fma chain, the
// four lanes of a row are combined by two xor-shuffles, the 8 wave partials are added in order 0..7.  The residual
// stream is bit-identical to the unfused kernels'; xn differs from rmsnorm_kernel's by the rounding of a differently
// associated fp32 sum (at most 1 bf16 ulp on a few elements in 10^4).
//
// HB = true (round 3): the residual stream is bf16 (the model dtype).  The epilogue brings two
// adjacent 16-column tiles into 8 consecutive columns a lane (one permlane16_swap per register, as EPI_RESID_H of
// gemm256.hip), adds the old bf16 values in fp32 and rounds once -- the same bits as the unfused residual GEMM -- and
// normalises the ROUNDED values, which is what the separate norm kernel would read back.  The sum of squares follows
// rmsnorm_h16_kernel's order exactly (elementwise.hip: per 8-column piece an fma chain from zero, the pieces l, l + 32,
// l + 64 of virtual lane l added in that order, then the 16 / 8 / 4 / 2 / 1 butterfly over the 32 lanes), so xn is
// bit-identical to the two-launch form too and the two can be mixed per shape.
#include "common.h"

namespace dark {

namespace {
constexpr int R_BM = 96, R_N = 768, R_BK = 32;
constexpr int R_WSLOT = R_N * R_BK * 2;           // 48 KiB: one K-tile of W
constexpr int R_ASLOT = R_BM * R_BK * 2;          // 6 KiB: one K-tile of A
constexpr int R_AOFF = 3 * R_WSLOT;               // A ring behind the three W slots
constexpr int R_RED = R_AOFF + 2 * R_ASLOT;       // 8 x 96 wave partials of the row sums
constexpr int R_DUMMY = R_RED + 8 * R_BM * 4;     // 512 B that the second group's (dead) A writes land in
constexpr int R_SMEM = R_DUMMY + 512;             // 163328 of 163840 bytes
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr int waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }  // vmcnt(n), nothing else
constexpr unsigned OOB = 0xF0000000u;  // beyond every descriptor: loads return 0, stores are dropped

__device__ __forceinline__ uint32_t pk2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}
template <int I> struct IC { static constexpr int value = I; };
}  // namespace

template <bool NORM, bool HB = false>
__global__ __launch_bounds__(512) void gemm_bf16_rowpanel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                          void* __restrict__ hidden, const float* __restrict__ ln_w,
                                                          bf16_t* __restrict__ xn, int M, int K, float eps,
                                                          unsigned a_bytes, unsigned w_bytes, unsigned h_bytes,
                                                          unsigned x_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int m0 = blockIdx.x * R_BM;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hidden, 0, (int)h_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (int)x_bytes, 0x00020000);
  const unsigned kb = (unsigned)K * 2u;
  const int KT = K / R_BK;  // a multiple of 3, >= 3

  // ---- W: this wave's six 16-row pieces of a K-tile, lane-linear in LDS; the 16-B chunk c of row r sits at position
  // c ^ ((r >> 2) & 2) (conflict-free ds_read_b128 fragments), applied to the per-lane SOURCE offset
  const unsigned w_voff = (unsigned)(wave * 96 + (lane >> 2)) * kb + (unsigned)(((lane & 3) ^ ((lane >> 4) & 2)) << 4);
  auto stage_w = [&](int slot, int kt) {
    char* base = smem + slot * R_WSLOT + wave * 6144;
    const unsigned o = kt < KT ? w_voff + (unsigned)kt * (R_BK * 2) : OOB;  // (OOB + 80 rows stays out of range: K <= 65536)
#pragma unroll
    for (int j = 0; j < 6; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + j * 1024), 16, o + (unsigned)j * 16u * kb, 0, 0, 0);
  };

  // ---- A: group 0's wave w fetches rows 24w .. 24w+23 of the tile, 8 bytes a lane, three loads a K-tile; group 1
  // issues the same instructions with out-of-range offsets (nothing is fetched) and writes them to the dummy area
  unsigned a_goff[3];
  int a_lds[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int r = 24 * (wave & 3) + 8 * t + (lane >> 3), c8 = lane & 7;
    const bool live = grp == 0 && m0 + r < M;
    a_goff[t] = live ? (unsigned)(m0 + r) * kb + (unsigned)c8 * 8u : OOB;
    const int rr = r & 15;
    a_lds[t] = grp == 0 ? R_AOFF + (r >> 4) * 1024 + rr * 64 + ((((c8 >> 1) ^ ((rr >> 2) & 2))) << 4) + (c8 & 1) * 8
                        : R_DUMMY + lane * 8;
  }
  auto load_a = [&](u32x2_t (&dst)[3], int kt) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
      dst[t] = __builtin_amdgcn_raw_buffer_load_b64(ra, (kt < KT && a_goff[t] != OOB) ? a_goff[t] + (unsigned)kt * (R_BK * 2) : OOB,
                                                    0, 0);
  };
  auto write_a = [&](const u32x2_t (&src)[3], int slot) {
    const int so = grp == 0 ? slot * R_ASLOT : 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) *(u32x2_t*)(smem + a_lds[t] + so) = src[t];
  };

  const int frow = lane & 15, fq = lane >> 4;
  const int fch = (fq ^ ((frow >> 2) & 2)) << 4;
  const int w_rd = wave * 6144 + frow * 64 + fch, a_rd = R_AOFF + frow * 64 + fch;

  f32x4 acc[6][6];     // [column tile][row tile]
  bf16x8 fa[6], fw[6];
  u32x2_t areg[3][3];  // A K-tiles in flight, ring indexed by kt % 3

  // ---- prologue: A(0), A(1), W(0), A(2), W(1); A(0) into LDS
  load_a(areg[0], 0);
  load_a(areg[1], 1);
  stage_w(0, 0);
  load_a(areg[2], 2);
  stage_w(1, 1);
  __builtin_amdgcn_s_waitcnt(waitcnt_vm(9));  // A(0), A(1), W(0) have landed
  write_a(areg[0], 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 runs one interval behind group 0

  // (zeroed here rather than by a zero-C first K-tile: that peels the loop's first trip and the peel spills)
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto step = [&](auto SC, int kt) {
    constexpr int S = decltype(SC)::value;  // kt % 3
    // ================= LOAD interval =================
    {
      const char* ws = smem + S * R_WSLOT + w_rd;
      const char* as = smem + (kt & 1) * R_ASLOT + a_rd;
#pragma unroll
      for (int j = 0; j < 6; ++j) fw[j] = *(const bf16x8*)(ws + j * 1024);
#pragma unroll
      for (int i = 0; i < 6; ++i) fa[i] = *(const bf16x8*)(as + i * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    write_a(areg[(S + 1) % 3], (kt + 1) & 1);  // A(kt+1): landed (previous interval's wait), its slot is free
    load_a(areg[S], kt + 3);
    stage_w((S + 2) % 3, kt + 2);
    __builtin_amdgcn_sched_barrier(0);
    // outstanding, oldest first: A(kt+2) W(kt+1) | A(kt+3) W(kt+2)  ->  9 may stay in flight
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(9) & 0xF0FF);  // + lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= MFMA interval =================
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[j][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  for (int kt = 0; kt < KT; kt += 3) {
    step(IC<0>{}, kt);
    step(IC<1>{}, kt + 1);
    step(IC<2>{}, kt + 2);
  }

  // ================= epilogue =================
  if constexpr (HB) {
    // ---- bf16 residual stream: lane = row (lane & 15) of each 16-row tile, 8 consecutive columns of each tile pair
    const int hcol = (fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq;
    const int hp = ((fq & 1) << 1) | (fq >> 1);            // hcol / 8
    u32x4_t hb[6][3];                                      // the new residual values, packed bf16 (kept for the norm)
    float sp[6][3];                                        // per piece: sum of squares of the rounded values
    u32x4_t old[2][3];
    auto load_old = [&](u32x4_t (&dst)[3], int i) {
      const int m = m0 + 16 * i + frow;
#pragma unroll
      for (int np = 0; np < 3; ++np)
        dst[np] = __builtin_amdgcn_raw_buffer_load_b128(rh, m < M ? (unsigned)(m * R_N + wave * 96 + np * 32 + hcol) * 2u : OOB, 0, 0);
    };
    load_old(old[0], 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int m = m0 + 16 * i + frow;
      if (i + 1 < 6) load_old(old[(i + 1) & 1], i + 1);
#pragma unroll
      for (int np = 0; np < 3; ++np) {
        const f32x4 a = acc[2 * np][i], b = acc[2 * np + 1][i];
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
          v[e] = __uint_as_float(sw[0]);
          v[4 + e] = __uint_as_float(sw[1]);
        }
        const u32x4_t o = old[i & 1][np];
        u32x4_t r;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          r[j] = pk2(v[2 * j] + __uint_as_float(o[j] << 16), v[2 * j + 1] + __uint_as_float(o[j] & 0xffff0000u));
        hb[i][np] = r;
        __builtin_amdgcn_raw_buffer_store_b128(r, rh, m < M ? (unsigned)(m * R_N + wave * 96 + np * 32 + hcol) * 2u : OOB, 0, 0);
        if constexpr (NORM) {
          float q = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(r[j] << 16), hi = __uint_as_float(r[j] & 0xffff0000u);
            q = __builtin_fmaf(lo, lo, q);
            q = __builtin_fmaf(hi, hi, q);
          }
          sp[i][np] = q;
        }
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's last interval
    if constexpr (NORM) {
      __syncthreads();   // every wave is out of the main loop: the K-tile ring is dead and holds the piece sums now
      float* ps = (float*)smem;                    // [96 rows][96 pieces of 8 columns]
      float* tot = (float*)(smem + R_RED);         // [96 rows]
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int np = 0; np < 3; ++np) ps[(16 * i + frow) * 96 + 12 * wave + 4 * np + hp] = sp[i][np];
      __syncthreads();
      if (tid < 4 * R_BM) {
        // four threads a row rebuild rmsnorm_h16_kernel's sum: thread qd holds virtual lanes 8 qd .. 8 qd + 7
        const int row = tid >> 2, qd = tid & 3;
        float part[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int L = 8 * qd + e;
          part[e] = (ps[row * 96 + L] + ps[row * 96 + L + 32]) + ps[row * 96 + L + 64];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[e] += __shfl_xor(part[e], 2, 64);   // lanes L and L ^ 16
#pragma unroll
        for (int e = 0; e < 8; ++e) part[e] += __shfl_xor(part[e], 1, 64);   // L ^ 8
#pragma unroll
        for (int e = 0; e < 4; ++e) part[e] += part[e + 4];                  // L ^ 4
        part[0] += part[2]; part[1] += part[3];                              // L ^ 2
        part[0] += part[1];                                                  // L ^ 1
        if (qd == 0) tot[row] = part[0];
      }
      __syncthreads();
      float g[3][8];
#pragma unroll
      for (int np = 0; np < 3; ++np) {
        const float4 g0 = *(const float4*)(ln_w + wave * 96 + np * 32 + hcol), g1 = *(const float4*)(ln_w + wave * 96 + np * 32 + hcol + 4);
        g[np][0] = g0.x; g[np][1] = g0.y; g[np][2] = g0.z; g[np][3] = g0.w;
        g[np][4] = g1.x; g[np][5] = g1.y; g[np][6] = g1.z; g[np][7] = g1.w;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int row = 16 * i + frow, m = m0 + row;
        const float r = 1.0f / sqrtf(tot[row] / (float)R_N + eps);
#pragma unroll
        for (int np = 0; np < 3; ++np) {
          u32x4_t y;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(hb[i][np][j] << 16), hi = __uint_as_float(hb[i][np][j] & 0xffff0000u);
            y[j] = pk2(g[np][2 * j] * (lo * r), g[np][2 * j + 1] * (hi * r));
          }
          __builtin_amdgcn_raw_buffer_store_b128(y, rx, m < M ? (unsigned)(m * R_N + wave * 96 + np * 32 + hcol) * 2u : OOB, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    return;
  }
  // this lane owns row (lane & 15) of each 16-row tile and columns 96w + 16j + 4*fq .. +3 of each column tile
  const unsigned col_b = (unsigned)(wave * 96 + 4 * fq) * 4u;
  float* red = (float*)(smem + R_RED);
  u32x4_t old[2][6];  // the residual's old values, fetched one row tile ahead of their use
  auto load_old = [&](u32x4_t (&dst)[6], int i) {
    const int m = m0 + 16 * i + frow;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rh, m < M ? (unsigned)m * (R_N * 4) + col_b + j * 64 : OOB, 0, 0);
  };
  load_old(old[0], 0);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int m = m0 + 16 * i + frow;
    const unsigned rowoff = m < M ? (unsigned)m * (R_N * 4) + col_b : OOB;
    if (i + 1 < 6) load_old(old[(i + 1) & 1], i + 1);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[j][i][e] + __uint_as_float(old[i & 1][j][e]);
        q = __builtin_fmaf(v[e], v[e], q);
      }
      acc[j][i] = v;
      __builtin_amdgcn_raw_buffer_store_b128(
          (u32x4_t){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rh,
          m < M ? rowoff + j * 64 : OOB, 0, 0);
    }
    if constexpr (NORM) {
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (fq == 0) red[wave * R_BM + 16 * i + frow] = q;
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's last interval
  if constexpr (NORM) {
    __syncthreads();
    float4 g[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) g[j] = *(const float4*)(ln_w + wave * 96 + 16 * j + 4 * fq);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = 16 * i + frow, m = m0 + row;
      float tot = red[row];
#pragma unroll
      for (int w = 1; w < 8; ++w) tot += red[w * R_BM + row];
      const float r = 1.0f / sqrtf(tot / (float)R_N + eps);
#pragma unroll
      for (int np = 0; np < 3; ++np) {
        float y[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 v = acc[2 * np + t][i];
          const float4 gg = g[2 * np + t];
          y[t][0] = gg.x * (v[0] * r); y[t][1] = gg.y * (v[1] * r);
          y[t][2] = gg.z * (v[2] * r); y[t][3] = gg.w * (v[3] * r);
        }
        // two adjacent 16-column tiles -> one 16-byte store per lane (one permlane16_swap per dword)
        const uint32_t a_lo = pk2(y[0][0], y[0][1]), a_hi = pk2(y[0][2], y[0][3]);
        const uint32_t b_lo = pk2(y[1][0], y[1][1]), b_hi = pk2(y[1][2], y[1][3]);
        const auto sx = __builtin_amdgcn_permlane16_swap(a_lo, b_lo, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(a_hi, b_hi, false, false);
        const int c = wave * 96 + np * 32 + ((fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){sx[0], sy[0], sx[1], sy[1]}, rx,
                                               m < M ? (unsigned)(m * R_N + c) * 2u : OOB, 0, 0);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
}

// 16-byte rows, K in whole triples of 32-deep tiles, 32-bit byte offsets for every buffer
bool rowpanel_supports(int M, int N, int K, const void* A, const void* W, const void* hidden, const void* ln_w,
                       const void* xn) {
  if (N != R_N || K % (3 * R_BK) != 0 || K < 3 * R_BK || K > 65536 || M <= 0) return false;
  if ((int64_t)M * K * 2 >= 0xE0000000ll || ((int64_t)M + R_BM) * R_N * 4 >= 0xE0000000ll) return false;
  if ((int64_t)R_N * K * 2 >= 0xE0000000ll) return false;
  if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)hidden | (uintptr_t)ln_w | (uintptr_t)xn) & 15) != 0) return false;
  return true;
}

int launch_gemm_resid_norm_h(const void* A, const void* W, void* hidden, const float* ln_w, void* xn, int M, int K,
                             float eps, hipStream_t st) {
  DARK_CHECK_ARG(A && W && hidden && (ln_w == nullptr) == (xn == nullptr), "gemm_resid_norm_h: null operand");
  DARK_CHECK_ARG(rowpanel_supports(M, R_N, K, A, W, hidden, ln_w, xn), "gemm_resid_norm_h: unsupported shape M=%d K=%d", M, K);
  const int grid = cdiv(M, R_BM);
  const unsigned a_bytes = (unsigned)((int64_t)M * K * 2), w_bytes = (unsigned)((int64_t)R_N * K * 2);
  const unsigned h_bytes = (unsigned)((int64_t)M * R_N * 2), x_bytes = (unsigned)((int64_t)M * R_N * 2);
  if (ln_w) {
    static SmemOptIn optin;
    DARK_TRY(optin.ensure((const void*)gemm_bf16_rowpanel<true, true>, R_SMEM, st));
    hipLaunchKernelGGL((gemm_bf16_rowpanel<true, true>), dim3(grid), dim3(512), R_SMEM, st, (const bf16_t*)A, (const bf16_t*)W,
                       hidden, ln_w, (bf16_t*)xn, M, K, eps, a_bytes, w_bytes, h_bytes, x_bytes);
  } else {
    static SmemOptIn optin;
    DARK_TRY(optin.ensure((const void*)gemm_bf16_rowpanel<false, true>, R_SMEM, st));
    hipLaunchKernelGGL((gemm_bf16_rowpanel<false, true>), dim3(grid), dim3(512), R_SMEM, st, (const bf16_t*)A, (const bf16_t*)W,
                       hidden, ln_w, (bf16_t*)xn, M, K, eps, a_bytes, w_bytes, h_bytes, 0u);
  }
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int launch_gemm_resid_norm(const void* A, const void* W, float* hidden, const float* ln_w, void* xn, int M, int K,
                           float eps, hipStream_t st) {
  DARK_CHECK_ARG(A && W && hidden && (ln_w == nullptr) == (xn == nullptr), "gemm_resid_norm: null operand");
  DARK_CHECK_ARG(rowpanel_supports(M, R_N, K, A, W, hidden, ln_w, xn), "gemm_resid_norm: unsupported shape M=%d K=%d", M, K);
  const int grid = cdiv(M, R_BM);
  const unsigned a_bytes = (unsigned)((int64_t)M * K * 2), w_bytes = (unsigned)((int64_t)R_N * K * 2);
  const unsigned h_bytes = (unsigned)((int64_t)M * R_N * 4), x_bytes = (unsigned)((int64_t)M * R_N * 2);
  if (ln_w) {
    static SmemOptIn optin;
    DARK_TRY(optin.ensure((const void*)gemm_bf16_rowpanel<true>, R_SMEM, st));
    hipLaunchKernelGGL((gemm_bf16_rowpanel<true>), dim3(grid), dim3(512), R_SMEM, st, (const bf16_t*)A, (const bf16_t*)W,
                       (void*)hidden, ln_w, (bf16_t*)xn, M, K, eps, a_bytes, w_bytes, h_bytes, x_bytes);
  } else {
    static SmemOptIn optin;
    DARK_TRY(optin.ensure((const void*)gemm_bf16_rowpanel<false>, R_SMEM, st));
    hipLaunchKernelGGL((gemm_bf16_rowpanel<false>), dim3(grid), dim3(512), R_SMEM, st, (const bf16_t*)A, (const bf16_t*)W,
                       (void*)hidden, ln_w, (bf16_t*)xn, M, K, eps, a_bytes, w_bytes, h_bytes, 0u);
  }
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

}  // namespace dark
