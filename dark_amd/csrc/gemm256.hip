// Persistent 256 x 256 bf16 GEMMs for the wide projections of the denoise forward: C[M,N] = A[M,K] * W[N,K]^T, both
// operands K-contiguous.  Two kernels: gemm_bf16_p256 (first form, kept as A/B variant 2) and gemm_bf16_s256 (the
// staggered form further down: the one that runs).
//
// Replaces the same nn.Linear calls as gemm.hip for the shapes where a 256-wide macro tile quantises well:
// gate/up with the fused SwiGLU (modeling_dream_length.py:221-222), the LM head (:809), incl. the fused head's
// slice-record flavour (EPI_HEAD), down_proj + residual (:222, :514) on a bf16 residual stream, whose 23552 rows are
// ONE 96 %-full round of 288 x 256 tiles, and q/k/v + bias (:381-383: three 96 %-full rounds of 288-row tiles).
// gemm.hip keeps o_proj (K = 768: a one-round launch has nothing to hide its prologue and epilogue behind), every
// residual GEMM on an fp32 stream and every launch with too few tiles (pick_kernel in gemm.hip).
//
// Why another kernel: the 128 x 192 kernel of gemm.hip moves (128 + 192) * 128 B from L2 into LDS per 128*192*64
// MACs and reads 10 fragments per 24 MFMAs; a 256 x 256 tile with 128 x 64 per wave moves 39 % fewer bytes per MAC
// through the LDS-DMA path and reads 12 fragments per 32 MFMAs.  At 128 KiB of LDS only ONE workgroup fits a CU, so
// what the second workgroup did for gemm.hip (covering the other's barrier, DMA wait, prologue and epilogue) has to
// come from the structure of the single workgroup:
//   * 8 waves (2 x 4), two per SIMD; fragments are double-buffered in registers per half K-tile: the ds_read_b128 of
//     half h+1 are issued in front of the 32 MFMAs of half h, so LDS latency never sits in front of an MFMA;
//   * ONE barrier per K-tile, in the middle of the tile, in front of which every wave waits for its own LDS-DMA of
//     the next K-tile (issued a whole K-tile earlier); behind it the buffer just read is refilled two K-tiles ahead;
//   * persistent over output tiles: the first two K-tiles of the NEXT output tile are in flight while the epilogue of
//     the current one runs, so a tile's prologue costs nothing and the K = 768 main loop (12 K-tiles) never restarts
//     from an empty pipeline.
// Staging: buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB per wave-instruction); the LDS image is lane-linear with the
// XOR swizzle of gemm.hip applied to the per-lane SOURCE offset.  Out-of-range rows are read as zeros by the buffer
// descriptor's bounds check (row offsets travel in the VGPR operand, the only one the check covers), so edge tiles
// need no clamping.
#include "common.h"
#include <type_traits>

namespace dark {

namespace {

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int P_OPER = PBM * PBK * 2;      // bytes of one operand tile (32 KiB)
constexpr int P_STAGE = 2 * P_OPER;        // A + W
constexpr int P_SMEM = 2 * P_STAGE;        // two K-tiles in LDS: 128 KiB
constexpr int P_MI = 8, P_NI = 4;          // 16-row / 16-column MFMA tiles per wave (128 x 64)

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float silu_hw(float g) {
  return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.44269504088896340736f));
}
// silu(g) * u for four accumulator values at once, the same operations as silu_hw(g) * u with the multiplies and the
// add on the packed fp32 pipe (v_pk_mul_f32 / v_pk_add_f32: two lanes' worth per instruction, same IEEE results): the
// epilogue is VALU-bound and exposed, and the scalar form spent 160 of its 520 instructions on these.
__device__ __forceinline__ void silu_mul4(const f32x4& g, const f32x4& u, float (&out)[4]) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x2_t g2 = {g[2 * h], g[2 * h + 1]}, u2 = {u[2 * h], u[2 * h + 1]};
    const f32x2_t t = g2 * (f32x2_t){-1.44269504088896340736f, -1.44269504088896340736f};
    const f32x2_t d = (f32x2_t){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + (f32x2_t){1.0f, 1.0f};
    const f32x2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2_t o = (g2 * r) * u2;
    out[2 * h] = o[0];
    out[2 * h + 1] = o[1];
  }
}

// Two horizontally adjacent 16 x 16 tiles of this lane's final values (row = lane & 15, columns cb + 16t + 4q + e)
// as ONE 16-byte bf16 store per lane (see store_pair_bf16 in gemm.hip: one permlane16_swap per dword).
__device__ __forceinline__ void store_pair(const float (&v0)[4], const float (&v1)[4], bf16_t* __restrict__ orow, int cb,
                                           int ncols, int q) {
  const uint32_t a_lo = pk_bf16(v0[0], v0[1]), a_hi = pk_bf16(v0[2], v0[3]);
  const uint32_t b_lo = pk_bf16(v1[0], v1[1]), b_hi = pk_bf16(v1[2], v1[3]);
  const auto sx = __builtin_amdgcn_permlane16_swap(a_lo, b_lo, false, false);
  const auto sy = __builtin_amdgcn_permlane16_swap(a_hi, b_hi, false, false);
  const int c = cb + ((q & 1) ? 16 + 4 * (q - 1) : 4 * q);
  if (c + 8 <= ncols) *(uint4*)(orow + c) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) unsigned lds_u32;

}  // namespace

template <int EPI, bool NODMA = false>
__global__ __launch_bounds__(512) void gemm_bf16_p256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                      const float* __restrict__ bias, void* __restrict__ out, int M,
                                                      int N, int K, int64_t ldo, int tilesM, int tilesN, int ntiles,
                                                      unsigned a_bytes, unsigned w_bytes) {
  constexpr int GM = 8;  // row panels per raster group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);

  // ---- LDS-DMA: each wave stages 4 pieces (8 rows x 128 B) of A and 4 of W per K-tile.  Piece p covers rows
  // 8p .. 8p+7; lane l -> row 8p + (l >> 3), physical 16-B chunk l & 7 holds logical chunk (l & 7) ^ ((row >> 1) & 7).
  // (row >> 1) & 7 = (4p + (l >> 4)) & 7 depends on the piece's parity only -> two per-lane offsets per operand.
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)w_bytes, 0x00020000);
  const int rib = lane >> 3;  // row inside the piece
  const unsigned kb = (unsigned)K * 2u;  // bytes per operand row
  unsigned voff[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int lch = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);
    voff[par] = (unsigned)rib * kb + (unsigned)lch * 16u;
  }
  // stage K-tile `kt` of the tile at byte offsets (a0, w0) into LDS buffer `buf`
  auto stage = [&](int buf, unsigned a0, unsigned w0, int kt) {
    char* base = smem + buf * P_STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = wave * 4 + q;  // piece 0..31 of the operand tile
      const unsigned so = (unsigned)p * 8u * kb + (unsigned)kt * (PBK * 2);
      // (the whole offset goes into the VGPR operand: the SGPR offset is EXCLUDED from the descriptor's bounds check)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + p * 1024), 16, voff[q & 1] + (a0 + so), 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = wave * 4 + q;
      const unsigned so = (unsigned)p * 8u * kb + (unsigned)kt * (PBK * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + P_OPER + p * 1024), 16, voff[q & 1] + (w0 + so), 0, 0, 0);
    }
  };

  const int frow = lane & 15, fq = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const int a_rd = (wm * 128 + frow) * 128, w_rd = P_OPER + (wn * 64 + frow) * 128;
  const int ch0 = ((0 + fq) ^ fsw) << 4, ch1 = ((4 + fq) ^ fsw) << 4;

  auto tile_of = [&](int id, int& tm, int& tn) {
    const int per_group = GM * tilesN;
    const int group = id / per_group;
    const int first_m = group * GM;
    const int gsz = min(GM, tilesM - first_m);
    const int rem = id - group * per_group;
    tn = rem / gsz;
    tm = first_m + (rem - tn * gsz);
  };

  const int KT = K / PBK;
  int gk = 0;  // K-tiles staged so far by this workgroup: K-tile number g lives in LDS buffer g & 1
  int id = wg;
  if (id >= ntiles) return;
  {
    int tm, tn;
    tile_of(id, tm, tn);
    const unsigned a0 = (unsigned)tm * PBM * kb, w0 = (unsigned)tn * PBN * kb;
    stage(0, a0, w0, 0);
    stage(1, a0, w0, 1 % KT);  // KT == 1: staged twice, harmless
  }

  for (; id < ntiles; id += G) {
    int tm, tn;
    tile_of(id, tm, tn);
    const int m0 = tm * PBM, n0 = tn * PBN;
    const int nid = id + G;
    unsigned na0 = 0, nw0 = 0;
    if (nid < ntiles) {
      int ntm, ntn;
      tile_of(nid, ntm, ntn);
      na0 = (unsigned)ntm * PBM * kb;
      nw0 = (unsigned)ntn * PBN * kb;
    }
    const unsigned a0 = (unsigned)m0 * kb, w0 = (unsigned)n0 * kb;

    float4 b4[P_NI];
    if constexpr (EPI == EPI_BIAS) {
#pragma unroll
      for (int ni = 0; ni < P_NI; ++ni) {
        const int c = n0 + wn * 64 + ni * 16 + 4 * fq;
        b4[ni].x = c + 0 < N ? bias[c + 0] : 0.f; b4[ni].y = c + 1 < N ? bias[c + 1] : 0.f;
        b4[ni].z = c + 2 < N ? bias[c + 2] : 0.f; b4[ni].w = c + 3 < N ? bias[c + 3] : 0.f;
      }
    }

    f32x4 acc[P_NI][P_MI];
    bf16x8 fa0[P_MI], fw0[P_NI], fa1[P_MI], fw1[P_NI];

    auto rd = [&](bf16x8 (&fa)[P_MI], bf16x8 (&fw)[P_NI], int buf, int ch) {
      const char* as = smem + buf * P_STAGE + a_rd + ch;
      const char* ws = smem + buf * P_STAGE + w_rd + ch;
#pragma unroll
      for (int j = 0; j < P_NI; ++j) fw[j] = *(const bf16x8*)(ws + j * 2048);
#pragma unroll
      for (int i = 0; i < P_MI; ++i) fa[i] = *(const bf16x8*)(as + i * 2048);
    };
    auto mm = [&](const bf16x8 (&fa)[P_MI], const bf16x8 (&fw)[P_NI]) {
#pragma unroll
      for (int j = 0; j < P_NI; ++j)
#pragma unroll
        for (int i = 0; i < P_MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[j][i], 0, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < P_NI; ++j)
#pragma unroll
      for (int i = 0; i < P_MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K-tile 0 of this tile was issued 16 DMAs ago (8 for it, 8 for K-tile 1), before the previous tile's epilogue
    // stores: once at most 8 operations are outstanding it has landed (vmcnt retires in order).
    __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8) (the builtin, so that the compiler's own wait bookkeeping sees it)
    __builtin_amdgcn_s_barrier();
    rd(fa0, fw0, gk & 1, ch0);
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = (gk + kt) & 1;
      rd(fa1, fw1, buf, ch1);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa0, fw0);
      __builtin_amdgcn_sched_barrier(0);
      // my LDS-DMA of the next K-tile has landed, my reads of this one have returned; behind the barrier that holds
      // for every wave: the next K-tile may be read and this buffer refilled
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!NODMA) {  // (NODMA: timing probe only -- what the main loop costs without its LDS-DMA issue)
        if (kt + 2 < KT) stage(buf, a0, w0, kt + 2);
        else if (nid < ntiles) stage(buf, na0, nw0, (kt + 2 - KT) % KT);  // the next output tile's K-tiles 0 and 1
      }
      if (kt + 1 < KT) rd(fa0, fw0, buf ^ 1, ch0);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa1, fw1);
      __builtin_amdgcn_sched_barrier(0);
    }
    gk += KT;

    // ---- epilogue: lane owns row (lane & 15) of each 16-row tile and 4 consecutive columns of each 16-column tile
#pragma unroll
    for (int mi = 0; mi < P_MI; ++mi) {
      const int m = m0 + wm * 128 + mi * 16 + frow;
      const bool row_ok = m < M;
      const int64_t mrow = row_ok ? m : 0;
      if constexpr (EPI == EPI_SWIGLU) {
        // gate/up rows interleaved in groups of 16: column tiles 2p / 2p+1 are gate / up of 16 output columns
        float v[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          silu_mul4(acc[2 * p][mi], acc[2 * p + 1][mi], v[p]);
        store_pair(v[0], v[1], (bf16_t*)out + mrow * ldo, (n0 + wn * 64) >> 1, row_ok ? (N >> 1) : 0, fq);
      } else if constexpr (EPI == EPI_BIAS) {
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          float v[2][4];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int ni = 2 * np + t;
            v[t][0] = acc[ni][mi][0] + b4[ni].x; v[t][1] = acc[ni][mi][1] + b4[ni].y;
            v[t][2] = acc[ni][mi][2] + b4[ni].z; v[t][3] = acc[ni][mi][3] + b4[ni].w;
          }
          store_pair(v[0], v[1], (bf16_t*)out + mrow * ldo, n0 + wn * 64 + np * 32, row_ok ? N : 0, fq);
        }
      } else {  // EPI_F32
        if (row_ok) {
          float* orow = (float*)out + mrow * ldo;
#pragma unroll
          for (int ni = 0; ni < P_NI; ++ni) {
            const int c = n0 + wn * 64 + ni * 16 + 4 * fq;
            if (c + 4 <= N) *(float4*)(orow + c) = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ====================================================================================================================
// Staggered variant (s256): what the measurements on gemm_bf16_p256 asked for.  Timing that kernel with its main-loop
// LDS-DMA issue removed (dark_dbg_set_gemm_variant(3)) takes the K-tile from 1.51 to 1.20 us: a wave spends ~80 cycles
// issuing each `buffer_load ... lds`, all eight waves do it at the same point behind the barrier, and no wave feeds the
// MFMA pipe meanwhile; the SwiGLU epilogue (exp2 + rcp per output) is exposed in the same way, 4 us per output tile.
// Here the eight waves are two groups of four (wave >= 4 is the partner of wave - 4 on its SIMD) that run the SAME
// program one barrier interval apart: while one group's 32 MFMAs of a K-tile execute, the other group reads its
// fragments for the next K-tile and issues its share of the LDS-DMA (and, at a tile boundary, runs its epilogue), then
// the roles swap.  K-tiles are 32 deep (one MFMA phase per wave and K-tile), four of them live in LDS (4 x 32 KiB), the
// DMA runs three K-tiles ahead and continues across output tiles; one fragment set in registers suffices.
//   LDS image of an operand K-tile: 256 rows x 64 B, lane-linear in 16-row pieces (one wave-instruction each); the
//   16-B chunk c of row r sits at position c ^ ((r >> 2) & 2), which makes the ds_read_b128 fragment reads
//   conflict-free (searched exhaustively over the four 16-lane groups); applied to the per-lane SOURCE offset.
// Output goes through buffer stores: rows >= M and columns >= N fall outside the descriptor and are dropped by the
// hardware, so every wave issues the same number of stores per tile and the counted `vmcnt` waits stay exact.
namespace {
constexpr int SBK = 32;
// K-tiles live in LDS.  Five (all 160 KiB) for every flavour but the bias one, which keeps its bias vector behind four:
// per-phase timestamps (tools/probe/gemm_s256_probe.py) showed ~90 cycles per interval of EXPOSED wait for the DMA
// issued three K-tiles (~1500 cycles) earlier -- L2 / Infinity-Cache latency under a load of 256 CUs x 40 GB/s -- on
// the LOAD side, which is the critical side of every interval; a fifth buffer lets the DMA run four K-tiles ahead.
// MI = 16-row MFMA tiles per wave: the tile is 32 * MI rows x 256 columns.  MI = 8 is the 256 x 256 tile; MI = 10
// (320 x 256: 160 accumulator + 56 fragment registers a lane, 243 VGPRs) moves 10 % fewer L2 -> LDS bytes per MAC and
// quantises M = 23552 better (74 x 24 = 1776 tiles = 6.94 rounds of 256 CUs against 8.63 -> 9); its K-tile is
// 36 KiB, so four of them fit instead of five.  Measured: gate/up 181.4 vs 186.7 us, K = 3072 642 vs 674 us.
constexpr int s_nbuf(int epi, int mi) { return (epi == EPI_BIAS || mi > 8) ? 4 : 5; }
constexpr int S_WOPER = 256 * SBK * 2;                                   // 16 KiB: the weight K-tile
constexpr int s_aoper(int mi) { return 32 * mi * SBK * 2; }              // the activation K-tile
constexpr int s_buf(int mi) { return s_aoper(mi) + S_WOPER; }
constexpr int s_smem(int epi, int mi) { return s_nbuf(epi, mi) * s_buf(mi); }  // 160 KiB (MI = 10: 144; EPI_BIAS: 128 + bias)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int EPI> struct StoresPerTile;                                   // buffer stores a wave issues per tile
template <> struct StoresPerTile<EPI_SWIGLU> { static constexpr int PER_MI = 1; };   // ... per 16-row MFMA tile
template <> struct StoresPerTile<EPI_BIAS> { static constexpr int PER_MI = 2; };
template <> struct StoresPerTile<EPI_F32> { static constexpr int PER_MI = 4; };
template <> struct StoresPerTile<EPI_HEAD> { static constexpr int PER_MI = 1; };
template <> struct StoresPerTile<EPI_RESID_H> { static constexpr int PER_MI = 2; };
constexpr int waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }  // vmcnt(n), nothing else
// vmcnt(n) for a wave-uniform run-time n (s_waitcnt takes an immediate); LGKM0: lgkmcnt(0) as well.  n > 63 waits for 63.
template <bool LGKM0>
__device__ __forceinline__ void wait_vm_dyn(int n) {
#define DARK_W(N_) case N_: __builtin_amdgcn_s_waitcnt(LGKM0 ? (waitcnt_vm(N_) & 0xF0FF) : waitcnt_vm(N_)); break;
#define DARK_W8(B_) DARK_W(B_) DARK_W(B_ + 1) DARK_W(B_ + 2) DARK_W(B_ + 3) DARK_W(B_ + 4) DARK_W(B_ + 5) DARK_W(B_ + 6) DARK_W(B_ + 7)
  switch (n) {
    DARK_W8(0) DARK_W8(8) DARK_W8(16) DARK_W8(24) DARK_W8(32) DARK_W8(40) DARK_W8(48)
    DARK_W(56) DARK_W(57) DARK_W(58) DARK_W(59) DARK_W(60) DARK_W(61) DARK_W(62)
    default: __builtin_amdgcn_s_waitcnt(LGKM0 ? (waitcnt_vm(63) & 0xF0FF) : waitcnt_vm(63)); break;
  }
#undef DARK_W8
#undef DARK_W
}
}  // namespace

// PROBE != 0: timing-only builds (WRONG RESULTS) that tell where an interval's time goes -- bit 0: no LDS-DMA issue in
// the steady state, bit 1: no fragment reads after the first K-tile, bit 2: no epilogue arithmetic / stores; bit 3
// (results stay correct): per-wave cycle sums of the interval's phases, written through `bias` as int64[grid][8 waves][8]
// = {LOAD work until its reads have returned, wait for the next K-tile's DMA, barrier wait behind LOAD, MFMA issue,
// barrier wait behind the MFMAs, K-tiles} (s_memtime costs a few dozen cycles per sample: proportions, not absolutes).
// TUNE (results stay correct; A/B hooks of the interval structure, dark_dbg_set_gemm_variant(64 + TUNE)):
//   bit 0: group 0 waits for its pieces of the next K-tile at the END of its MFMA interval instead of the end of its
//          LOAD interval (they are first read behind the barrier that closes that MFMA interval);
//   bit 1: s_setprio 1 for the LOAD interval;
//   bit 2: the K-tile's DMA pieces are dealt unevenly: (TUNE >> 4) & 7 (0 = 6) a wave in group 0, the rest over group 1.
template <int EPI, int PROBE = 0, int MI = 8, int TUNE = 0>
__global__ __launch_bounds__(512) void gemm_bf16_s256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                      const float* __restrict__ bias, void* __restrict__ out, int M,
                                                      int N, int K, int64_t ldo, int tilesM, int tilesN, int ntiles,
                                                      unsigned a_bytes, unsigned w_bytes, unsigned o_bytes,
                                                      const int* __restrict__ m_dev, float hp_c2, float hp_range) {
  constexpr int GM = 8;
  constexpr int TM = 32 * MI, GROWS = 16 * MI;   // tile rows; rows per wave group
  constexpr int NST = StoresPerTile<EPI>::PER_MI * MI;
  constexpr int NB = s_nbuf(EPI, MI);      // LDS buffers; the DMA runs NB - 1 K-tiles ahead of the reads
  constexpr int S_OPER = s_aoper(MI), S_BUF = s_buf(MI);
  constexpr int S_SMEM = NB * S_BUF;
  // DMA pieces (16 rows x 64 B, one wave-instruction each) of a K-tile: 2 * MI of A, 16 of W.  Every wave brings two W
  // pieces and NALO A pieces; the first NHI waves bring one A piece more (MI = 8: none, MI = 9: waves 0 and 1, MI = 10:
  // the four waves of group 0), so "hi" and "lo" waves issue different numbers of DMAs per batch and carry their own
  // counted waits.
  constexpr int NALO = (2 * MI) / 8, NHI = 2 * MI - 8 * NALO;
  static_assert(NHI >= 0 && NHI < 8 && NHI * (NALO + 1) + (8 - NHI) * NALO == 2 * MI, "A pieces do not divide over the waves");
  constexpr int OPSHI = NALO + 3, OPSLO = NALO + 2;      // DMAs per batch of a hi / lo wave
  if (m_dev) {  // row count from device memory (never more than the host's M, which sized the buffers)
    const int md = __builtin_amdgcn_readfirstlane(*m_dev);
    M = md < M ? (md > 0 ? md : 0) : M;
    tilesM = (M + TM - 1) / TM;
    ntiles = tilesM * tilesN;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;  // group = row half of the tile
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  if (wg >= ntiles) return;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)o_bytes, 0x00020000);
  const unsigned kb = (unsigned)K * 2u;
  // piece = 16 rows x 64 B: lane l -> row (l >> 2), position l & 3 holds logical chunk (l & 3) ^ ((l >> 4) & 2)
  // (PROBE bit 4, timing only, WRONG results: every DMA fetches 8 rows x 128 B -- whole cache lines -- instead of 16 rows
  // x 64 B; the same bytes per K-tile, every line fetched once instead of as two halves one K-tile apart)
  const unsigned voff = (PROBE & 16) ? (unsigned)(lane >> 3) * kb + (unsigned)((lane & 7) << 4)
                                     : (unsigned)(lane >> 2) * kb + (unsigned)(((lane & 3) ^ ((lane >> 4) & 2)) << 4);
  auto piece_off = [&](int p, int kt) -> unsigned {
    if constexpr (PROBE & 16) return ((unsigned)p * 16u + 8u * (unsigned)(kt & 1)) * kb + (unsigned)(kt >> 1) * 128u;
    else return (unsigned)p * 16u * kb + (unsigned)kt * (SBK * 2);
  };
  // this wave's pieces of K-tile kt: A pieces first (NALO, or NALO + 1 for a hi wave), then its two W pieces
  const bool hi = wave < NHI;   // (wave-uniform)
  const int pa0 = hi ? wave * (NALO + 1) : NHI * (NALO + 1) + (wave - NHI) * NALO;   // first A piece of this wave
  // TUNE bit 2: pieces [A 0 .. 2 MI - 1 | W 0 .. 15] dealt as C0 a wave in group 0, the rest evenly over group 1
  constexpr int NP = 2 * MI + 16;
  constexpr int C0 = ((TUNE >> 4) & 7) ? ((TUNE >> 4) & 7) : 6;
  constexpr int R1 = NP - 4 * C0, B1 = R1 / 4, X1 = R1 - 4 * B1;
  static_assert(!(TUNE & 4) || (R1 >= 0 && C0 <= 8), "TUNE: bad piece split");
  constexpr int MAXP = C0 > B1 + 1 ? C0 : B1 + 1;
  const int t_beg = wave < 4 ? wave * C0 : 4 * C0 + (wave - 4) * B1 + min(wave - 4, X1);
  const int t_cnt = wave < 4 ? C0 : B1 + ((wave - 4) < X1 ? 1 : 0);
  // counted waits with immediate operands: this wave's DMAs per K-tile take one of three values (wave-uniform)
  constexpr int OPS_A = (TUNE & 4) ? C0 : OPSHI, OPS_B = (TUNE & 4) ? B1 + 1 : OPSLO, OPS_C = (TUNE & 4) ? B1 : OPSLO;
  const int ops_class = (TUNE & 4) ? (wave < 4 ? 0 : ((wave - 4) < X1 ? 1 : 2)) : ((NHI > 0 && hi) ? 0 : 1);
  // wait until at most (ops per K-tile) * (NB - 2) [+ NST behind an epilogue] vector-memory operations are outstanding
  // (TUNE bit 5: at the waits inside the loop the newest batch is one piece short -- that piece follows the MFMAs)
  auto wait_batches_d = [&](bool with_stores, bool lgkm0, auto defer_c) {
    constexpr int defer = decltype(defer_c)::value;
#define DARK_WB(OPS_)                                                                                     \
    {                                                                                                     \
      if (with_stores) {                                                                                  \
        if (lgkm0) __builtin_amdgcn_s_waitcnt(waitcnt_vm((OPS_) * (NB - 2) + NST - defer) & 0xF0FF);      \
        else __builtin_amdgcn_s_waitcnt(waitcnt_vm((OPS_) * (NB - 2) + NST - defer));                     \
      } else {                                                                                            \
        if (lgkm0) __builtin_amdgcn_s_waitcnt(waitcnt_vm((OPS_) * (NB - 2) - defer) & 0xF0FF);            \
        else __builtin_amdgcn_s_waitcnt(waitcnt_vm((OPS_) * (NB - 2) - defer));                           \
      }                                                                                                   \
    }
    if (ops_class == 0) DARK_WB(OPS_A)
    else if (ops_class == 1) DARK_WB(OPS_B)
    else DARK_WB(OPS_C)
#undef DARK_WB
  };
  auto wait_batches = [&](bool with_stores, bool lgkm0) { wait_batches_d(with_stores, lgkm0, std::integral_constant<int, 0>{}); };
  auto stage = [&](int buf, unsigned a0, unsigned w0, int kt) {
    char* base = smem + buf * S_BUF;
    if constexpr (TUNE & 4) {
#pragma unroll
      for (int q = 0; q < MAXP; ++q) {
        if (q < t_cnt) {
          const int p = t_beg + q;
          const bool isA = p < 2 * MI;
          const int pp = isA ? p : p - 2 * MI;
          const unsigned so = piece_off(pp, kt);
          if (isA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + pp * 1024), 16, voff + (a0 + so), 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + S_OPER + pp * 1024), 16, voff + (w0 + so), 0, 0, 0);
        }
      }
    } else {
    if constexpr (NHI > 0) {
      if (hi) {  // the extra A piece of the hi waves
        const int p = pa0 + NALO;
        const unsigned so = piece_off(p, kt);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + p * 1024), 16, voff + (a0 + so), 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < NALO; ++q) {
      const int p = pa0 + q;
      const unsigned so = piece_off(p, kt);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + p * 1024), 16, voff + (a0 + so), 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = wave * 2 + q;
      const unsigned so = piece_off(p, kt);
      // (the whole offset goes into the VGPR operand: the SGPR offset is EXCLUDED from the descriptor's bounds check)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + S_OPER + p * 1024), 16, voff + (w0 + so), 0, 0, 0);
    }
    }
  };

  // (TUNE bit 3) piece q of this wave's share of a K-tile, one call per DMA: q < NALO -> A piece, then the two W pieces,
  // then the extra A piece of a hi wave; q >= this wave's count does nothing
  constexpr int NQ = (TUNE & 4) ? MAXP : NALO + 2 + (NHI > 0 ? 1 : 0);
  auto stage_piece = [&](int q, int buf, unsigned a0, unsigned w0, int kt) {
    char* base = smem + buf * S_BUF;
    if constexpr (TUNE & 4) {
      if (q < t_cnt) {
        const int p = t_beg + q;
        const bool isA = p < 2 * MI;
        const int pp = isA ? p : p - 2 * MI;
        const unsigned so = piece_off(pp, kt);
        if (isA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + pp * 1024), 16, voff + (a0 + so), 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + S_OPER + pp * 1024), 16, voff + (w0 + so), 0, 0, 0);
      }
    } else {
      if (q < NALO) {
        const int pp = pa0 + q;
        const unsigned so = piece_off(pp, kt);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + pp * 1024), 16, voff + (a0 + so), 0, 0, 0);
      } else if (q < NALO + 2) {
        const int pp = wave * 2 + (q - NALO);
        const unsigned so = piece_off(pp, kt);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + S_OPER + pp * 1024), 16, voff + (w0 + so), 0, 0, 0);
      } else if (NHI > 0 && hi) {
        const int pp = pa0 + NALO;
        const unsigned so = piece_off(pp, kt);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + pp * 1024), 16, voff + (a0 + so), 0, 0, 0);
      }
    }
  };

  const int frow = lane & 15, fq = lane >> 4;
  const int fch = (fq ^ ((frow >> 2) & 2)) << 4;
  const int a_rd = (grp * GROWS + frow) * 64 + fch, w_rd = S_OPER + (wn * 64 + frow) * 64 + fch;

  auto tile_of = [&](int id, int& tm, int& tn) {
    const int per_group = GM * tilesN;
    const int group = id / per_group;
    const int first_m = group * GM;
    const int gsz = min(GM, tilesM - first_m);
    const int rem = id - group * per_group;
    tn = rem / gsz;
    tm = first_m + (rem - tn * gsz);
  };

  const int KT = K / SBK;  // >= 4
  // ---- DMA cursor: the K-tile that is staged next (three phases ahead of the one being read)
  int d_id = wg, d_kt = 0, d_buf = 0;
  unsigned d_a0, d_w0;
  {
    int tm, tn;
    tile_of(d_id, tm, tn);
    // (PROBE bit 5, timing only, WRONG results: every tile streams the SAME row panel and column tile, so that every DMA
    // hits in the XCD's L2: what the launch costs when no piece waits for the Infinity Cache)
    d_a0 = (PROBE & 32) ? 0u : (unsigned)tm * (unsigned)TM * kb;
    d_w0 = (PROBE & 32) ? 0u : (unsigned)tn * 256u * kb;
  }
  bool probe_warm = false;  // (PROBE builds: set once the pipeline is primed)
  auto dma_advance = [&]() {
    d_buf = d_buf + 1 == NB ? 0 : d_buf + 1;
    if (++d_kt == KT) {
      d_kt = 0;
      d_id += G;
      if (d_id < ntiles) {
        int tm, tn;
        tile_of(d_id, tm, tn);
        d_a0 = (PROBE & 32) ? 0u : (unsigned)tm * (unsigned)TM * kb;
        d_w0 = (PROBE & 32) ? 0u : (unsigned)tn * 256u * kb;
      }
    }
  };
  auto stage_next = [&]() {
    if constexpr (PROBE & 1) { if (probe_warm) return; }
    if (d_id < ntiles) stage(d_buf, d_a0, d_w0, d_kt);
    else {  // past the last tile: keep the per-phase DMA count constant (the counted waits rely on it)
      stage(d_buf, 0xF0000000u, 0xF0000000u, 0);  // out of range: nothing is fetched, zeros land in a dead buffer
    }
    d_buf = d_buf + 1 == NB ? 0 : d_buf + 1;
    if (++d_kt == KT) {
      d_kt = 0;
      d_id += G;
      if (d_id < ntiles) {
        int tm, tn;
        tile_of(d_id, tm, tn);
        d_a0 = (PROBE & 32) ? 0u : (unsigned)tm * (unsigned)TM * kb;
        d_w0 = (PROBE & 32) ? 0u : (unsigned)tn * 256u * kb;
      }
    }
  };
#pragma unroll
  for (int i = 0; i < NB - 1; ++i) stage_next();
  probe_warm = true;
  // K-tile 0 has landed (this wave's pieces): NB - 2 younger batches are still in flight
  if constexpr (TUNE != 0) wait_batches(false, false);
  else if (NHI == 0 || !hi) __builtin_amdgcn_s_waitcnt(waitcnt_vm(OPSLO * (NB - 2)));
  else __builtin_amdgcn_s_waitcnt(waitcnt_vm(OPSHI * (NB - 2)));
  // TUNE bit 6 (decoupled groups): per-buffer arrival counters in LDS replace the two workgroup barriers per K-tile --
  // fill[b] counts the waves whose DMA pieces of the K-tile in buffer b have landed, rel[b] the waves that have read
  // it (both monotone: 8 per use of the buffer) -- so that a wave waits for the data it needs and for nothing else.
  lds_u32* const cnt = (lds_u32*)(lds_void*)(smem + S_SMEM + (EPI == EPI_BIAS ? tilesN * 256 * 4 : 0));   // fill[NB], rel[NB]
  if constexpr (TUNE & 64) {
    if (tid < 2 * NB) cnt[tid] = 0u;
  }
  __builtin_amdgcn_s_barrier();
  if constexpr (!(TUNE & 64)) {
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the stagger: group 1 runs one interval behind group 0
  }

  f32x4 acc[P_NI][MI];
  bf16x8 fa[MI], fw[P_NI];
  int rbuf = 0;  // LDS buffer of the K-tile being read

  // epilogue of the tile at (m0, n0): this lane owns row (lane & 15) of each 16-row tile, 4 consecutive columns of
  // each 16-column tile; exactly NST buffer stores per wave (out-of-range ones are dropped by the descriptor)
  auto epilogue = [&](int m0, int n0) {
    if constexpr (PROBE & 4) {  // keep the accumulators alive with one never-taken store
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < P_NI; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) sacc += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
      if (sacc == 12345.678f) __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){0u, 0u, 0u, 0u}, ro, 0u, 0, 0);
      return;
    }
    if constexpr (EPI == EPI_RESID_H) {
      // bf16 residual stream: out = bf16(float(out) + acc), one rounding (gemm.hip EPI_RESID_H).  The old values are
      // fetched in two batches of 16-byte pieces (the accumulators leave room for half of them), each batch waited
      // for in full -- every older DMA has then landed too, so the counted waits behind the epilogue stay valid --
      // and written back in the store layout of the bias flavour: NST buffer stores a wave, out-of-range ones dropped.
      const int hcol = (fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq;
      constexpr int HB = (MI + 1) / 2;
#pragma unroll
      for (int h0 = 0; h0 < MI; h0 += HB) {
        u32x4_t old[HB][2];
        unsigned offs[HB][2];
#pragma unroll
        for (int i = 0; i < HB; ++i) {
          const int mi = h0 + i;
          if (mi >= MI) break;
          const int m = m0 + grp * GROWS + mi * 16 + frow;
#pragma unroll
          for (int np = 0; np < 2; ++np) {
            const int c = n0 + wn * 64 + np * 32 + hcol;
            offs[i][np] = (m < M && c + 8 <= N) ? (unsigned)(((int64_t)m * ldo + c) * 2) : 0xF0000000u;
            old[i][np] = __builtin_amdgcn_raw_buffer_load_b128(ro, offs[i][np], 0, 0);
          }
        }
#pragma unroll
        for (int i = 0; i < HB; ++i) {
          const int mi = h0 + i;
          if (mi >= MI) break;
#pragma unroll
          for (int np = 0; np < 2; ++np) {
            const f32x4 a = acc[2 * np][mi], b = acc[2 * np + 1][mi];
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
              v[e] = __uint_as_float(sw[0]);
              v[4 + e] = __uint_as_float(sw[1]);
            }
            u32x4_t r;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              r[j] = pk_bf16(v[2 * j] + __uint_as_float(old[i][np][j] << 16), v[2 * j + 1] + __uint_as_float(old[i][np][j] & 0xffff0000u));
            __builtin_amdgcn_raw_buffer_store_b128(r, ro, offs[i][np], 0, 0);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + grp * GROWS + mi * 16 + frow;
      if constexpr (EPI == EPI_SWIGLU) {
        float v[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          silu_mul4(acc[2 * p][mi], acc[2 * p + 1][mi], v[p]);
        const uint32_t a_lo = pk_bf16(v[0][0], v[0][1]), a_hi = pk_bf16(v[0][2], v[0][3]);
        const uint32_t b_lo = pk_bf16(v[1][0], v[1][1]), b_hi = pk_bf16(v[1][2], v[1][3]);
        const auto sx = __builtin_amdgcn_permlane16_swap(a_lo, b_lo, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(a_hi, b_hi, false, false);
        const int c = ((n0 + wn * 64) >> 1) + ((fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq);
        const unsigned off = (m < M && c + 8 <= (N >> 1)) ? (unsigned)(((int64_t)m * ldo + c) * 2) : 0xF0000000u;
        __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){sx[0], sy[0], sx[1], sy[1]}, ro, off, 0, 0);
      } else if constexpr (EPI == EPI_BIAS) {
        const float* bl = (const float*)(smem + S_SMEM);  // the whole bias vector, copied to LDS at kernel start
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          float v[2][4];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int ni = 2 * np + t;
            const float4 b4 = *(const float4*)(bl + n0 + wn * 64 + ni * 16 + 4 * fq);
            v[t][0] = acc[ni][mi][0] + b4.x; v[t][1] = acc[ni][mi][1] + b4.y;
            v[t][2] = acc[ni][mi][2] + b4.z; v[t][3] = acc[ni][mi][3] + b4.w;
          }
          const uint32_t a_lo = pk_bf16(v[0][0], v[0][1]), a_hi = pk_bf16(v[0][2], v[0][3]);
          const uint32_t b_lo = pk_bf16(v[1][0], v[1][1]), b_hi = pk_bf16(v[1][2], v[1][3]);
          const auto sx = __builtin_amdgcn_permlane16_swap(a_lo, b_lo, false, false);
          const auto sy = __builtin_amdgcn_permlane16_swap(a_hi, b_hi, false, false);
          const int c = n0 + wn * 64 + np * 32 + ((fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq);
          const unsigned off = (m < M && c + 8 <= N) ? (unsigned)(((int64_t)m * ldo + c) * 2) : 0xF0000000u;
          // (sc1 stores, cache-policy bit 4: the 108 MB of q/k/v leave the XCD's L2 instead of staying as dirty lines
          // that the next round's stores have to evict; attention reads them from the Infinity Cache either way.  -2 %
          // per launch, +0.2 % in the loop in six alternating pairs.  The same bit on the SwiGLU output changes nothing,
          // on the fp32 logits and the residual stream it costs 2 - 2.6 %: their consumers do find them in L2.)
          __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){sx[0], sy[0], sx[1], sy[1]}, ro, off, 0, 16);
        }
      } else if constexpr (EPI == EPI_HEAD) {
        // No logits leave the kernel: the row's 64 columns of this wave (16 per lane, 4 lanes per row) are reduced to
        // {max, column of the max (lowest among equals), how many columns lie within hp_range of it, sum over the
        // other columns of exp2((l - max) * hp_c2)}.  The max itself contributes exactly exp2(0) = 1 to the sum and
        // is taken out afterwards; columns >= N (zero weight rows) count as -inf.
        const int nb = n0 + wn * 64;
        float l[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) l[k] = acc[k >> 2][mi][k & 3];
        if (nb + 64 > N) {  // wave-uniform: only the last column tile
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (nb + (k >> 2) * 16 + 4 * fq + (k & 3) >= N) l[k] = -INFINITY;
        }
        float mxl = l[0];
        int ik = 0;
#pragma unroll
        for (int k = 1; k < 16; ++k) {  // columns grow with k: a strict compare keeps the lowest among equals
          const bool gt = l[k] > mxl;
          mxl = gt ? l[k] : mxl;
          ik = gt ? k : ik;
        }
        int col = (ik >> 2) * 16 + 4 * fq + (ik & 3);
        // the four lanes of a row (l, l + 16, l + 32, l + 48) are combined with the row-swap instructions instead of
        // xor-shuffles through LDS: a swap of x with itself hands every lane BOTH values of its pair (even row's, odd
        // row's), so the pair's result is formed identically in both lanes -- the same values as the shuffle form,
        // which combined own and partner symmetrically -- without an LDS round trip and its wait
        {
          const auto pm = __builtin_amdgcn_permlane16_swap(__float_as_uint(mxl), __float_as_uint(mxl), false, false);
          const auto pc = __builtin_amdgcn_permlane16_swap((unsigned)col, (unsigned)col, false, false);
          const float m0 = __uint_as_float(pm[0]), m1 = __uint_as_float(pm[1]);
          const int c0 = (int)pc[0], c1 = (int)pc[1];
          const bool take = m1 > m0 || (m1 == m0 && c1 < c0);
          mxl = take ? m1 : m0;
          col = take ? c1 : c0;
        }
        {
          const auto pm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxl), __float_as_uint(mxl), false, false);
          const auto pc = __builtin_amdgcn_permlane32_swap((unsigned)col, (unsigned)col, false, false);
          const float m0 = __uint_as_float(pm[0]), m1 = __uint_as_float(pm[1]);
          const int c0 = (int)pc[0], c1 = (int)pc[1];
          const bool take = m1 > m0 || (m1 == m0 && c1 < c0);
          mxl = take ? m1 : m0;
          col = take ? c1 : c0;
        }
        const bool any = mxl > -INFINITY;
        const float thr = mxl - (hp_range + 2e-6f * fabsf(mxl));  // (the slack covers the roundings of l / T, see sample.hip)
        float sum = 0.f;
        int nn = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          sum += __builtin_amdgcn_exp2f((l[k] - mxl) * hp_c2);  // (-inf - finite) * c2 = -inf -> 0
          nn += l[k] > thr ? 1 : 0;
        }
        {
          const auto ps = __builtin_amdgcn_permlane16_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
          const auto pn = __builtin_amdgcn_permlane16_swap((unsigned)nn, (unsigned)nn, false, false);
          sum = __uint_as_float(ps[0]) + __uint_as_float(ps[1]);
          nn = (int)(pn[0] + pn[1]);
        }
        {
          const auto ps = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
          const auto pn = __builtin_amdgcn_permlane32_swap((unsigned)nn, (unsigned)nn, false, false);
          sum = __uint_as_float(ps[0]) + __uint_as_float(ps[1]);
          nn = (int)(pn[0] + pn[1]);
        }
        const float tsum = any ? sum - 1.0f : 0.f;
        const int P = tilesN * 4;
        const unsigned off = (m < M && fq == 0 && nb < N) ? (unsigned)(((int64_t)m * P + (nb >> 6)) * 16) : 0xF0000000u;
        __builtin_amdgcn_raw_buffer_store_b128(
            (u32x4_t){__float_as_uint(mxl), __float_as_uint(tsum), (unsigned)((any ? nn : 0) | (col << 16)), 0u}, ro, off, 0, 0);
      } else {  // EPI_F32
#pragma unroll
        for (int ni = 0; ni < P_NI; ++ni) {
          const int c = n0 + wn * 64 + ni * 16 + 4 * fq;
          const unsigned off = (m < M && c + 4 <= N) ? (unsigned)(((int64_t)m * ldo + c) * 4) : 0xF0000000u;
          const f32x4 a = acc[ni][mi];
          __builtin_amdgcn_raw_buffer_store_b128(
              (u32x4_t){__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])}, ro,
              off, 0, 0);
        }
      }
    }
  };

  if constexpr (EPI == EPI_BIAS) {  // bias -> LDS behind the K-tile buffers (ordinary loads: before the steady state)
    float* bl = (float*)(smem + S_SMEM);
    const int npad = tilesN * 256;
    for (int i = tid; i < npad; i += 512) bl[i] = i < N ? bias[i] : 0.f;
    // (visible to every wave behind the first barrier of the loop below)
  }

  int pm0 = 0, pn0 = 0;   // the tile whose accumulators are waiting for their epilogue
  bool pending = false;
  // (TUNE bit 5) the deferred piece of this wave's batch: its last one (the extra A piece of a hi wave, else the second W piece)
  const int def_q = (NHI > 0 && hi) ? NALO + 2 : NALO + 1;
  unsigned def_a0 = 0, def_w0 = 0;
  int def_kt = 0, def_buf = 0;
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto now = [&]() -> long long {
    if constexpr (PROBE & 8) return (long long)__builtin_readcyclecounter();
    else return 0;
  };
  if constexpr (TUNE & 64) {
    bool dead = false;   // a wait that never ends (a bug) must not hang the GPU: give up on all waits, results are wrong
    auto spin = [&](const lds_u32* c, unsigned target) {
      if (!dead) {
        int guard = 0;
        for (;;) {
          const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)*(const volatile lds_u32*)c);
          if (v >= target) break;
          __builtin_amdgcn_s_sleep(1);
          if (++guard > (1 << 18)) { dead = true; break; }
        }
      }
      asm volatile("" ::: "memory");
    };
    auto signal = [&](lds_u32* c) {
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    lds_u32* const fill = cnt;
    lds_u32* const rel = cnt + NB;
    signal(fill + 0);                       // (this wave's pieces of K-tile 0: waited for above)
    if (grp == 1) spin(fill + 1, 4u);       // the stagger: group 1 starts when group 0 has finished its first LOAD
    unsigned gen8 = 8u;                     // 8 * (uses of buffer rbuf so far, this one included)
    // the two counters an iteration depends on are read one MFMA phase early (the reads return while the MFMAs issue);
    // only a value that is still short sends the wave into the polling loop
    unsigned vf = 0u, vr = 0xffffffffu;     // fill[rbuf]; rel[d_buf] (nothing to release before the first batch)
    for (int id = wg; id < ntiles; id += G) {
      int tm, tn;
      tile_of(id, tm, tn);
      for (int kt = 0; kt < KT; ++kt) {
        if (kt == 0 && pending) epilogue(pm0, pn0);
        const int nbuf = rbuf + 1 == NB ? 0 : rbuf + 1;
        // every wave's pieces of this K-tile are in LDS; every wave has read the K-tile whose buffer (d_buf = rbuf - 1)
        // the next DMA batch overwrites
        if (vf < gen8) spin(fill + rbuf, gen8);
        if (vr < (rbuf == 0 ? gen8 - 8u : gen8)) spin(rel + d_buf, rbuf == 0 ? gen8 - 8u : gen8);
        asm volatile("" ::: "memory");
        {
          const char* as = smem + rbuf * S_BUF + a_rd;
          const char* ws = smem + rbuf * S_BUF + w_rd;
#pragma unroll
          for (int j = 0; j < P_NI; ++j) fw[j] = *(const bf16x8*)(ws + j * 1024);
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(as + i * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        signal(rel + rbuf);   // (LDS operations of a wave execute in order: the add follows the reads)
        stage_next();
        __builtin_amdgcn_sched_barrier(0);
        // this wave's pieces of the NEXT K-tile have landed (counted as in the lock-step loop below)
        wait_batches(pending && kt < NB - 2, true);
        signal(fill + nbuf);
        // next iteration's counters: fill[nbuf] and rel[rbuf] (its d_buf), wanted at gen8' and at this gen8
        vf = *(const volatile lds_u32*)(fill + nbuf);
        vr = *(const volatile lds_u32*)(rel + rbuf);
        __builtin_amdgcn_sched_barrier(0);
        if (kt == 0) {
#pragma unroll
          for (int j = 0; j < P_NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
              acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < P_NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[j][i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        vf = (unsigned)__builtin_amdgcn_readfirstlane((int)vf);
        vr = (unsigned)__builtin_amdgcn_readfirstlane((int)vr);
        rbuf = nbuf;
        if (rbuf == 0) gen8 += 8u;
      }
      pm0 = tm * TM;
      pn0 = tn * 256;
      pending = true;
    }
    epilogue(pm0, pn0);
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    return;
  }
  for (int id = wg; id < ntiles; id += G) {
    int tm, tn;
    tile_of(id, tm, tn);
    for (int kt = 0; kt < KT; ++kt) {
      // ================= LOAD interval =================
      const long long t0 = now();
      if constexpr ((TUNE & 2) && !(TUNE & 8)) __builtin_amdgcn_s_setprio(1);
      bool early_dma = false;
      if constexpr ((TUNE & 26) == 26) {
        // (TUNE 26) behind a finished tile the interval's DMA batch goes out BEFORE the epilogue: 1.3 us of epilogue in
        // front of it would eat most of the batch's two K-tiles of lead
        if (kt == 0 && pending) {
          const bool live = d_id < ntiles;
          const unsigned ca0 = live ? d_a0 : 0xF0000000u, cw0 = live ? d_w0 : 0xF0000000u;
          const int ckt = live ? d_kt : 0;
#pragma unroll
          for (int q = 0; q < NQ; ++q) stage_piece(q, d_buf, ca0, cw0, ckt);
          dma_advance();
          __builtin_amdgcn_sched_barrier(0);
          early_dma = true;
        }
      }
      if (kt == 0 && pending) epilogue(pm0, pn0);   // the partner group's MFMAs run meanwhile
      const long long t0e = now();
      long long t0a = 0, t0b = 0;
      if constexpr (TUNE & 8) {
        // fragment reads (LDS-bound: 1 KiB each at 256 B per clock, four waves at once) and DMA pieces (bound by the
        // texture addresser: 1 KiB each at 64 B per clock) issued alternately, a DMA first, so that the two units work
        // at the same time instead of one after the other
        const char* as = smem + rbuf * S_BUF + a_rd;
        const char* ws = smem + rbuf * S_BUF + w_rd;
        const bool live = d_id < ntiles;
        const unsigned ca0 = live ? d_a0 : 0xF0000000u, cw0 = live ? d_w0 : 0xF0000000u;
        const int ckt = live ? d_kt : 0;
        constexpr int NR = P_NI + MI;
        if ((TUNE & 2) && kt == 0 && pending) {
          // (TUNE 10: behind an epilogue the sequential form -- the reads first, while the epilogue's stores drain)
#pragma unroll
          for (int j = 0; j < P_NI; ++j) fw[j] = *(const bf16x8*)(ws + j * 1024);
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(as + i * 1024);
          __builtin_amdgcn_sched_barrier(0);
          if (!early_dma) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) stage_piece(q, d_buf, ca0, cw0, ckt);
          }
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr ((TUNE & 42) == 42) {
          // (TUNE 42) the four waves of a group place their DMA pieces at different reads (piece j in front of read
          // 3 j + wave % 3), so that they do not all queue at the texture addresser at the same moment; one copy of the
          // unrolled sequence per offset (a run-time piece index costs far more than it saves: 1.75 x slower)
          auto il = [&](auto offc) {
            constexpr int OFF = decltype(offc)::value;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
#pragma unroll
              for (int q = 0; q < NQ; ++q)
                if (3 * q + OFF == r) {
                  stage_piece(q, d_buf, ca0, cw0, ckt);
                  __builtin_amdgcn_sched_barrier(0);
                }
              if (r < P_NI) fw[r] = *(const bf16x8*)(ws + r * 1024);
              else fa[r - P_NI] = *(const bf16x8*)(as + (r - P_NI) * 1024);
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              if (3 * q + OFF >= NR) stage_piece(q, d_buf, ca0, cw0, ckt);
          };
          const int off = wn % 3;
          if (off == 0) il(std::integral_constant<int, 0>{});
          else if (off == 1) il(std::integral_constant<int, 1>{});
          else il(std::integral_constant<int, 2>{});
        } else
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
          for (int q = (r * NQ + NR - 1) / NR; q < ((r + 1) * NQ + NR - 1) / NR; ++q) {
            stage_piece(q, d_buf, ca0, cw0, ckt);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (r < P_NI) fw[r] = *(const bf16x8*)(ws + r * 1024);
          else fa[r - P_NI] = *(const bf16x8*)(as + (r - P_NI) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!early_dma) dma_advance();
        t0a = t0b = now();
      } else if constexpr (TUNE & 32) {
        // one DMA piece of this wave's batch is issued behind its MFMAs instead of here: the LOAD interval (reads, then
        // DMA issue at the TA's 64 B per clock, then the landing wait) is the longer of the two, the MFMA interval has
        // the slack of one piece
        const char* as = smem + rbuf * S_BUF + a_rd;
        const char* ws = smem + rbuf * S_BUF + w_rd;
#pragma unroll
        for (int j = 0; j < P_NI; ++j) fw[j] = *(const bf16x8*)(ws + j * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(as + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
        t0a = now();
        const bool live = d_id < ntiles;
        def_a0 = live ? d_a0 : 0xF0000000u;
        def_w0 = live ? d_w0 : 0xF0000000u;
        def_kt = live ? d_kt : 0;
        def_buf = d_buf;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q != def_q) stage_piece(q, def_buf, def_a0, def_w0, def_kt);
        dma_advance();
        __builtin_amdgcn_sched_barrier(0);
        t0b = now();
      } else {
      if (!(PROBE & 2) || (!pending && kt == 0)) {
        const char* as = smem + rbuf * S_BUF + a_rd;
        const char* ws = smem + rbuf * S_BUF + w_rd;
#pragma unroll
        for (int j = 0; j < P_NI; ++j) fw[j] = *(const bf16x8*)(ws + j * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(as + i * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      t0a = now();
      stage_next();  // K-tile NB - 1 phases ahead, into the buffer both groups finished reading one interval ago
      __builtin_amdgcn_sched_barrier(0);
      t0b = now();
      }
      // this wave's pieces of the NEXT K-tile have landed: its batch is followed by NB - 2 younger batches (OPS0 / OPS1
      // DMAs each) and, for the NB - 2 phases behind an epilogue, by that epilogue's NST stores (the batch of K-tile
      // kt + 1 was issued NB - 2 intervals before this one; the stores at kt = 0, in front of that interval's batch)
      long long t1 = 0, t2 = 0;
      if constexpr (PROBE & 8) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only: the fragment reads have returned
        t1 = now();
      }
      const bool behind_epi = pending && kt < (((TUNE & 26) == 26) ? NB - 1 : NB - 2);   // (TUNE 26: the stores follow that interval's batch)
      if constexpr ((TUNE & 32) && !(TUNE & 8)) {
        wait_batches_d(behind_epi, true, std::integral_constant<int, 1>{});
      } else if constexpr (TUNE != 0) {
        if ((TUNE & 1) && grp == 0) __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0); the DMA wait follows the MFMAs
        else wait_batches(behind_epi, true);
      } else if (NHI == 0 || !hi) {
        if (pending && kt < NB - 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm(OPSLO * (NB - 2) + NST) & 0xF0FF);  // + lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(waitcnt_vm(OPSLO * (NB - 2)) & 0xF0FF);
      } else {
        if (pending && kt < NB - 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm(OPSHI * (NB - 2) + NST) & 0xF0FF);
        else __builtin_amdgcn_s_waitcnt(waitcnt_vm(OPSHI * (NB - 2)) & 0xF0FF);
      }
      if constexpr (PROBE & 8) t2 = now();
      if constexpr ((TUNE & 2) && !(TUNE & 8)) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const long long t3 = now();
      // ================= MFMA interval =================
      if constexpr (PROBE & 128) {  // timing probe: data movement only (one MFMA keeps the fragments alive)
        // (the fragments are kept alive by empty asm operands: a run-time register index would send them to scratch)
#pragma unroll
        for (int j = 0; j < P_NI; ++j) asm volatile("" ::"v"(fw[j]));
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(fa[i]));
        if (kt == 0) {
#pragma unroll
          for (int j = 0; j < P_NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      } else if (kt == 0) {
#pragma unroll
        for (int j = 0; j < P_NI; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < P_NI; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[j][i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((TUNE & 32) && !(TUNE & 8)) {
        stage_piece(def_q, def_buf, def_a0, def_w0, def_kt);
        __builtin_amdgcn_sched_barrier(0);
      }
      const long long t4 = now();
      if constexpr (TUNE & 1) {
        // group 0's pieces of K-tile kt + 1: first read (by group 0 itself) behind the barrier below, which group 1
        // passes only after its own wait for the same K-tile at the end of its LOAD interval
        if (grp == 0) wait_batches(behind_epi, false);
      }
      __builtin_amdgcn_s_barrier();
      if constexpr (PROBE & 8) {
        const long long t5 = now();
        pt[0] += t1 - t0; pt[1] += t2 - t1; pt[2] += t3 - t2; pt[3] += t4 - t3; pt[4] += t5 - t4; pt[5] += 1;
        pt[6] += t0a - t0e; pt[7] += t0b - t0a;   // fragment-read issue; DMA issue (both inside pt[0], as is the epilogue)
      }
      rbuf = rbuf + 1 == NB ? 0 : rbuf + 1;
    }
    pm0 = tm * TM;
    pn0 = tn * 256;
    pending = true;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // matches group 1's last interval
  if constexpr (PROBE & 8) {
    if (lane == 0 && bias) {
      long long* o = (long long*)bias + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = pt[i];
    }
  }
  epilogue(pm0, pn0);
  __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
}

// Can this shape go to the persistent 256-wide kernel?  (16-byte output pieces, K tiles of 64, 32-bit buffer sizes.)
bool p256_supports(int epi, int M, int N, int K, int64_t ldo, const void* out, const void* bias) {
  if (epi != EPI_SWIGLU && epi != EPI_BIAS && epi != EPI_F32 && epi != EPI_RESID_H) return false;
  if (K % 64 != 0 || K < 128) return false;
  const int per16 = epi == EPI_F32 ? 4 : 8;
  const int ncols = epi == EPI_SWIGLU ? N / 2 : N;
  if (ldo % per16 != 0 || ncols % per16 != 0 || ((uintptr_t)out & 15) != 0) return false;
  if (epi == EPI_SWIGLU && N % 32 != 0) return false;
  if ((int64_t)M * K * 2 >= (1ll << 31) || ((int64_t)N + 256) * K * 2 >= (1ll << 31)) return false;
  return true;
}

static std::atomic<int> g_p256_nodma{0};
void set_p256_nodma(int v) { g_p256_nodma.store(v); }

int launch_gemm_p256(int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                     int64_t ldo, int w_rows, hipStream_t st) {
  DARK_CHECK_ARG(epi != EPI_RESID_H && p256_supports(epi, M, N, K, ldo, out, bias), "gemm_p256: unsupported shape M=%d N=%d K=%d", M, N, K);
  const int tilesM = cdiv(M, PBM), tilesN = cdiv(N, PBN);
  const int ntiles = tilesM * tilesN;
  int dev = 0, cus = 256;
  DARK_CHECK_HIP(hipGetDevice(&dev));
  static std::atomic<int> cu_cache[DARK_MAX_DEVICES] = {};
  if (dev >= 0 && dev < DARK_MAX_DEVICES) {
    cus = cu_cache[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
      DARK_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      cu_cache[dev].store(cus, std::memory_order_relaxed);
    }
  }
  const int grid = min(ntiles, cus);
  const unsigned a_bytes = (unsigned)((int64_t)M * K * 2), w_bytes = (unsigned)((int64_t)w_rows * K * 2);
#define DARK_P256(E)                                                                                               \
  {                                                                                                                \
    static SmemOptIn optin;                                                                                        \
    DARK_TRY(optin.ensure((const void*)gemm_bf16_p256<E>, P_SMEM, st));                                            \
    hipLaunchKernelGGL((gemm_bf16_p256<E>), dim3(grid), dim3(512), P_SMEM, st, (const bf16_t*)A, (const bf16_t*)W, \
                       bias, out, M, N, K, ldo, tilesM, tilesN, ntiles, a_bytes, w_bytes);                         \
  }
#ifdef DARK_PROBES   // (timing-only instantiation with WRONG results: the probe build only, see the Makefile)
  if (epi == EPI_SWIGLU && g_p256_nodma.load()) {
    static SmemOptIn optin;
    DARK_TRY(optin.ensure((const void*)gemm_bf16_p256<EPI_SWIGLU, true>, P_SMEM, st));
    hipLaunchKernelGGL((gemm_bf16_p256<EPI_SWIGLU, true>), dim3(grid), dim3(512), P_SMEM, st, (const bf16_t*)A,
                       (const bf16_t*)W, bias, out, M, N, K, ldo, tilesM, tilesN, ntiles, a_bytes, w_bytes);
  } else
#endif
  if (epi == EPI_SWIGLU) DARK_P256(EPI_SWIGLU)
  else if (epi == EPI_BIAS) DARK_P256(EPI_BIAS)
  else DARK_P256(EPI_F32)
#undef DARK_P256
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

bool s256_supports(int epi, int M, int N, int K, int64_t ldo, const void* out, const void* bias) {
  if (!p256_supports(epi, M, N, K, ldo, out, bias) || K < 128) return false;
  const int64_t es = epi == EPI_F32 ? 4 : 2;
  if (((int64_t)M + 320) * ldo * es >= 0xE0000000ll) return false;  // 32-bit byte offsets of the buffer stores
  if (epi == EPI_BIAS && (cdiv(N, 256) * 256 * 4 > 160 * 1024 - s_smem(EPI_BIAS, 8) || ((uintptr_t)bias & 3))) return false;
  return true;
}

static std::atomic<int> g_s256_probe{0};
void set_s256_probe(int v) { g_s256_probe.store(v); }
static std::atomic<int> g_s256_rows{0};   // 0 = automatic, 256 / 288 / 320 = forced tile height (A/B hook)
void set_s256_rows(int v) { g_s256_rows.store(v); }
static std::atomic<int> g_s256_interleave{1};   // interleaved LOAD interval for the bias / residual flavours (variant 14 = off)
void set_s256_interleave(int v) { g_s256_interleave.store(v); }
static std::atomic<int> g_s256_tune{0};   // TUNE of gemm_bf16_s256 (A/B hook; the instantiated values only)
void set_s256_tune(int v) { g_s256_tune.store(v); }

static int cu_count(int* cus) {   // CUs of the current device (cached per device: this sits on every launch path)
  static std::atomic<int> cache[DARK_MAX_DEVICES] = {};
  int dev = 0;
  DARK_CHECK_HIP(hipGetDevice(&dev));
  const bool cached = dev >= 0 && dev < DARK_MAX_DEVICES;
  int n = cached ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (n == 0) {
    DARK_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    if (cached) cache[dev].store(n, std::memory_order_relaxed);
  }
  *cus = n;
  return DARK_OK;
}

// 16-row MFMA tiles per wave: 8 (256 x 256 tile), 9 (288 x 256) or 10 (320 x 256).  Per row the heights run at the
// same rate within 2 % (tools/gemm_ab.py --variants 8,10: the fewer bytes per MAC of the taller tiles pay for their four
// instead of five LDS buffers), so the choice is tile quantisation: the height that needs the fewest row-rounds over
// the CUs (M = 23552: gate/up 7 rounds of 320; q/k/v 3 rounds of 288; N = 768 one round of 246 tiles of 288).
static double s256_cost(int mi, int M, int tilesN, int cus) {
  return (double)cdiv(cdiv(M, 32 * mi) * tilesN, cus) * mi * (mi > 8 ? 0.98 : 1.0);
}
static int s256_pick_mi(int epi, int M, int tilesN, int cus) {
  // (the bias flavour keeps its bias vector behind the K-tile buffers: the taller tiles leave it 16 KiB)
  if (epi == EPI_BIAS && tilesN * 256 * 4 > 160 * 1024 - s_smem(EPI_BIAS, 10)) return 8;
  const int forced = g_s256_rows.load(std::memory_order_relaxed);
  if (forced == 256) return 8;
  if (forced == 288) return 9;
  if (forced == 320) return 10;
  int best = 8;
  for (int mi = 9; mi <= 10; ++mi)
    if (s256_cost(mi, M, tilesN, cus) < s256_cost(best, M, tilesN, cus)) best = mi;
  return best;
}
// what the automatic choice would cost, in row-rounds (pick_kernel of gemm.hip compares flavours with it)
double s256_best_cost(int epi, int M, int N, int* cus_out) {
  int cus = 256;
  if (cu_count(&cus) != DARK_OK || cus <= 0) cus = 256;   // (the launch itself reports a HIP error properly)
  if (cus_out) *cus_out = cus;
  const int tilesN = cdiv(N, 256);
  const int mi = s256_pick_mi(epi, M, tilesN, cus);
  return s256_cost(mi, M, tilesN, cus) * 32.0;
}

// one launch of gemm_bf16_s256<E, PB, MI_> with everything the kernel derives from the tile height
#define DARK_S256_LAUNCH_T(E, PB, MI_, TU, BIAS_PTR, MDEV, C2, RANGE)                                                 \
  {                                                                                                                   \
    const int tilesM = cdiv(M, 32 * MI_), ntiles = tilesM * tilesN, grid = min(ntiles, cus);                          \
    const int smem = s_smem(E, MI_) + (E == EPI_BIAS ? tilesN * 256 * 4 : 0) + (((TU) & 64) ? 64 : 0);                \
    static SmemOptIn optin;                                                                                           \
    DARK_TRY(optin.ensure((const void*)gemm_bf16_s256<E, PB, MI_, TU>, smem, st));                                    \
    hipLaunchKernelGGL((gemm_bf16_s256<E, PB, MI_, TU>), dim3(grid), dim3(512), smem, st, (const bf16_t*)A,           \
                       (const bf16_t*)W, BIAS_PTR, out, M, N, K, ldo, tilesM, tilesN, ntiles, a_bytes, w_bytes, o_bytes, \
                       MDEV, C2, RANGE);                                                                              \
  }
#define DARK_S256_LAUNCH(E, PB, MI_, BIAS_PTR, MDEV, C2, RANGE) DARK_S256_LAUNCH_T(E, PB, MI_, 0, BIAS_PTR, MDEV, C2, RANGE)
#define DARK_S256_EITHER(E, MDEV, C2, RANGE)                                \
  if (mi == 10) DARK_S256_LAUNCH(E, 0, 10, bias, MDEV, C2, RANGE)           \
  else if (mi == 9) DARK_S256_LAUNCH(E, 0, 9, bias, MDEV, C2, RANGE)        \
  else DARK_S256_LAUNCH(E, 0, 8, bias, MDEV, C2, RANGE)

int launch_gemm_s256(int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                     int64_t ldo, int w_rows, hipStream_t st) {
  DARK_CHECK_ARG(s256_supports(epi, M, N, K, ldo, out, bias), "gemm_s256: unsupported shape M=%d N=%d K=%d", M, N, K);
  const int tilesN = cdiv(N, 256);
  int cus = 0;
  DARK_TRY(cu_count(&cus));
  const int64_t es = epi == EPI_F32 ? 4 : 2;
  const unsigned a_bytes = (unsigned)((int64_t)M * K * 2), w_bytes = (unsigned)((int64_t)w_rows * K * 2);
  const unsigned o_bytes = (unsigned)((int64_t)M * ldo * es);
  const int mi = s256_pick_mi(epi, M, tilesN, cus);
  const int probe = epi == EPI_SWIGLU ? g_s256_probe.load() : 0;   // timing probes: gate/up flavour, 256-row tile
  const int tune = g_s256_tune.load(std::memory_order_relaxed);
  const int* no_mdev = nullptr;
  const bool interleave = g_s256_interleave.load(std::memory_order_relaxed) && mi >= 9;   // (see the product launches below)
#ifndef DARK_PROBES
  // the shipped library holds the product instantiations only; timing probes (some give WRONG results) and the TUNE
  // A/B forms exist in the probe build (make probes -> libdark_hip_probes.so, loaded by tools/ when DARK_HIP_LIB names it)
  if (probe != 0 || tune != 0) {
    set_error("gemm_s256: probe %d / TUNE %d need the probe build of the library (make -C dark_amd/csrc probes)", probe, tune);
    return DARK_ERR_UNSUPPORTED;
  }
#else
  if (probe == 1) DARK_S256_LAUNCH(EPI_SWIGLU, 1, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 2) DARK_S256_LAUNCH(EPI_SWIGLU, 2, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 3) DARK_S256_LAUNCH(EPI_SWIGLU, 3, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 4) DARK_S256_LAUNCH(EPI_SWIGLU, 4, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 7) DARK_S256_LAUNCH(EPI_SWIGLU, 7, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 8) DARK_S256_LAUNCH(EPI_SWIGLU, 8, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 128) DARK_S256_LAUNCH(EPI_SWIGLU, 128, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 132) DARK_S256_LAUNCH(EPI_SWIGLU, 132, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 16) DARK_S256_LAUNCH(EPI_SWIGLU, 16, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 144) DARK_S256_LAUNCH(EPI_SWIGLU, 144, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 148) DARK_S256_LAUNCH(EPI_SWIGLU, 148, 8, bias, no_mdev, 0.f, 0.f)
  else if (probe == 9) DARK_S256_LAUNCH(EPI_SWIGLU, 8, 10, bias, no_mdev, 0.f, 0.f)   // (per-phase sums of the 320-row tile)
  else if (probe == 32) DARK_S256_LAUNCH(EPI_SWIGLU, 32, 10, bias, no_mdev, 0.f, 0.f)  // (every DMA an L2 hit, 320-row tile)
  else if (probe == 40) DARK_S256_LAUNCH(EPI_SWIGLU, 40, 10, bias, no_mdev, 0.f, 0.f)  // (the same with per-phase sums)
  else if (probe == 24 + 64) DARK_S256_LAUNCH_T(EPI_SWIGLU, 8, 10, 1, bias, no_mdev, 0.f, 0.f)   // (per-phase sums, TUNE 1)
  else if (probe == 24 + 65) DARK_S256_LAUNCH_T(EPI_SWIGLU, 8, 10, 5, bias, no_mdev, 0.f, 0.f)   // (per-phase sums, TUNE 5)
  else if (probe == 24 + 66) DARK_S256_LAUNCH_T(EPI_SWIGLU, 8, 10, 8, bias, no_mdev, 0.f, 0.f)   // (per-phase sums, TUNE 8)
  // (tune 16: TUNE 8 for the bias and residual flavours only -- the SwiGLU flavour is slower with it)
  else if (tune != 0 && !(tune == 16 && epi == EPI_SWIGLU) && mi >= 9 && (epi == EPI_SWIGLU || epi == EPI_BIAS || epi == EPI_RESID_H)) {
#define DARK_S256_TUNED(TU)                                                                     \
    if (epi == EPI_SWIGLU) {                                                                    \
      if (mi == 10) DARK_S256_LAUNCH_T(EPI_SWIGLU, 0, 10, TU, bias, no_mdev, 0.f, 0.f)          \
      else DARK_S256_LAUNCH_T(EPI_SWIGLU, 0, 9, TU, bias, no_mdev, 0.f, 0.f)                    \
    } else if (epi == EPI_BIAS) {                                                               \
      if (mi == 10) DARK_S256_LAUNCH_T(EPI_BIAS, 0, 10, TU, bias, no_mdev, 0.f, 0.f)            \
      else DARK_S256_LAUNCH_T(EPI_BIAS, 0, 9, TU, bias, no_mdev, 0.f, 0.f)                      \
    } else {                                                                                    \
      if (mi == 10) DARK_S256_LAUNCH_T(EPI_RESID_H, 0, 10, TU, bias, no_mdev, 0.f, 0.f)         \
      else DARK_S256_LAUNCH_T(EPI_RESID_H, 0, 9, TU, bias, no_mdev, 0.f, 0.f)                   \
    }
    if (tune == 1) { DARK_S256_TUNED(1) }
    else if (tune == 2) { DARK_S256_TUNED(2) }
    else if (tune == 5) { DARK_S256_TUNED(5) }
    else if (tune == 4) { DARK_S256_TUNED(4) }
    else if (tune == 8 || tune == 16) { DARK_S256_TUNED(8) }
    else if (tune == 10) { DARK_S256_TUNED(10) }
    else if (tune == 26) { DARK_S256_TUNED(26) }
    else if (tune == 42) { DARK_S256_TUNED(42) }
    else if (tune == 9) { DARK_S256_TUNED(9) }
    else if (tune == 5 + (5 << 4)) { DARK_S256_TUNED(85) }
    else if (tune == 5 + (7 << 4)) { DARK_S256_TUNED(117) }
    else if (tune == 64) { DARK_S256_TUNED(64) }
    else if (tune == 32) { DARK_S256_TUNED(32) }
    else { set_error("gemm_s256: TUNE %d is not instantiated", tune); return DARK_ERR_ARG; }
#undef DARK_S256_TUNED
  }
  else
#endif
  // On the taller tiles the LOAD interval issues fragment reads and DMA pieces alternately (the LDS and the texture
  // addresser work at the same time), except behind an epilogue, where the reads go first while the epilogue's stores
  // drain (TUNE 10).  Bit-identical; in the loop +0.5 % for the bias and residual flavours (profiles/r03zf_*) and a
  // further +0.3 % with the SwiGLU flavour included (r03zk_*), each in three alternating pairs; dominant launch -1 %.
  // dark_dbg_set_gemm_variant(14) switches it off.
  if (interleave && epi == EPI_SWIGLU) {
    if (mi == 10) DARK_S256_LAUNCH_T(EPI_SWIGLU, 0, 10, 10, bias, no_mdev, 0.f, 0.f)
    else DARK_S256_LAUNCH_T(EPI_SWIGLU, 0, 9, 10, bias, no_mdev, 0.f, 0.f)
  } else if (epi == EPI_SWIGLU) { DARK_S256_EITHER(EPI_SWIGLU, no_mdev, 0.f, 0.f) }
  else if (epi == EPI_BIAS && interleave) {
    if (mi == 10) DARK_S256_LAUNCH_T(EPI_BIAS, 0, 10, 10, bias, no_mdev, 0.f, 0.f)
    else DARK_S256_LAUNCH_T(EPI_BIAS, 0, 9, 10, bias, no_mdev, 0.f, 0.f)
  } else if (epi == EPI_RESID_H && interleave) {
    if (mi == 10) DARK_S256_LAUNCH_T(EPI_RESID_H, 0, 10, 10, bias, no_mdev, 0.f, 0.f)
    else DARK_S256_LAUNCH_T(EPI_RESID_H, 0, 9, 10, bias, no_mdev, 0.f, 0.f)
  }
  else if (epi == EPI_BIAS) { DARK_S256_EITHER(EPI_BIAS, no_mdev, 0.f, 0.f) }
  else if (epi == EPI_RESID_H) { DARK_S256_EITHER(EPI_RESID_H, no_mdev, 0.f, 0.f) }
  else { DARK_S256_EITHER(EPI_F32, no_mdev, 0.f, 0.f) }
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ---- fused LM head: the EPI_HEAD flavour (slice records instead of logits) and the plain fp32 flavour with a
// device-side row count (the rows the records could not finish)
bool head_gemm_supports(int M, int N, int K) {
  if (M <= 0 || N < 256 || K % 64 != 0 || K < 128) return false;
  if ((int64_t)M * K * 2 >= (1ll << 31) || ((int64_t)N + 256) * K * 2 >= (1ll << 31)) return false;
  if (((int64_t)M + 320) * (cdiv(N, 256) * 4) * 16 >= 0xE0000000ll) return false;
  return true;
}

int launch_gemm_head(const void* A, const void* W, void* records, int M, int N, int K, float c2, float range,
                     const int* m_dev, int w_rows, hipStream_t st) {
  DARK_CHECK_ARG(A && W && records && head_gemm_supports(M, N, K), "gemm_head: unsupported shape M=%d N=%d K=%d", M, N, K);
  DARK_CHECK_ARG(((uintptr_t)records & 15) == 0 && c2 > 0.f && range > 0.f, "gemm_head: bad argument");
  const int tilesN = cdiv(N, 256);
  int cus = 0;
  DARK_TRY(cu_count(&cus));
  const unsigned a_bytes = (unsigned)((int64_t)M * K * 2), w_bytes = (unsigned)((int64_t)w_rows * K * 2);
  const unsigned o_bytes = (unsigned)((int64_t)M * tilesN * 4 * 16);
  const int mi = s256_pick_mi(EPI_HEAD, M, tilesN, cus);
  const float* bias = nullptr;
  void* out = records;
  const int64_t ldo = 0;
  DARK_S256_EITHER(EPI_HEAD, m_dev, c2, range)
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int launch_gemm_s256_devrows(const void* A, const void* W, void* out, int M, int N, int K, int64_t ldo, const int* m_dev,
                             int w_rows, hipStream_t st) {
  DARK_CHECK_ARG(s256_supports(EPI_F32, M, N, K, ldo, out, nullptr) && m_dev, "gemm_s256_devrows: unsupported shape M=%d N=%d K=%d",
                 M, N, K);
  const int tilesN = cdiv(N, 256);
  int cus = 0;
  DARK_TRY(cu_count(&cus));
  const unsigned a_bytes = (unsigned)((int64_t)M * K * 2), w_bytes = (unsigned)((int64_t)w_rows * K * 2);
  const unsigned o_bytes = (unsigned)((int64_t)M * ldo * 4);
  const int mi = s256_pick_mi(EPI_F32, M, tilesN, cus);
  const float* bias = nullptr;
  DARK_S256_EITHER(EPI_F32, m_dev, 0.f, 0.f)
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}
#undef DARK_S256_EITHER
#undef DARK_S256_LAUNCH
#undef DARK_S256_LAUNCH_T

}  // namespace dark
