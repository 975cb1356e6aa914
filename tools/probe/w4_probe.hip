// Skeleton of a ONE-WAVE-PER-SIMD persistent bf16 GEMM main loop: C[M,N] = A[M,K] * W[N,K]^T with a
// TM x 256 tile per 256-thread workgroup, 4 waves in 2 x 2 (each (TM/2) x 128: accumulators in the unified 512-entry
// register file), 32-deep K-tiles in an LDS ring of NB buffers filled by LDS-DMA, ONE barrier per K-tile, fragments
// double-buffered in registers (the reads of K-tile t+1 and the DMA of K-tile t+NB are issued between the MFMAs of
// K-tile t).  Purpose: measure what this structure costs per K-tile BEFORE it is built into the library, against the
// two-waves-per-SIMD kernel that runs today (gemm_bf16_s256: 925 - 1030 ns per 320 x 256 x 32 K-tile).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 w4_probe.hip -o w4_probe ; run: ./w4_probe [M] [N] [K]
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
constexpr int waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }  // vmcnt(n), nothing else
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}

// MODE bits (timing ablations; results are wrong unless MODE == 0): 1 = no DMA in the steady state, 2 = no fragment reads
// in the steady state, 4 = no MFMAs (one per K-tile), 8 = no output stores
template <int MI, int NB, int MODE, bool STAG = true>
__global__ __launch_bounds__(256) void gemm_w4(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                               bf16_t* __restrict__ out, int M, int N, int K, int tilesM, int tilesN,
                                               int ntiles, unsigned a_bytes, unsigned w_bytes, unsigned o_bytes) {
  constexpr int TM = 32 * MI;            // tile rows; a wave owns 16 * MI rows x 128 columns
  constexpr int NJ = 8;                  // 16-column MFMA tiles per wave
  constexpr int A_OPER = TM * 64, W_OPER = 256 * 64, BUF = A_OPER + W_OPER;
  constexpr int NPA = TM / 16, NPW = 16, NPIECE = NPA + NPW;   // DMA pieces (16 rows x 64 B) per K-tile
  static_assert(NPIECE % 4 == 0, "pieces must divide over the four waves");
  constexpr int PPW = NPIECE / 4;        // pieces per wave and K-tile
  constexpr int GM = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  if (wg >= ntiles) return;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)o_bytes, 0x00020000);
  const unsigned kb = (unsigned)K * 2u;
  // piece = 16 rows x 64 B: lane l -> row (l >> 2), position l & 3 holds logical chunk (l & 3) ^ ((l >> 4) & 2)
  const unsigned voff = (unsigned)(lane >> 2) * kb + (unsigned)(((lane & 3) ^ ((lane >> 4) & 2)) << 4);

  auto tile_of = [&](int id, int& tm, int& tn) {
    const int per_group = GM * tilesN;
    const int group = id / per_group;
    const int first_m = group * GM;
    const int gsz = min(GM, tilesM - first_m);
    const int rem = id - group * per_group;
    tn = rem / gsz;
    tm = first_m + (rem - tn * gsz);
  };
  const int KT = K / 32;
  // ---- DMA cursor (runs NB K-tiles ahead of the reads, across output tiles)
  int d_id = wg, d_kt = 0, d_buf = 0;
  unsigned d_a0, d_w0;
  {
    int tm, tn;
    tile_of(d_id, tm, tn);
    d_a0 = (unsigned)tm * (unsigned)TM * kb;
    d_w0 = (unsigned)tn * 256u * kb;
  }
  // piece q (0 .. PPW-1) of this wave: global piece p = wave * PPW + q; p < NPA -> A piece p, else W piece p - NPA
  auto dma_piece = [&](int q) {
    const int p = wave * PPW + q;
    const bool live = d_id < ntiles;
    char* base = smem + d_buf * BUF;
    if (p < NPA) {
      const unsigned off = live ? d_a0 + (unsigned)p * 16u * kb + (unsigned)d_kt * 64u + voff : 0xF0000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + p * 1024), 16, off, 0, 0, 0);
    } else {
      const int pw = p - NPA;
      const unsigned off = live ? d_w0 + (unsigned)pw * 16u * kb + (unsigned)d_kt * 64u + voff : 0xF0000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + A_OPER + pw * 1024), 16, off, 0, 0, 0);
    }
  };
  auto dma_advance = [&]() {
    d_buf = d_buf + 1 == NB ? 0 : d_buf + 1;
    if (++d_kt == KT) {
      d_kt = 0;
      d_id += G;
      if (d_id < ntiles) {
        int tm, tn;
        tile_of(d_id, tm, tn);
        d_a0 = (unsigned)tm * (unsigned)TM * kb;
        d_w0 = (unsigned)tn * 256u * kb;
      }
    }
  };

  const int frow = lane & 15, fq = lane >> 4;
  const int fch = (fq ^ ((frow >> 2) & 2)) << 4;
  const int a_rd = (wr * (TM / 2) + frow) * 64 + fch, w_rd = A_OPER + (wc * 128 + frow) * 64 + fch;

  f32x4 acc[NJ][MI];
  bf16x8 xa[MI], xw[NJ], ya[MI], yw[NJ];

  // prologue: NB batches in flight; K-tiles 0 and 1 must have landed before the first barrier
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) dma_piece(q);
    dma_advance();
  }
  __builtin_amdgcn_s_waitcnt(waitcnt_vm(PPW * (NB - 2)));
  __builtin_amdgcn_s_barrier();
  {
    const char* as = smem + a_rd;
    const char* ws = smem + w_rd;
#pragma unroll
    for (int j = 0; j < NJ; ++j) xw[j] = *(const bf16x8*)(ws + j * 1024);
#pragma unroll
    for (int i = 0; i < MI; ++i) xa[i] = *(const bf16x8*)(as + i * 1024);
  }
  // the whole main loop once per wave number: the stagger positions are compile-time constants of each copy (a run-time
  // branch around a fragment read makes hipcc wait for the read at the join: measured 2.3x slower)
  auto main_loop = [&](auto WVC) {
  constexpr int WV = decltype(WVC)::value;
  int rbuf = 1;   // buffer of the K-tile whose fragments are read NEXT
  bool warm = false;

  // one K-tile: MFMAs on (ca, cw), the reads of the next K-tile into (na, nw) and this wave's DMA pieces between them
  auto ktile = [&](auto FIRST, bf16x8 (&ca)[MI], bf16x8 (&cw)[NJ], bf16x8 (&na)[MI], bf16x8 (&nw)[NJ]) {
    constexpr bool first = decltype(FIRST)::value;
    // every wave's pieces of the next K-tile have landed and its reads of the current one have returned (waits at
    // the end of the previous K-tile): behind the barrier the next K-tile may be read and the current buffer refilled
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const char* as = smem + rbuf * BUF + a_rd;
    const char* ws = smem + rbuf * BUF + w_rd;
    constexpr int NR = NJ + MI;   // fragment reads per K-tile
    if constexpr (STAG) {
      // time stagger: wave w starts its K-tile 32 w cycles late, so that the four waves' reads and DMA pieces (same code,
      // same program positions) reach the LDS and the texture addresser 32 cycles apart instead of together
      for (int d = 0; d < wave; ++d) asm volatile("s_nop 15\n\ts_nop 15");
    }
    #pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!(MODE & 2) || !warm) {
#pragma unroll
        for (int r = (j * NR) / NJ; r < ((j + 1) * NR) / NJ; ++r) {
          if (r < NJ) nw[r] = *(const bf16x8*)(ws + r * 1024);
          else na[r - NJ] = *(const bf16x8*)(as + (r - NJ) * 1024);
        }
      }
      if (!(MODE & 1) || !warm) {
#pragma unroll
        for (int q = (j * PPW) / NJ; q < ((j + 1) * PPW) / NJ; ++q) dma_piece(q);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(MODE & 4) || j == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if constexpr (first) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cw[j], ca[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cw[j], ca[i], acc[j][i], 0, 0, 0);
        }
      } else if (first) {
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    dma_advance();
    rbuf = rbuf + 1 == NB ? 0 : rbuf + 1;
    // the K-tile after next has landed (NB - 2 younger batches may stay in flight); this K-tile's reads have returned
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(PPW * (NB - 2)) & 0xF0FF);
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int id = wg; id < ntiles; id += G) {
    int tm, tn;
    tile_of(id, tm, tn);
    ktile(std::true_type{}, xa, xw, ya, yw);
    warm = true;
    ktile(std::false_type{}, ya, yw, xa, xw);
    for (int kt = 2; kt < KT; kt += 2) {   // KT is even
      ktile(std::false_type{}, xa, xw, ya, yw);
      ktile(std::false_type{}, ya, yw, xa, xw);
    }
    // ---- epilogue (probe: plain bf16 store of the accumulators; NST stores a wave)
    if constexpr (!(MODE & 8)) {
      const int m0 = tm * TM + wr * (TM / 2), n0 = tn * 256 + wc * 128;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + i * 16 + frow;
#pragma unroll
        for (int np = 0; np < NJ / 2; ++np) {
          const f32x4 a = acc[2 * np][i], b = acc[2 * np + 1][i];
          const uint32_t a_lo = pk_bf16(a[0], a[1]), a_hi = pk_bf16(a[2], a[3]);
          const uint32_t b_lo = pk_bf16(b[0], b[1]), b_hi = pk_bf16(b[2], b[3]);
          const auto sx = __builtin_amdgcn_permlane16_swap(a_lo, b_lo, false, false);
          const auto sy = __builtin_amdgcn_permlane16_swap(a_hi, b_hi, false, false);
          const int c = n0 + np * 32 + ((fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq);
          const unsigned off = (m < M && c + 8 <= N) ? (unsigned)(((int64_t)m * N + c) * 2) : 0xF0000000u;
          __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){sx[0], sy[0], sx[1], sy[1]}, ro, off, 0, 0);
        }
      }
      // the stores sit in front of the DMA batches in this wave's queue: drain them so that the counted waits of the
      // next tile stay exact (probe: simplest form)
      __builtin_amdgcn_s_waitcnt(waitcnt_vm(PPW * (NB - 2)));
    } else {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) s += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
      if (s == 12345.678f) __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){0u, 0u, 0u, 0u}, ro, 0u, 0, 0);
    }
  }
  };
  main_loop(std::integral_constant<int, 0>{});
  __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
}

__global__ void ref_kernel(const bf16_t* A, const bf16_t* W, float* ref, const int* rows, const int* cols, int n, int K) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(size_t)rows[t] * K + k] * (float)W[(size_t)cols[t] * K + k];
  ref[t] = s;
}

template <int MI, int NB, int MODE, bool STAG = true>
static double run(const bf16_t* A, const bf16_t* W, bf16_t* out, int M, int N, int K, int iters) {
  constexpr int TM = 32 * MI;
  const int tilesM = (M + TM - 1) / TM, tilesN = (N + 255) / 256, ntiles = tilesM * tilesN;
  const int grid = std::min(ntiles, 256);
  const int smem = NB * (TM * 64 + 256 * 64);
  CK(hipFuncSetAttribute((const void*)gemm_w4<MI, NB, MODE, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&]() {
    hipLaunchKernelGGL((gemm_w4<MI, NB, MODE, STAG>), dim3(grid), dim3(256), smem, 0, A, W, out, M, N, K, tilesM, tilesN, ntiles,
                       (unsigned)((size_t)M * K * 2), (unsigned)((size_t)N * K * 2), (unsigned)((size_t)M * N * 2));
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const int rounds = (ntiles + 255) / 256;
  printf("  MI=%2d NB=%d MODE=%2d STAG=%d: %8.1f us  %6.0f ns per K-tile (%d rounds)  %7.1f TFLOP/s\n", MI, NB, MODE, (int)STAG, us,
         us * 1e3 / rounds / (K / 32), rounds, 2.0 * M * N * K / us / 1e6);
  return us;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 23552, N = argc > 2 ? atoi(argv[2]) : 6144, K = argc > 3 ? atoi(argv[3]) : 768;
  std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f);
  for (auto& v : hW) v = (bf16_t)((rand() % 2001 - 1000) / 20000.0f);
  bf16_t *A, *W, *out;
  CK(hipMalloc(&A, hA.size() * 2));
  CK(hipMalloc(&W, hW.size() * 2));
  CK(hipMalloc(&out, (size_t)M * N * 2));
  CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(out, 0, (size_t)M * N * 2));
  printf("M=%d N=%d K=%d (bf16 out, plain epilogue)\n", M, N, K);
  // correctness of MODE 0 on 4096 random elements
  {
    run<8, 4, 0>(A, W, out, M, N, K, 1);
    const int n = 4096;
    std::vector<int> r(n), c(n);
    for (int i = 0; i < n; ++i) { r[i] = rand() % M; c[i] = rand() % N; }
    r[0] = M - 1; c[0] = N - 1; r[1] = 0; c[1] = 0;
    int *dr, *dc; float* dref;
    CK(hipMalloc(&dr, n * 4)); CK(hipMalloc(&dc, n * 4)); CK(hipMalloc(&dref, n * 4));
    CK(hipMemcpy(dr, r.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, A, W, dref, dr, dc, n, K);
    std::vector<float> ref(n);
    CK(hipMemcpy(ref.data(), dref, n * 4, hipMemcpyDeviceToHost));
    std::vector<bf16_t> ho((size_t)M * N);
    CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
      const double got = (float)ho[(size_t)r[i] * N + c[i]], d = fabs(got - ref[i]);
      worst = std::max(worst, d);
      if (d > 0.02 + 0.01 * fabs(ref[i])) ++bad;
    }
    printf("  check (MI=8): max |err| %.4f over %d samples, %d beyond the bf16 bound\n", worst, n, bad);
    CK(hipMemset(out, 0, (size_t)M * N * 2));
    run<8, 4, 0, false>(A, W, out, M, N, K, 1);
    CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
    worst = 0; bad = 0;
    for (int i = 0; i < n; ++i) {
      const double got = (float)ho[(size_t)r[i] * N + c[i]], d = fabs(got - ref[i]);
      worst = std::max(worst, d);
      if (d > 0.02 + 0.01 * fabs(ref[i])) ++bad;
    }
    printf("  check (MI=8, no stagger): max |err| %.4f over %d samples, %d beyond the bf16 bound\n", worst, n, bad);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run<8, 4, 0>(A, W, out, M, N, K, 20);
    run<8, 4, 0, false>(A, W, out, M, N, K, 20);
    run<8, 5, 0>(A, W, out, M, N, K, 20);
    run<8, 4, 1>(A, W, out, M, N, K, 20);
    run<8, 4, 2>(A, W, out, M, N, K, 20);
    run<8, 4, 3>(A, W, out, M, N, K, 20);
    run<8, 4, 4>(A, W, out, M, N, K, 20);
    run<8, 4, 8>(A, W, out, M, N, K, 20);
    run<8, 4, 11>(A, W, out, M, N, K, 20);
  }
  return 0;
}
