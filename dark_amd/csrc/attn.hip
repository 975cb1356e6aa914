// Fused bidirectional attention for short sequences (N <= 128): RoPE + QK^T + pad/mask + softmax + PV.
//
// Replaces apply_rotary_pos_emb (modeling_dream_length.py:183-207, :399) and
// F.scaled_dot_product_attention(..., attn_mask=bool[B,1,N,N], is_causal=False) (:424-431).
//
// One workgroup per (sequence, head).  Q, K (rotated) and V of the head are staged once in LDS;
// each wave then owns 16-query tiles.  Scores are computed TRANSPOSED (S^T = K Q^T) with 16x16 MFMA
// so that a lane ends up holding, for ITS query (lane & 15), the keys 4*(lane>>4)+r of every key
// tile: exactly the A-operand layout of the following P V MFMA.  P never touches LDS; the softmax
// row reductions are two xor-shuffles (lanes l, l^16, l^32, l^48 share a query).  The pad/mask
// vector is applied in-register as a per-key additive bias (0 / -inf).
//   bf16: v_mfma_f32_16x16x32_bf16 (P rounded to bf16 for P V, fp32 accumulate)
//   fp32: v_mfma_f32_16x16x4_f32   (exact fp32 fma chains)
#include "common.h"

namespace dark {

template <typename T> struct AttnVec;
template <> struct AttnVec<float> {
  static constexpr int VEC = 4;
  typedef float4 raw_t;
  __device__ static __forceinline__ void unpack(const raw_t& r, float (&f)[4]) {
    f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
  }
  __device__ static __forceinline__ raw_t pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct AttnVec<bf16_t> {
  static constexpr int VEC = 8;
  typedef bf16x8 raw_t;
  __device__ static __forceinline__ void unpack(const raw_t& r, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)r[i];
  }
  __device__ static __forceinline__ raw_t pack(const float (&f)[8]) {
    raw_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)f[i];
    return r;
  }
};

template <typename T, int HD, int NT, int NW, bool HOIST = false>
__global__ __launch_bounds__(NW * 64) void attn_kernel(const T* __restrict__ qkv, const float* __restrict__ cosT,
                                                   const float* __restrict__ sinT,
                                                   const int32_t* __restrict__ pos,
                                                   const uint8_t* __restrict__ kmask, T* __restrict__ out,
                                                   int N, int H, int max_pos, float scale, int causal) {
  constexpr int VEC = AttnVec<T>::VEC;
  typedef typename AttnVec<T>::raw_t raw_t;
  constexpr int NP = NT * 16;     // padded token count
  constexpr int NTH = NW * 64;    // threads per workgroup
  constexpr int NK2 = (NT + 1) / 2;  // 32-key slots of the P V MFMA
  constexpr int LD = HD + VEC;    // padded LDS row (elements): keeps 16-B alignment, breaks bank strides
  constexpr int HALF = HD / 2;
  constexpr int PC = HALF / VEC;  // 16-B chunk pairs per row for RoPE
  constexpr int VC = HD / VEC;    // 16-B chunks per V row
  constexpr bool kF32 = sizeof(T) == 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Qs = (T*)smem;
  T* Ks = Qs + NP * LD;
  T* Vs = Ks + NP * LD;
  float* kbias = (float*)(Vs + NP * LD);  // [NP]: 0 = attend, -inf = masked / padding key

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int dm = H * HD;
  const int64_t ld = 3 * (int64_t)dm;
  const T* base = qkv + (int64_t)b * N * ld + h * HD;

  if constexpr (!kF32 && HOIST) {
    // ---- bf16: every global load of the staging phase is issued before the first one is used (one round trip to
    // memory per workgroup instead of one per loop iteration: the workgroup's life is mostly this wait).  Loads are
    // UNCONDITIONAL on a clamped token index -- an EXEC-masked load leaves its registers undefined -- and the padding
    // rows are zeroed when they are written.
    constexpr int ITQK = 2 * NP * PC / NTH, ITV = NP * VC / NTH;
    static_assert(ITQK * NTH == 2 * NP * PC && ITV * NTH == NP * VC, "staging items must divide over the threads");
    raw_t r1[ITQK], r2[ITQK], rv[ITV];
    float4 cs[ITQK][2], sn[ITQK][2];
#pragma unroll
    for (int u = 0; u < ITQK; ++u) {
      const int it = tid + u * NTH;
      const int which = it / (NP * PC);
      const int rem = it - which * (NP * PC);
      const int t = rem / PC, c = rem - t * PC;
      const int tc = t < N ? t : N - 1;
      int p = pos ? pos[b * N + tc] : tc;
      p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
      const T* src = base + (int64_t)tc * ld + which * dm + c * VEC;
      r1[u] = *(const raw_t*)src;
      r2[u] = *(const raw_t*)(src + HALF);
      const float* cp = cosT + (int64_t)p * HALF + c * VEC;
      const float* sp = sinT + (int64_t)p * HALF + c * VEC;
      cs[u][0] = *(const float4*)cp; cs[u][1] = *(const float4*)(cp + 4);
      sn[u][0] = *(const float4*)sp; sn[u][1] = *(const float4*)(sp + 4);
    }
#pragma unroll
    for (int u = 0; u < ITV; ++u) {
      const int it = tid + u * NTH;
      const int t = it / VC, c = it - t * VC;
      const int tc = t < N ? t : N - 1;
      rv[u] = *(const raw_t*)(base + (int64_t)tc * ld + 2 * dm + c * VEC);
    }
#pragma unroll
    for (int u = 0; u < ITQK; ++u) {
      const int it = tid + u * NTH;
      const int which = it / (NP * PC);
      const int rem = it - which * (NP * PC);
      const int t = rem / PC, c = rem - t * PC;
      float x1[VEC], x2[VEC], o1[VEC], o2[VEC];
      AttnVec<T>::unpack(r1[u], x1);
      AttnVec<T>::unpack(r2[u], x2);
      const float cc[8] = {cs[u][0].x, cs[u][0].y, cs[u][0].z, cs[u][0].w, cs[u][1].x, cs[u][1].y, cs[u][1].z, cs[u][1].w};
      const float ss[8] = {sn[u][0].x, sn[u][0].y, sn[u][0].z, sn[u][0].w, sn[u][1].x, sn[u][1].y, sn[u][1].z, sn[u][1].w};
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        o1[e] = t < N ? x1[e] * cc[e] - x2[e] * ss[e] : 0.f;
        o2[e] = t < N ? x2[e] * cc[e] + x1[e] * ss[e] : 0.f;
      }
      T* dst = (which ? Ks : Qs) + t * LD + c * VEC;
      *(raw_t*)dst = AttnVec<T>::pack(o1);
      *(raw_t*)(dst + HALF) = AttnVec<T>::pack(o2);
    }
#pragma unroll
    for (int u = 0; u < ITV; ++u) {
      const int it = tid + u * NTH;
      const int t = it / VC, c = it - t * VC;
      raw_t v = rv[u];
      if (t >= N) {
        float z[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) z[e] = 0.f;
        v = AttnVec<T>::pack(z);
      }
      *(raw_t*)(Vs + t * LD + c * VEC) = v;
    }
  } else {
  // ---- stage Q, K with RoPE
  for (int it = tid; it < 2 * NP * PC; it += NTH) {
    const int which = it / (NP * PC);
    const int rem = it - which * (NP * PC);
    const int t = rem / PC, c = rem - t * PC;
    float o1[VEC], o2[VEC];
    if (t < N) {
      int p = pos ? pos[b * N + t] : t;
      p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
      const T* src = base + (int64_t)t * ld + which * dm + c * VEC;
      float x1[VEC], x2[VEC];
      AttnVec<T>::unpack(*(const raw_t*)src, x1);
      AttnVec<T>::unpack(*(const raw_t*)(src + HALF), x2);
      const float* cp = cosT + (int64_t)p * HALF + c * VEC;
      const float* sp = sinT + (int64_t)p * HALF + c * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float cs = cp[e], sn = sp[e];
        o1[e] = x1[e] * cs - x2[e] * sn;
        o2[e] = x2[e] * cs + x1[e] * sn;
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o1[e] = o2[e] = 0.f;
    }
    T* dst = (which ? Ks : Qs) + t * LD + c * VEC;
    *(raw_t*)dst = AttnVec<T>::pack(o1);
    *(raw_t*)(dst + HALF) = AttnVec<T>::pack(o2);
  }
  // ---- stage V
  for (int it = tid; it < NP * VC; it += NTH) {
    const int t = it / VC, c = it - t * VC;
    raw_t v;
    if (t < N) {
      v = *(const raw_t*)(base + (int64_t)t * ld + 2 * dm + c * VEC);
    } else {
      float z[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) z[e] = 0.f;
      v = AttnVec<T>::pack(z);
    }
    *(raw_t*)(Vs + t * LD + c * VEC) = v;
  }
  }
  for (int j = tid; j < NP; j += NTH) {
    const bool on = j < N && (!kmask || kmask[b * N + j] != 0);
    kbias[j] = on ? 0.f : -INFINITY;
  }
  __syncthreads();

  const int li = lane & 15, g = lane >> 4;
  const int n_t = (N + 15) >> 4;  // live 16-token tiles (queries and keys)
  for (int it = wave; it < n_t; it += NW) {
    // ---- S^T tiles: rows = keys, cols = queries
    f32x4 S[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      S[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (jt < n_t) {
        const T* kr = Ks + (jt * 16 + li) * LD;
        const T* qr = Qs + (it * 16 + li) * LD;
        if constexpr (kF32) {
#pragma unroll
          for (int ks = 0; ks < HD / 4; ++ks)
            S[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[ks * 4 + g], qr[ks * 4 + g], S[jt], 0, 0, 0);
        } else {
#pragma unroll
          for (int ks = 0; ks < HD / 32; ++ks)
            S[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kr + ks * 32 + g * 8),
                                                            *(const bf16x8*)(qr + ks * 32 + g * 8), S[jt], 0, 0, 0);
        }
      }
    }
    // ---- softmax over keys for query qi = it*16 + li (keys jt*16 + 4g + r live in this lane)
    const int qi = it * 16 + li;
    const bool q_on = qi < N && (!kmask || kmask[b * N + qi] != 0);
    float mx = -INFINITY;
    if constexpr (!kF32) {
      // bf16: P is rounded to bf16 before P V, so the softmax runs on the hardware exp2 (scale * log2(e) folded into
      // one fma with the key bias) and ONE reciprocal per query instead of a libm expf and an IEEE division per
      // score -- at N = 128 the kernel was bound by exactly that arithmetic (~35 vector instructions a score)
      const float c2 = scale * 1.44269504088896340736f;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = __builtin_fmaf(S[jt][r], c2, kbias[jt * 16 + 4 * g + r]);
          if (causal && jt * 16 + 4 * g + r > qi) s = -INFINITY;
          S[jt][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mz = (mx == -INFINITY) ? 0.f : mx;   // a fully masked row: exp2(-inf - 0) = 0 everywhere
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(S[jt][r] - mz);
          S[jt][r] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = (q_on && sum > 0.f) ? 1.0f / sum : 0.f;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[jt][r] *= inv;
    } else {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = S[jt][r] * scale + kbias[jt * 16 + 4 * g + r];
        if (causal && jt * 16 + 4 * g + r > qi) s = -INFINITY;  // (GPT-2 backbone only; wave-uniform flag)
        S[jt][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (mx == -INFINITY) ? 0.f : expf(S[jt][r] - mx);
        S[jt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const bool live = q_on && sum > 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) S[jt][r] = live ? S[jt][r] / sum : 0.f;
    }

    // ---- O = P V
    T* orow = out + ((int64_t)b * N) * dm + h * HD;
    if constexpr (kF32) {
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          if (jt < n_t) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(S[jt][r], Vs[(jt * 16 + 4 * g + r) * LD + dt * 16 + li],
                                                         acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = it * 16 + 4 * g + r;
          if (i < N) orow[(int64_t)i * dm + dt * 16 + li] = acc[r];
        }
      }
    } else {
      bf16x8 pa[NK2];  // an odd tile count leaves the upper half of the last 32-key slot at zero
#pragma unroll
      for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int jt = 2 * k2 + (jj >> 2);
          pa[k2][jj] = jt < NT ? (bf16_t)S[jt < NT ? jt : 0][jj & 3] : (bf16_t)0.f;
        }
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2) {
          if (2 * k2 < n_t) {
            // B operand = V[key][dim] for this lane's dim (dt*16 + li) and the 8 keys of its P slots: two transposed
            // LDS reads (ds_read_b64_tr_b16: each 16-lane group fetches a 4-key x 16-dim block row-wise and receives
            // it column-wise; lane 4q+p of the group supplies the address of key q, dims 4p..4p+3) instead of eight
            // 2-byte reads.  The wave is fully active here (wave-uniform loops and conditions only), as the
            // instruction requires; an odd tile count leaves the upper half of the last slot at zero.
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            const T* vrow = Vs + (2 * k2 * 16 + 4 * g + (li >> 2)) * LD + dt * 16 + 4 * (li & 3);
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)vrow);
            bf16x4 hi = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            if (2 * k2 + 1 < NT) hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vrow + 16 * LD));
            const bf16x8 vb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[k2], vb, acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = it * 16 + 4 * g + r;
          if (i < N) orow[(int64_t)i * dm + dt * 16 + li] = (bf16_t)acc[r];
        }
      }
    }
  }
}

template <typename T, int HD, int NT, int NW, bool HOIST = false>
static int launch_attn_t(const void* qkv, const float* cosT, const float* sinT, const int32_t* pos,
                         const uint8_t* kmask, void* out, int B, int N, int H, int max_pos, hipStream_t st, int causal) {
  constexpr int VEC = AttnVec<T>::VEC;
  constexpr int smem = 3 * NT * 16 * (HD + VEC) * (int)sizeof(T) + NT * 16 * 4;
  static SmemOptIn optin;
  DARK_TRY(optin.ensure((const void*)attn_kernel<T, HD, NT, NW, HOIST>, smem, st));
  const float scale = 1.0f / sqrtf((float)HD);
  hipLaunchKernelGGL((attn_kernel<T, HD, NT, NW, HOIST>), dim3(B * H), dim3(NW * 64), smem, st, (const T*)qkv, cosT, sinT, pos,
                     kmask, (T*)out, N, H, max_pos, scale, causal);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// A/B hook (dark_dbg_set_gemm_variant(12)): 1 = the N <= 128 instance without the hoisted staging loads
static std::atomic<int> g_attn_variant{0};
void set_attn_variant(int v) { g_attn_variant.store(v); }
static int attn_variant() { return g_attn_variant.load(std::memory_order_relaxed); }

int launch_attention(int dtype, const void* qkv, const float* cosT, const float* sinT, const int32_t* pos,
                     const uint8_t* kmask, void* out, int B, int N, int H, int HD, int max_pos,
                     hipStream_t st, bool causal) {
  DARK_CHECK_ARG(N >= 1 && N <= 128, "attention: N=%d must be in [1,128]", N);
  DARK_CHECK_ARG(HD == 64 || HD == 96, "attention: head_dim=%d must be 64 or 96", HD);
  DARK_CHECK_ARG(max_pos >= 1, "attention: empty RoPE table");
  // 16-token tiles: 3 (N <= 48, the default max_length 46: one wave per query tile, 30 KB of LDS -> 5 workgroups
  // per CU; 38.9 vs 42.0 us at B = 512 against the 4-tile / 4-wave instance), 4 (N <= 64) or 8 (N <= 128: 55 KB of
  // LDS -> 2 workgroups per CU, so one wave per query tile -- 8 waves -- to have 16 instead of 8 waves per CU hiding
  // the staging latency: BASELINE config 4, 906 -> 923 sequences/s against two query tiles per wave)
#define DARK_ATTN(T, HDv)                                                                                        \
  if (N <= 48) return launch_attn_t<T, HDv, 3, 3>(qkv, cosT, sinT, pos, kmask, out, B, N, H, max_pos, st, causal ? 1 : 0);        \
  if (N <= 64) return launch_attn_t<T, HDv, 4, 4>(qkv, cosT, sinT, pos, kmask, out, B, N, H, max_pos, st, causal ? 1 : 0);        \
  if (attn_variant() == 1) return launch_attn_t<T, HDv, 8, 8, false>(qkv, cosT, sinT, pos, kmask, out, B, N, H, max_pos, st, causal ? 1 : 0); \
  return launch_attn_t<T, HDv, 8, 8, true>(qkv, cosT, sinT, pos, kmask, out, B, N, H, max_pos, st, causal ? 1 : 0)
  if (dtype == DARK_BF16) {
    if (HD == 96) { DARK_ATTN(bf16_t, 96); } else { DARK_ATTN(bf16_t, 64); }
  } else {
    if (HD == 96) { DARK_ATTN(float, 96); } else { DARK_ATTN(float, 64); }
  }
#undef DARK_ATTN
}

}  // namespace dark
