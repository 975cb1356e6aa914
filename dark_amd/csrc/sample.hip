// Fused temperature / top-p / top-k / softmax / categorical draw / confidence over one logits row.
//
// Replaces sample_tokens (generation_utils.py:58-90), top_p_logits (:37-48) and top_k_logits
// (:50-55): a full descending sort, a cumsum, a scatter and ~8 elementwise passes over [M,V] in the
// reference become one workgroup per masked row with the row resident in LDS (HBM traffic: the row
// and its noise row once each).
//
// Nucleus filter without a sort: the kept set of top_p_logits is the prefix of the descending order
// up to and including the first element whose inclusive cumulative probability exceeds top_p.  All
// cumulative sums are integers (probability MASS in 2^-40 fixed point), so they are order independent
// and deterministic.  Peaked rows take a head path: the elements within 6 of the row maximum are gathered in
// the pass that sums the masses, and if they hold more than top_p of the mass the crossing is resolved among
// them directly.  Otherwise the crossing element is found in two stages:
//   1. one histogram pass over bins that are LINEAR in (row max - logit), 28/1023 wide (mass is 0 beyond 27.7),
//      so the row spreads over the bins and the LDS atomics do not pile up on a few addresses; the bin
//      map is monotone in the logit, hence bins are contiguous ranges of the descending order;
//   2. the handful of elements of the crossing bin are gathered and resolved exactly (O(c^2/512) per thread).
// If the crossing bin holds more than CAND_CAP elements (e.g. a flat row), an exact 4 x 8-bit radix select
// over the order-preserving integer image of the logits takes over; top_k always uses that select with
// unit weights.  Inside a run of equal logits the kept ones are the first by index (the reference's unstable
// torch.sort leaves that order unspecified; oracle.top_p_logits(stable_ties=True) is the same rule).
//
// Draw: x0 = argmax_i (p_i / psum) / q_i with q ~ Exp(1) supplied by the caller, which is the ATen
// multinomial(1) that Categorical(probs).sample() runs; ties -> lowest index (torch argmax).
#include "common.h"
#include <rocrand/rocrand_kernel.h>

namespace dark {

constexpr int ST = 768;         // threads per row: 12 waves (2 workgroups per CU by LDS = 24 waves, 78 VGPRs allow 24; with 8
                                // waves the VALU-bound mass pass had 4 waves per SIMD to hide its LDS reads: +0.3 % in the loop)
constexpr int NLIN = 1024;      // linear bins of stage 1
constexpr int NRAD = 256;       // radix bins of the fallback select
constexpr int CAND_CAP = 512;        // candidates resolved exactly, one per thread (CAND_CAP <= ST)
constexpr float HEAD_RANGE = 6.0f;    // head path: candidates are the logits within this of the row maximum
constexpr float LIN_RANGE = 28.0f;
#define DARK_FLT_LOWEST (-3.4028234663852886e38f)  // torch.finfo(torch.float32).min

__device__ __forceinline__ uint32_t fkey(float f) {  // order-preserving float -> uint
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
// exp(x) for x <= 0 on the hardware exp2 (v_exp_f32, 1 ulp): the argument x*log2(e) is carried as a rounded
// product plus its exact residual (fma) and the low word of log2(e), so the result is good to ~2 ulp over the
// whole range instead of the |x| * 6e-8 relative error of the plain exp2f(x * log2e).  Results below 2^-126
// flush to zero (they are < 1e-37 relative to the row maximum).
__device__ __forceinline__ float exp_neg(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
  const float t = x * L2E_HI;
  float r = fmaf(x, L2E_HI, -t);
  r = fmaf(x, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.693147180559945309f, e);
}
// ln(x) for 1e-10 <= x <= ~1 on v_log_f32 (1 ulp in log2)
__device__ __forceinline__ float log_pos(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }

constexpr float MASS_RANGE = 27.8f;  // exp(-27.8) * 2^40 < 1: fixed-point mass is 0 beyond
__device__ __forceinline__ unsigned long long mass_fx(float l, float mx) {
  const float d = l - mx;
  if (!(d > -MASS_RANGE)) return 0ull;
  return (unsigned long long)(exp_neg(d) * 1099511627776.0f);  // 2^40
}
// the same value without a branch around the exp (d = NaN or -inf: the product is NaN or 0, the select gives 0)
__device__ __forceinline__ unsigned long long mass_fx_nb(float l, float mx) {
  const float d = l - mx;
  const unsigned long long m = (unsigned long long)(exp_neg(fmaxf(d, -MASS_RANGE)) * 1099511627776.0f);
  return d > -MASS_RANGE ? m : 0ull;
}
constexpr float LIN_SCALE = (float)(NLIN - 1) / LIN_RANGE;
// (any positive scale gives a monotone map, and the search is exact whatever the bins are: a row known to be flat uses
// bins over its own range, so that the crossing bin holds a few dozen elements instead of a few hundred)
__device__ __forceinline__ int lin_bin(float l, float mx, float sc = LIN_SCALE) {
  const float d = (mx - l) * sc;                      // monotone non-increasing in l
  return d < (float)(NLIN - 1) ? (int)d : NLIN - 1;   // NaN / inf / huge -> last bin
}

struct Scratch {
  unsigned long long u64[ST / 64], u64b[ST / 64];   // one slot per wave
  double f64a[ST / 64], f64b[ST / 64];
  float f32a[ST / 64], f32b[ST / 64], f32c[ST / 64];
  int i32[ST / 64];
  uint32_t u32[ST / 64];
  // select broadcast
  int found, bin, ncand;
  unsigned long long above;
};

template <int NT = ST>
__device__ __forceinline__ float block_max(float v, Scratch* s, int tid) {
  v = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) s->f32a[tid >> 6] = v;
  __syncthreads();
  float r = s->f32a[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = fmaxf(r, s->f32a[w]);
  return r;
}
template <int NT = ST>
__device__ __forceinline__ void block_max2(float& a, float& b, Scratch* s, int tid) {
  a = wave_max(a);
  b = wave_max(b);
  __syncthreads();
  if ((tid & 63) == 0) { s->f32a[tid >> 6] = a; s->f32b[tid >> 6] = b; }
  __syncthreads();
  a = s->f32a[0];
  b = s->f32b[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) { a = fmaxf(a, s->f32a[w]); b = fmaxf(b, s->f32b[w]); }
}
template <int NT = ST>
__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, Scratch* s, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) s->u32[tid >> 6] = v;
  __syncthreads();
  uint32_t r = s->u32[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = max(r, s->u32[w]);
  return r;
}
template <int NT = ST>
__device__ __forceinline__ double block_sum(double v, Scratch* s, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) s->f64a[tid >> 6] = v;
  __syncthreads();
  double r = s->f64a[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r += s->f64a[w];
  return r;
}
template <int NT = ST>
__device__ __forceinline__ void block_sum2(double& a, double& b, Scratch* s, int tid) {
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();
  if ((tid & 63) == 0) { s->f64a[tid >> 6] = a; s->f64b[tid >> 6] = b; }
  __syncthreads();
  a = s->f64a[0];
  b = s->f64b[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) { a += s->f64a[w]; b += s->f64b[w]; }
}
template <int NT = ST>
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, Scratch* s, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) s->u64[tid >> 6] = v;
  __syncthreads();
  unsigned long long r = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) r += s->u64[w];
  return r;
}

template <int NT = ST>
__device__ __forceinline__ void block_sum2_u64(unsigned long long& a, unsigned long long& b, Scratch* s, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  __syncthreads();
  if ((tid & 63) == 0) { s->u64[tid >> 6] = a; s->u64b[tid >> 6] = b; }
  __syncthreads();
  a = 0; b = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { a += s->u64[w]; b += s->u64b[w]; }
}

// Wave 0 scans `nbin` weight bins in the order first = bin `start`, step `dir` (the descending-logit order)
// for the first non-empty bin whose inclusive cumulative weight, starting from A, exceeds T.
// Publishes s->found / s->bin / s->above (weight strictly before that bin).  nbin must be a multiple of 64.
__device__ __forceinline__ void scan_crossing(const unsigned long long* hist, int nbin, int start, int dir,
                                              unsigned long long A, unsigned long long T, Scratch* s, int tid) {
  if (tid < 64) {
    const int per = nbin >> 6;
    const int first = start + dir * per * tid;  // lane L owns `per` consecutive bins of the scan order
    unsigned long long tot = 0;
    for (int q = 0; q < per; ++q) tot += hist[first + dir * q];
    unsigned long long inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long t = __shfl_up(inc, o, 64);
      if (tid >= o) inc += t;
    }
    unsigned long long run = A + inc - tot;  // weight strictly before this lane's bins
    int hit = -1;
    unsigned long long hit_above = 0;
    for (int q = 0; q < per; ++q) {
      const unsigned long long hq = hist[first + dir * q];
      if (hit < 0 && hq != 0 && run + hq > T) { hit = first + dir * q; hit_above = run; }
      run += hq;
    }
    const unsigned long long bal = __ballot(hit >= 0);
    if (bal == 0) {
      if (tid == 0) s->found = 0;
    } else {
      const int firstl = __ffsll((long long)bal) - 1;
      if (tid == firstl) { s->found = 1; s->bin = hit; s->above = hit_above; }
    }
  }
}

// Exact radix select (4 levels of 8 bits) over the integer image of the logits: finds the first element, in
// descending key order, whose inclusive cumulative weight exceeds T.  weight = 2^-40 fixed-point softmax
// mass (MASS) or 1 (count).  On success returns true with *tau = its key, *above = total weight of keys
// strictly greater, *n_tau = multiplicity of that key.
template <bool MASS>
__device__ bool select_crossing(const float* row, int V, float mx, unsigned long long T,
                                unsigned long long* hist, unsigned int* cnth, Scratch* s, int tid,
                                uint32_t* tau, unsigned long long* above, int* n_tau) {
  uint32_t prefix = 0;
  int pbits = 0;
  unsigned long long A = 0;
  for (int level = 0; level < 4; ++level) {
    const int shift = 24 - 8 * level;
    for (int i = tid; i < NRAD; i += ST) { hist[i] = 0; cnth[i] = 0; }
    __syncthreads();
    for (int i = tid; i < V; i += ST) {
      const float l = row[i];
      const uint32_t k = fkey(l);
      if (pbits == 0 || (k >> (32 - pbits)) == prefix) {
        const int dg = (k >> shift) & 0xffu;
        atomicAdd(&hist[dg], MASS ? mass_fx(l, mx) : 1ull);
        atomicAdd(&cnth[dg], 1u);
      }
    }
    __syncthreads();
    scan_crossing(hist, NRAD, NRAD - 1, -1, A, T, s, tid);
    __syncthreads();
    if (!s->found) return false;
    const int b = s->bin;
    A = s->above;
    prefix = (prefix << 8) | (uint32_t)b;
    pbits += 8;
    if (level == 3) *n_tau = (int)cnth[b];
    __syncthreads();  // hist/cnth are re-zeroed at the top of the next level
  }
  *tau = prefix;
  *above = A;
  return true;
}

// Exact resolution among c gathered candidates (keys cand[], fixed-point masses cfx[]) that form a contiguous
// range of the descending order with mass A strictly above it: tau = the largest key whose inclusive
// cumulative mass exceeds T.  Returns 1 and *tau / *above (mass of keys > tau) / *n_tau (multiplicity), or 0.
template <int NT = ST>
__device__ int resolve_candidates(const uint32_t* cand, const unsigned long long* cfx, int c, unsigned long long A,
                                  unsigned long long T, Scratch* s, int tid, uint32_t* tau,
                                  unsigned long long* above, int* n_tau) {
  // c <= CAND_CAP = ST: thread j owns candidate j and sums the masses of all candidates not below it
  const bool own = tid < c;
  const uint32_t kj = own ? cand[tid] : 0u;
  const unsigned long long fj = own ? cfx[tid] : 0ull;
  unsigned long long S = A;
  if (tid < ((c + 63) & ~63)) {  // waves that own no candidate skip the O(c) loop
    const int c4 = c & ~3;
    for (int i = 0; i < c4; i += 4) {  // broadcast LDS reads, independent of each other: issued back to back
      const uint32_t k0 = cand[i], k1 = cand[i + 1], k2 = cand[i + 2], k3 = cand[i + 3];
      const unsigned long long f0 = cfx[i], f1 = cfx[i + 1], f2 = cfx[i + 2], f3 = cfx[i + 3];
      S += (k0 >= kj ? f0 : 0ull) + (k1 >= kj ? f1 : 0ull) + (k2 >= kj ? f2 : 0ull) + (k3 >= kj ? f3 : 0ull);
    }
    for (int i = c4; i < c; ++i) S += cand[i] >= kj ? cfx[i] : 0ull;
  }
  const uint32_t best = block_max_u32<NT>((own && S > T) ? kj : 0u, s, tid);
  if (best == 0) return 0;
  // mass strictly above tau and the multiplicity of tau, again one candidate per thread
  unsigned long long ab = (own && kj > best) ? fj : 0ull, nt = (own && kj == best) ? 1ull : 0ull;
  block_sum2_u64<NT>(ab, nt, s, tid);
  *tau = best;
  *above = A + ab;
  *n_tau = (int)nt;
  return 1;
}

// The nucleus search (see the header).  Returns 1 = found, 0 = no crossing (nothing to filter), -1 = the
// crossing bin overflows CAND_CAP (caller falls back to the radix select), 2 = (only if defer_small) the head path
// applies with at most 64 candidates and nothing has been resolved yet: the caller hands the row to list_tail_wave.
// *head_c: number of gathered head candidates when the head path resolved the crossing (every kept element is
// then one of cand[0..c) / cidx[0..c)), -1 otherwise.
__device__ int nucleus_crossing(const float* row, int V, float mx, float top_p, unsigned long long* hist,
                                uint32_t* cand, unsigned long long* cfx, int* cidx, Scratch* s, int tid,
                                uint32_t* tau, unsigned long long* above, int* n_tau, unsigned long long* Tout,
                                int* head_c, bool defer_small, bool flat_hint, float mn) {
  *head_c = -1;
  unsigned long long T;
  float sc = LIN_SCALE;
  if (flat_hint) {
    const float span = fminf(LIN_RANGE, mx - mn);  // mn: the row minimum (only read here)
    if (span > 0.f) sc = (float)(NLIN - 1) / span;
    // The first stage already found this row flat (more than SH pieces within reach of the maximum): the head path
    // cannot resolve it, so the pass that sums the masses fills the histogram as well -- the same integers summed
    // in another order, one exp pass and the candidate atomics less.
    for (int i = tid; i < NLIN; i += ST) hist[i] = 0;
    if (tid == 0) s->ncand = 0;
    __syncthreads();
    unsigned long long zl = 0;
    const int v4 = V >> 2;
    for (int j0 = tid; j0 < v4; j0 += 2 * ST) {
      float4 l4[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (j0 + u * ST < v4) l4[u] = *(const float4*)(row + 4 * (j0 + u * ST));
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (j0 + u * ST < v4) {
          const float le[4] = {l4[u].x, l4[u].y, l4[u].z, l4[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned long long fx = mass_fx_nb(le[e], mx);
            zl += fx;
            if (fx) atomicAdd(&hist[lin_bin(le[e], mx, sc)], fx);
          }
        }
      }
    }
    for (int i = 4 * v4 + tid; i < V; i += ST) {
      const float l = row[i];
      const unsigned long long fx = mass_fx(l, mx);
      zl += fx;
      if (fx) atomicAdd(&hist[lin_bin(l, mx, sc)], fx);
    }
    zl = block_sum_u64(zl, s, tid);  // (its barriers also publish the histogram)
    T = (unsigned long long)((double)top_p * (double)zl);  // floor
    *Tout = T;
  } else {
  // ---- head path: the pass that sums the masses also gathers the elements within HEAD_RANGE of the
  // maximum; when they hold more than top_p of the mass (any peaked row) the crossing is among them
  if (tid == 0) s->ncand = 0;
  __syncthreads();
  unsigned long long zl = 0, hl = 0;  // row mass and the mass of the elements this thread gathered
  int mine = 0;  // a thread gathers at most 2 of its ~V/512 elements: a flat row overflows after <= 1024 atomics
  auto gather = [&](float l, unsigned long long fx, int i) {
    if (mine < 2) {
      const int slot = atomicAdd(&s->ncand, 1);
      if (slot < CAND_CAP) { cand[slot] = fkey(l); cfx[slot] = fx; cidx[slot] = i; hl += fx; }
    } else if (mine == 2) {
      atomicAdd(&s->ncand, CAND_CAP + 1);  // poison the count: the head set is incomplete
    }
    ++mine;
  };
  // 16-byte LDS reads, two in flight, masses computed without branches (the sums are integers: which thread adds
  // which element does not matter); as a loop over single elements every iteration waited out an LDS round trip
  // and two divergent branches, and this pass was a third of the workgroup's lifetime
  const int v4 = V >> 2;
  for (int j0 = tid; j0 < v4; j0 += 2 * ST) {
    float4 l4[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (j0 + u * ST < v4) l4[u] = *(const float4*)(row + 4 * (j0 + u * ST));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = j0 + u * ST;
      if (j < v4) {
        const float le[4] = {l4[u].x, l4[u].y, l4[u].z, l4[u].w};
        unsigned long long fx[4];
        bool any_head = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          fx[e] = mass_fx_nb(le[e], mx);
          zl += fx[e];
          any_head |= le[e] - mx > -HEAD_RANGE;
        }
        if (any_head) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (le[e] - mx > -HEAD_RANGE) gather(le[e], fx[e], 4 * j + e);
        }
      }
    }
  }
  for (int i = 4 * v4 + tid; i < V; i += ST) {
    const float l = row[i];
    const unsigned long long fx = mass_fx(l, mx);
    zl += fx;
    if (l - mx > -HEAD_RANGE) gather(l, fx, i);
  }
  block_sum2_u64(zl, hl, s, tid);  // (its barriers also publish the candidates)
  const unsigned long long Z = zl;
  T = (unsigned long long)((double)top_p * (double)Z);  // floor
  *Tout = T;
  const int ch = s->ncand;
  if (ch <= CAND_CAP && hl > T) {  // ch <= CAND_CAP: every gathered element has a slot, hl is their total mass
    *head_c = ch;
    if (defer_small && ch <= 64) return 2;
    return resolve_candidates(cand, cfx, ch, 0ull, T, s, tid, tau, above, n_tau);
  }
  // ---- general path: mass histogram over linear bins, then the elements of the crossing bin
  for (int i = tid; i < NLIN; i += ST) hist[i] = 0;
  __syncthreads();
  if (tid == 0) s->ncand = 0;
  for (int i = tid; i < V; i += ST) {
    const float l = row[i];
    const unsigned long long fx = mass_fx(l, mx);
    if (fx) atomicAdd(&hist[lin_bin(l, mx)], fx);
  }
  __syncthreads();
  }  // !flat_hint
  scan_crossing(hist, NLIN, 0, +1, 0ull, T, s, tid);
  __syncthreads();
  if (!s->found) return 0;
  const int b = s->bin;
  const unsigned long long A = s->above;
  auto take = [&](float l) {
    if (lin_bin(l, mx, sc) == b) {
      const int slot = atomicAdd(&s->ncand, 1);
      if (slot < CAND_CAP) { cand[slot] = fkey(l); cfx[slot] = mass_fx(l, mx); }
    }
  };
  {  // 16-byte reads, two in flight (which thread records which element does not matter: resolve_candidates sums)
    const int v4 = V >> 2;
    for (int j0 = tid; j0 < v4; j0 += 2 * ST) {
      float4 l4[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (j0 + u * ST < v4) l4[u] = *(const float4*)(row + 4 * (j0 + u * ST));
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (j0 + u * ST < v4) { take(l4[u].x); take(l4[u].y); take(l4[u].z); take(l4[u].w); }
    }
    for (int i = 4 * v4 + tid; i < V; i += ST) take(row[i]);
  }
  __syncthreads();
  const int c = s->ncand;
  if (c > CAND_CAP) return -1;
  return resolve_candidates(cand, cfx, c, A, T, s, tid, tau, above, n_tau);
}

// ------------------------------------------------------------------ lazy categorical noise
// Element `li` of what `torch.empty(numel).exponential_()` writes on this device for a generator at (seed, offset):
// ATen's distribution_elementwise_grid_stride_kernel runs `nthreads` threads, thread t = li % nthreads draws its
// n-th rocrand_uniform4 (Philox4x32-10, subsequence t, offset + 4n) for the elements li = t + nthreads*(4n + ii) and
// applies transformation::exponential (ATen/core/TransformationHelper.h).  Computing it where it is consumed
// replaces an [M, V] fp32 fill (up to 1 GB per step) by a few Philox calls per row; the values are identical.
struct PhiloxNoise {
  unsigned long long seed, offset;
  unsigned int nthreads;  // 0: noise comes from memory
};
// element li = q * nthreads + t, t < nthreads
__device__ __forceinline__ float philox_exponential_qt(const PhiloxNoise& pn, unsigned long long q, unsigned int t) {
  rocrand_state_philox4x32_10 st;
  rocrand_init(pn.seed, t, pn.offset + 4ull * (q >> 2), &st);
  const uint4 raw = rocrand4(&st);
  const unsigned int ii = (unsigned int)(q & 3);
  const unsigned int rv = ii == 0 ? raw.x : (ii == 1 ? raw.y : (ii == 2 ? raw.z : raw.w));
  const float v = fmaf((float)rv, ROCRAND_2POW32_INV, ROCRAND_2POW32_INV);  // rocrand uniform_distribution: (0, 1]
  // transformation::exponential with lambda = 1.  The log torch's wheel was built with is the hardware log2 times
  // ln 2 with a single rounding (found by bit-comparison on 85 M values; pinned by
  // tests/test_hip_kernels.py::test_philox_exponential_is_torchs_stream; the library logf of
  // this ROCm release differs from it in the last bit for a third of the arguments).
  const float eps_half = 1.1920928955078125e-07f / 2;
  const float lgv = (float)((double)__builtin_amdgcn_logf(v) * 0.6931471805599453);
  const float lg = v >= 1.0f - eps_half ? -eps_half : lgv;
  return -1.0f / 1.0f * lg;
}
__device__ __forceinline__ float philox_exponential(const PhiloxNoise& pn, unsigned long long li) {
  const unsigned long long q = li / pn.nthreads;
  return philox_exponential_qt(pn, q, (unsigned int)(li - q * pn.nthreads));
}

__global__ __launch_bounds__(256) void philox_exponential_kernel(PhiloxNoise pn, unsigned long long numel,
                                                                 float* __restrict__ out) {
  const unsigned long long li = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (li < numel) out[li] = philox_exponential(pn, li);
}

// A peaked row whose head set has at most 64 candidates (what T = 0.1 makes of almost every row) is finished by
// wave 0 alone: crossing, kept set, softmax, confidence and draw over one candidate per lane, with wave shuffles in
// place of the block reductions of the general path (each of which costs two barriers and an LDS round trip, ~1 us
// apiece: a third of the block's lifetime).  The arithmetic is that of the general list mode operation for
// operation -- the other seven waves would only ever contribute zeros -- so the results are bit-identical.
__device__ void list_tail_wave(const uint32_t* cand, const unsigned long long* cfx, const int* cidx, int c,
                               unsigned long long T, float mx, int V, float temperature, int conf_mode,
                               const float* qr, const PhiloxNoise& pn, unsigned long long q0, int lane,
                               float* conf_out, int64_t* x0_out) {
  const bool own = lane < c;
  const uint32_t kj = own ? cand[lane] : 0u;
  const unsigned long long fj = own ? cfx[lane] : 0ull;
  const int my_idx = own ? cidx[lane] : 0;
  // mass of the candidates not below mine (resolve_candidates with A = 0)
  unsigned long long S = 0;
  {
    const int c4 = c & ~3;
    for (int i = 0; i < c4; i += 4) {
      const uint32_t k0 = cand[i], k1 = cand[i + 1], k2 = cand[i + 2], k3 = cand[i + 3];
      const unsigned long long f0 = cfx[i], f1 = cfx[i + 1], f2 = cfx[i + 2], f3 = cfx[i + 3];
      S += (k0 >= kj ? f0 : 0ull) + (k1 >= kj ? f1 : 0ull) + (k2 >= kj ? f2 : 0ull) + (k3 >= kj ? f3 : 0ull);
    }
    for (int i = c4; i < c; ++i) S += cand[i] >= kj ? cfx[i] : 0ull;
  }
  uint32_t tau = (own && S > T) ? kj : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tau = max(tau, (uint32_t)__shfl_xor((int)tau, o, 64));
  unsigned long long above = (own && kj > tau) ? fj : 0ull, nt = (own && kj == tau) ? 1ull : 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { above += __shfl_xor(above, o, 64); nt += __shfl_xor(nt, o, 64); }
  const int n_tau = (int)nt;
  float my_e = 0.f;
  // tau != 0: the lowest candidate has S = head mass > T, so some lane qualified
  const unsigned long long fx = mass_fx(fkey_inv(tau), mx);
  const unsigned long long t_keep = fx ? (T - above) / fx + 1ull : (unsigned long long)n_tau;
  if (own) {
    bool keep = kj > tau;
    if (kj == tau) {
      keep = true;
      if (t_keep < (unsigned long long)n_tau) {  // only the first t_keep (by index) of the equal logits stay
        unsigned long long rank = 0;
        for (int i = 0; i < c; ++i) rank += (cand[i] == tau && cidx[i] < my_idx);
        keep = rank < t_keep;
      }
    }
    if (keep) my_e = exp_neg(fkey_inv(kj) - mx);
  }
  const float Zf = (float)wave_sum((double)my_e);
  double ps = 0.0, ent = 0.0;
  float t1 = -INFINITY, t2 = -INFINITY, bestp = -INFINITY, my_p = 0.f;
  int besti = 0x7fffffff;
  if (my_e > 0.f) {
    my_p = my_e / Zf;
    ps = (double)my_p;
    if (conf_mode == DARK_CONF_NEG_ENTROPY) ent = (double)(my_p * log_pos(my_p + 1e-10f));
    t1 = my_p;
    bestp = my_p;
    besti = my_idx;
  }
  const float psum = (float)wave_sum(ps);
  const float entropy = (float)wave_sum(ent);
  if (temperature > 0.f) {
    bestp = -INFINITY;
    besti = 0x7fffffff;
    if (my_p > 0.f) {
      bestp = (my_p / psum) / (pn.nthreads ? philox_exponential(pn, q0 + my_idx) : qr[my_idx]);
      besti = my_idx;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bestp, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > bestp || (ov == bestp && oi < besti)) { bestp = ov; besti = oi; }
    const float o1 = __shfl_xor(t1, o, 64), o2 = __shfl_xor(t2, o, 64);
    const float n1 = fmaxf(t1, o1), n2 = fmaxf(fminf(t1, o1), fmaxf(t2, o2));
    t1 = n1; t2 = n2;
  }
  int bi = besti;
  if (bi < 0 || bi >= V) bi = 0;
  if (t2 < 0.f) t2 = 0.f;
  if (lane == 0) {
    *conf_out = conf_mode == DARK_CONF_NEG_ENTROPY ? entropy : (conf_mode == DARK_CONF_MARGIN ? t1 - t2 : 0.f);
    *x0_out = (int64_t)bi;
  }
  if (conf_mode == DARK_CONF_PROB && my_e > 0.f && my_idx == bi) *conf_out = my_p;  // after lane 0's store
}

// ---- two-stage sampler, first stage: the head path with the row in REGISTERS.
// sample_kernel keeps a row in LDS (60 KB at V = 15 080: two workgroups per CU), which its fallback paths need; the
// path that peaked rows take -- masses, the candidates within HEAD_RANGE of the maximum, crossing / softmax /
// confidence / draw over that list -- reads every element exactly twice and needs no row storage beyond the registers
// of 256 threads (NV4 float4 each).  The arithmetic is sample_kernel's: same temperature division, same maximum,
// integer masses, the same candidate records, list_tail_wave for <= 64
// candidates and the operations of sample_row's list mode for up to one candidate per thread; every sum is an integer
// sum or a double sum of floats of like magnitude (exact), so conf / x0 are bit-identical whatever the order.  A row
// this stage cannot finish (more than SH candidates, or candidates that do not hold top_p of the mass) is marked
// x0 = -1 for the second stage (sample_kernel over all M rows: a workgroup whose row is done returns at once; such a
// launch costs 12 us at M = 16 896), untouched otherwise.  No state outlives a launch.
constexpr int SH = 256;

template <int NV4>
__global__ __launch_bounds__(SH) void sample_head_kernel(const float* __restrict__ logits, int64_t ldl, int V,
                                                         float temperature, float top_p, int conf_mode,
                                                         const float* __restrict__ noise, int64_t ldn, PhiloxNoise pn,
                                                         float* __restrict__ conf, int64_t* __restrict__ x0) {
  __shared__ uint32_t cand[SH];
  __shared__ unsigned long long cfx[SH];
  __shared__ int cidx[SH];
  __shared__ Scratch sc;
  Scratch* s = &sc;
  const int tid = threadIdx.x, m = blockIdx.x;
  auto give_up = [&](int64_t mark = -1) {  // block-uniform: x0 = -1 hands the row to the second stage (-2: and it is flat)
    if (tid == 0) x0[m] = mark;
  };
  if (tid == 0) s->ncand = 0;

  // 1. the row and its maximum; the division by the temperature (generation_utils.py:60-61) follows below
  const float* lr = logits + (int64_t)m * ldl;
  const int v4 = V >> 2;
  float4 r[NV4];
#pragma unroll
  for (int u = 0; u < NV4; ++u) {
    const int i = tid + u * SH;
    r[u] = i < v4 ? *(const float4*)(lr + 4 * i) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  }
  float tail = (4 * v4 + tid < V) ? lr[4 * v4 + tid] : -INFINITY;   // V % 4 elements, threads 0..2
  // The maximum of the RAW row first: division by a positive temperature is monotone and correctly rounded, so
  // max(l / T) = max(l) / T bit for bit, and whether a row is flat can be told before paying for V divisions.
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < NV4; ++u) mx = fmaxf(fmaxf(mx, fmaxf(r[u].x, r[u].y)), fmaxf(r[u].z, r[u].w));
  mx = block_max<SH>(fmaxf(mx, tail), s, tid);

  // 2. flat rows leave here, before the expensive passes: more than SH float4 pieces that hold an element within
  // (about) HEAD_RANGE of the maximum mean more than SH candidates.  A heuristic may err either way without touching
  // a result: both stages compute the same function.  (The second stage reads the row again: +5 % on rows like
  // T = 0.5 on random-init logits, which this stage cannot help.)
  {
    const float reach = HEAD_RANGE * (temperature > 0.f ? temperature : 1.f);
    int c4 = 0;
#pragma unroll
    for (int u = 0; u < NV4; ++u)
      c4 += fmaxf(fmaxf(r[u].x, r[u].y), fmaxf(r[u].z, r[u].w)) - mx > -reach;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c4 += __shfl_xor(c4, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) s->i32[tid >> 6] = c4;
    __syncthreads();
    c4 = 0;
#pragma unroll
    for (int w = 0; w < SH / 64; ++w) c4 += s->i32[w];
    if (c4 > SH) { give_up(-2); return; }
  }
  if (temperature > 0.f) {
#pragma unroll
    for (int u = 0; u < NV4; ++u) {
      r[u].x = r[u].x / temperature; r[u].y = r[u].y / temperature;
      r[u].z = r[u].z / temperature; r[u].w = r[u].w / temperature;
    }
    tail = tail / temperature;
    mx = mx / temperature;
  }

  // 3. masses and the candidate records
  unsigned long long zl = 0, hl = 0;      // row mass; mass of the candidates this thread recorded
  int mine = 0;  // a thread records at most 8 of its 64 elements, then poisons the count (one LDS address serves them all)
  auto element = [&](float l, int i) {
    const unsigned long long fx = mass_fx_nb(l, mx);
    zl += fx;
    if (l - mx > -HEAD_RANGE) {
      if (mine < 8) {
        const int slot = atomicAdd(&s->ncand, 1);
        if (slot < SH) { cand[slot] = fkey(l); cfx[slot] = fx; cidx[slot] = i; hl += fx; }
      } else if (mine == 8) {
        atomicAdd(&s->ncand, SH + 1);  // the candidate set is incomplete: the row goes to the second stage
      }
      ++mine;
    }
  };
#pragma unroll
  for (int u = 0; u < NV4; ++u) {
    // (a wave whose 256 elements all lie below the mass range -- most of a peaked row -- skips their exp chains: they
    // would add zeros)
    const float top4 = fmaxf(fmaxf(r[u].x, r[u].y), fmaxf(r[u].z, r[u].w));
    if (__any(top4 - mx > -MASS_RANGE)) {
      element(r[u].x, 4 * (tid + u * SH));
      element(r[u].y, 4 * (tid + u * SH) + 1);
      element(r[u].z, 4 * (tid + u * SH) + 2);
      element(r[u].w, 4 * (tid + u * SH) + 3);
    }
    __builtin_amdgcn_sched_barrier(0);  // (one float4 at a time: interleaving all the exp chains costs registers)
  }
  element(tail, 4 * v4 + tid);
  block_sum2_u64<SH>(zl, hl, s, tid);  // (its barriers also publish the candidates)
  const unsigned long long T = (unsigned long long)((double)top_p * (double)zl);  // floor, as nucleus_crossing
  const int c = s->ncand;
  if (c < 1 || c > SH || !(hl > T)) { give_up(); return; }

  const float* qr = noise + (int64_t)m * ldn;
  const unsigned long long q0 = (unsigned long long)m * V;
  if (c <= 64) {  // wave 0 finishes the row exactly as sample_kernel's head path does
    if (tid < 64) list_tail_wave(cand, cfx, cidx, c, T, mx, V, temperature, conf_mode, qr, pn, q0, tid, conf + m, x0 + m);
    return;
  }

  // 4. up to one candidate per thread: sample_row's list mode, operation for operation
  uint32_t tau = 0;
  unsigned long long above = 0;
  int n_tau = 0;
  if (resolve_candidates<SH>(cand, cfx, c, 0ull, T, s, tid, &tau, &above, &n_tau) <= 0) { give_up(); return; }
  const unsigned long long fxt = mass_fx(fkey_inv(tau), mx);
  const unsigned long long t_keep = fxt ? (T - above) / fxt + 1ull : (unsigned long long)n_tau;
  int my_idx = 0;
  float my_e = 0.f;
  if (tid < c) {
    const uint32_t kj = cand[tid];
    my_idx = cidx[tid];
    bool keep = kj > tau;
    if (kj == tau) {
      keep = true;
      if (t_keep < (unsigned long long)n_tau) {  // only the first t_keep (by index) of the equal logits stay
        unsigned long long rank = 0;
        for (int i = 0; i < c; ++i) rank += (cand[i] == tau && cidx[i] < my_idx);
        keep = rank < t_keep;
      }
    }
    if (keep) my_e = exp_neg(fkey_inv(kj) - mx);
  }
  const float Zf = (float)block_sum<SH>((double)my_e, s, tid);
  double ps = 0.0, ent = 0.0;
  float t1 = -INFINITY, t2 = -INFINITY, bestp = -INFINITY, my_p = 0.f;
  int besti = 0x7fffffff;
  if (my_e > 0.f) {
    my_p = my_e / Zf;
    ps = (double)my_p;
    if (conf_mode == DARK_CONF_NEG_ENTROPY) ent = (double)(my_p * log_pos(my_p + 1e-10f));
    t1 = my_p;
    bestp = my_p;
    besti = my_idx;
  }
  block_sum2<SH>(ps, ent, s, tid);
  const float psum = (float)ps;
  const float entropy = (float)ent;
  if (temperature > 0.f) {
    bestp = -INFINITY;
    besti = 0x7fffffff;
    if (my_p > 0.f) {
      bestp = (my_p / psum) / (pn.nthreads ? philox_exponential(pn, q0 + my_idx) : qr[my_idx]);
      besti = my_idx;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bestp, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > bestp || (ov == bestp && oi < besti)) { bestp = ov; besti = oi; }
    const float o1 = __shfl_xor(t1, o, 64), o2 = __shfl_xor(t2, o, 64);
    const float n1 = fmaxf(t1, o1), n2 = fmaxf(fminf(t1, o1), fmaxf(t2, o2));
    t1 = n1; t2 = n2;
  }
  __syncthreads();
  if ((tid & 63) == 0) {
    s->f32a[tid >> 6] = bestp; s->i32[tid >> 6] = besti;
    s->f32b[tid >> 6] = t1; s->f32c[tid >> 6] = t2;
  }
  __syncthreads();
  if (tid == 0) {
    float bv = s->f32a[0]; int bi = s->i32[0];
    float a1 = s->f32b[0], a2 = s->f32c[0];
    for (int w = 1; w < SH / 64; ++w) {
      const float ov = s->f32a[w]; const int oi = s->i32[w];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      const float o1 = s->f32b[w], o2 = s->f32c[w];
      const float n1 = fmaxf(a1, o1), n2 = fmaxf(fminf(a1, o1), fmaxf(a2, o2));
      a1 = n1; a2 = n2;
    }
    if (bi < 0 || bi >= V) bi = 0;
    if (a2 < 0.f) a2 = 0.f;
    conf[m] = conf_mode == DARK_CONF_NEG_ENTROPY ? entropy : (conf_mode == DARK_CONF_MARGIN ? a1 - a2 : 0.f);
    x0[m] = (int64_t)bi;
    s->bin = bi;
  }
  if (conf_mode == DARK_CONF_PROB) {  // the probability of the winner lives in its owner thread
    __syncthreads();
    if (my_e > 0.f && my_idx == s->bin) conf[m] = my_p;
  }
}

// GROW = false: the row lives in LDS (V <= SAMPLE_LDS_MAX_V, every vocabulary the benchmarked configs use).
// GROW = true: a vocabulary whose fp32 row does not fit in a CU's 160 KiB of LDS (WN18RR: V = 41 066 = 164 KB) keeps
// the row in the caller-registered device workspace (dark_sampler_set_row_workspace), one [vp] slice per workgroup;
// the program is the same, pass for pass, so the results are the same function of the row.  Global memory written by
// one wave is read by the others behind __syncthreads(): all waves of a workgroup share their CU's L1.
template <bool GROW>
__device__ __forceinline__ void sample_row(const float* __restrict__ logits, int64_t ldl, int V,
                                           float temperature, float top_p, int top_k, int conf_mode,
                                           const float* __restrict__ noise, int64_t ldn, PhiloxNoise pn,
                                           float* __restrict__ conf, int64_t* __restrict__ x0,
                                           const int32_t* __restrict__ row_ids,
                                           const int32_t* __restrict__ n_dev, float* __restrict__ row_ws,
                                           bool flat_hint = false) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (n_dev && (int)blockIdx.x >= *n_dev) return;  // row count from device memory (block-uniform)
  const int vp = (V + 3) & ~3;
  float* row = GROW ? row_ws + (size_t)blockIdx.x * vp : (float*)smem;
  unsigned long long* hist = (unsigned long long*)(smem + (GROW ? 0 : (size_t)vp * 4));  // NLIN u64; radix hist/cnt alias it
  unsigned int* cnth = (unsigned int*)(hist + NRAD);
  unsigned long long* cfx = hist + NLIN;
  uint32_t* cand = (uint32_t*)(cfx + CAND_CAP);
  int* cidx = (int*)(cand + CAND_CAP);
  Scratch* s = (Scratch*)(cidx + CAND_CAP);
  const int tid = threadIdx.x;
  // row_ids (dark_sample_rows_indexed): logits holds a compact list of rows; noise, Philox element index and the
  // outputs belong to row row_ids[blockIdx.x] of the full [M, V] problem
  const int m = row_ids ? row_ids[blockIdx.x] : blockIdx.x;
  const float* lr = logits + (int64_t)blockIdx.x * ldl;

  // 1. temperature (generation_utils.py:60-61) and row max (a row the first stage found flat: its minimum too, negated)
  float mx = -INFINITY, nmn = -INFINITY;
  if ((ldl & 3) == 0 && ((uintptr_t)logits & 15) == 0) {
    // the row is the only HBM traffic of this kernel: a thread's loads are issued back to back, eight at a time
    // (one at a time, each iteration waits out a full memory latency: 58 % of the wave cycles were spent so)
    const int v4 = V >> 2;
    constexpr int U = 8;
    for (int base = tid; base < v4; base += ST * U) {
      float4 buf[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * ST;
        if (i < v4) buf[u] = *(const float4*)(lr + 4 * i);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * ST;
        if (i < v4) {
          float4 l = buf[u];
          if (temperature > 0.f) { l.x = l.x / temperature; l.y = l.y / temperature; l.z = l.z / temperature; l.w = l.w / temperature; }
          *(float4*)(row + 4 * i) = l;
          mx = fmaxf(fmaxf(mx, fmaxf(l.x, l.y)), fmaxf(l.z, l.w));
          if (flat_hint) nmn = fmaxf(fmaxf(nmn, fmaxf(-l.x, -l.y)), fmaxf(-l.z, -l.w));
        }
      }
    }
    for (int i = 4 * v4 + tid; i < V; i += ST) {
      float l = lr[i];
      if (temperature > 0.f) l = l / temperature;
      row[i] = l;
      mx = fmaxf(mx, l);
      nmn = fmaxf(nmn, -l);
    }
  } else {
    for (int i = tid; i < V; i += ST) {
      float l = lr[i];
      if (temperature > 0.f) l = l / temperature;
      row[i] = l;
      mx = fmaxf(mx, l);
      nmn = fmaxf(nmn, -l);
    }
  }
  if (flat_hint) block_max2(mx, nmn, s, tid);  // (block-uniform)
  else mx = block_max(mx, s, tid);

  int list_c = -1;   // >= 0: list mode (see below)
  uint32_t cut = 0;  // keys below this are filtered (applied inside the softmax pass; 0 = nothing)
  int my_idx = 0;    // list mode: vocabulary index of this thread's candidate
  float my_e = 0.f;  // list mode: exp(l - max) of this thread's candidate if it is kept, else 0

  // 2. nucleus filter (generation_utils.py:37-48)
  if (top_p >= 0.f && top_p < 1.f) {  // 0 keeps the top-1 token alone, like the reference's `cum > 0` (:41-44)
    uint32_t tau = 0;
    unsigned long long above = 0, T = 0;
    int n_tau = 0;
    int head_c = -1;
    int rc = nucleus_crossing(row, V, mx, top_p, hist, cand, cfx, cidx, s, tid, &tau, &above, &n_tau, &T, &head_c,
                              !(top_k > 0 && top_k < V), flat_hint, -nmn);
    if (rc == 2) {  // block-uniform: waves 1..7 are done, wave 0 finishes the row without further barriers
      if (tid < 64)
        list_tail_wave(cand, cfx, cidx, head_c, T, mx, V, temperature, conf_mode, noise + (int64_t)m * ldn, pn,
                       (unsigned long long)m * V, tid, conf + m, x0 + m);
      return;
    }
    if (rc < 0) rc = select_crossing<true>(row, V, mx, T, hist, cnth, s, tid, &tau, &above, &n_tau) ? 1 : 0;
    if (rc > 0) {
      const unsigned long long fx = mass_fx(fkey_inv(tau), mx);
      unsigned long long t_keep = fx ? (T - above) / fx + 1ull : (unsigned long long)n_tau;
      if (head_c >= 0 && !(top_k > 0 && top_k < V)) {
        // list mode: the kept set lies inside the gathered head candidates, one per thread; the softmax,
        // confidence and draw below run on that list and the row is never rewritten or rescanned
        list_c = head_c;
        if (tid < head_c) {
          const uint32_t kj = cand[tid];
          my_idx = cidx[tid];
          bool keep = kj > tau;
          if (kj == tau) {
            keep = true;
            if (t_keep < (unsigned long long)n_tau) {  // only the first t_keep (by index) of the equal logits stay
              unsigned long long rank = 0;
              for (int i = 0; i < head_c; ++i) rank += (cand[i] == tau && cidx[i] < my_idx);
              keep = rank < t_keep;
            }
          }
          if (keep) my_e = exp_neg(fkey_inv(kj) - mx);
        }
      } else if (t_keep >= (unsigned long long)n_tau) {
        if (top_k > 0 && top_k < V) {
          for (int i = tid; i < V; i += ST)
            if (fkey(row[i]) < tau) row[i] = DARK_FLT_LOWEST;
        } else {
          cut = tau;  // no pass of its own: the softmax pass below gives these elements the 0 that finfo.min would
        }
      } else {
        // some, not all, of the equal-valued boundary elements are kept: the first t_keep by index
        const int chunk = (V + ST - 1) / ST;
        const int lo = min(tid * chunk, V), hi = min(lo + chunk, V);
        int c = 0;
        for (int i = lo; i < hi; ++i) c += fkey(row[i]) == tau;
        int inc = c;
        const int lane = tid & 63;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(inc, o, 64);
          if (lane >= o) inc += t;
        }
        __syncthreads();
        if (lane == 63) s->i32[tid >> 6] = inc;
        __syncthreads();
        int rank = inc - c;
        for (int w = 0; w < (tid >> 6); ++w) rank += s->i32[w];
        for (int i = lo; i < hi; ++i) {
          const uint32_t k = fkey(row[i]);
          if (k < tau) row[i] = DARK_FLT_LOWEST;
          else if (k == tau) { if ((unsigned long long)rank >= t_keep) row[i] = DARK_FLT_LOWEST; ++rank; }
        }
      }
    }
    __syncthreads();
  }

  // 3. top-k filter (generation_utils.py:50-55): keep logits >= k-th largest value
  if (top_k > 0 && top_k < V) {
    uint32_t tau;
    unsigned long long above;
    int n_tau;
    if (select_crossing<false>(row, V, mx, (unsigned long long)(top_k - 1), hist, cnth, s, tid, &tau, &above,
                               &n_tau)) {
      for (int i = tid; i < V; i += ST)
        if (fkey(row[i]) < tau) row[i] = DARK_FLT_LOWEST;
    }
    __syncthreads();
  }

  // 4. softmax (generation_utils.py:66): p_i = exp(l_i - max) / Z; the row now holds exp(l_i - max)
  // Filtered entries (finfo.min) are exactly 0 after the softmax and contribute exactly 0 to every sum
  // below, so the passes from here on only do work for the kept set: the branches are
  // wave-uniform-false for almost every wave of a filtered row, and the noise row is only touched at the
  // kept positions.
  const bool list = list_c >= 0;
  double zs = 0.0;
  if (list) {
    zs = (double)my_e;
  } else {
    // (here and in the two passes below: four of a thread's elements per trip, their reads issued together; the
    // elements are visited in the same order, so every sum is the same sum)
    for (int i0 = tid; i0 < V; i0 += 4 * ST) {
      float lv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) lv[u] = i0 + u * ST < V ? row[i0 + u * ST] : -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * ST;
        if (i < V) {
          const float d = lv[u] - mx;
          float e = 0.f;
          if (d > -104.f && fkey(lv[u]) >= cut) {  // exp(-104) < 2^-149: exactly 0 in fp32 below
            e = exp_neg(d);
            zs += (double)e;
          }
          row[i] = e;
        }
      }
    }
  }
  const float Zf = (float)block_sum(zs, s, tid);

  double ps = 0.0, ent = 0.0;
  float t1 = -INFINITY, t2 = -INFINITY;  // two largest probabilities
  float bestp = -INFINITY;
  int besti = 0x7fffffff;
  float my_p = 0.f;
  if (list) {
    if (my_e > 0.f) {
      my_p = my_e / Zf;
      ps = (double)my_p;
      if (conf_mode == DARK_CONF_NEG_ENTROPY) ent = (double)(my_p * log_pos(my_p + 1e-10f));
      t1 = my_p;
      bestp = my_p;
      besti = my_idx;
    }
  } else {
    for (int i0 = tid; i0 < V; i0 += 4 * ST) {
      float ev[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ev[u] = i0 + u * ST < V ? row[i0 + u * ST] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * ST;
        const float e = ev[u];
        if (e > 0.f) {
          const float p = e / Zf;
          row[i] = p;
          ps += (double)p;
          if (conf_mode == DARK_CONF_NEG_ENTROPY) ent += (double)(p * log_pos(p + 1e-10f));
          if (conf_mode == DARK_CONF_MARGIN) { if (p > t1) { t2 = t1; t1 = p; } else if (p > t2) { t2 = p; } }
          if (!(temperature > 0.f) && p > bestp) { bestp = p; besti = i; }  // (with T > 0 the draw below decides)
        }
      }
    }
  }
  block_sum2(ps, ent, s, tid);
  const float psum = (float)ps;
  const float entropy = (float)ent;

  // 5. draw (generation_utils.py:68-75): argmax of (p / psum) / q; entries with p == 0 give 0 and never win
  if (temperature > 0.f) {
    const float* qr = noise + (int64_t)m * ldn;
    const unsigned long long q0 = (unsigned long long)m * V;  // the lazy noise is an [M, V] fill
    bestp = -INFINITY;
    besti = 0x7fffffff;
    if (list) {
      if (my_p > 0.f) {
        bestp = (my_p / psum) / (pn.nthreads ? philox_exponential(pn, q0 + my_idx) : qr[my_idx]);
        besti = my_idx;
      }
    } else {
      // (q, t) of element q0 + i is carried along instead of a 64-bit division per element
      unsigned long long pq = 0;
      unsigned int pt = 0;
      if (pn.nthreads) {
        pq = (q0 + tid) / pn.nthreads;
        pt = (unsigned int)(q0 + tid - pq * pn.nthreads);
      }
      if (pn.nthreads) {
        for (int i = tid; i < V; i += ST) {
          const float p = row[i];
          if (p > 0.f) {
            const float r = (p / psum) / philox_exponential_qt(pn, pq, pt);
            if (r > bestp) { bestp = r; besti = i; }
          }
          pt += ST;  // pt < nthreads and ST < 2^31: no wrap
          while (pt >= pn.nthreads) { pt -= pn.nthreads; ++pq; }
        }
      } else {
        // the noise of the kept positions of four trips is requested before the first quotient is formed (one trip at
        // a time, every trip waited out a memory latency: this pass was a fifth of a flat row's lifetime)
        for (int i0 = tid; i0 < V; i0 += 4 * ST) {
          float pv[4], qv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) pv[u] = i0 + u * ST < V ? row[i0 + u * ST] : 0.f;
#pragma unroll
          for (int u = 0; u < 4; ++u) qv[u] = pv[u] > 0.f ? qr[i0 + u * ST] : 1.f;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (pv[u] > 0.f) {
              const float r = (pv[u] / psum) / qv[u];
              if (r > bestp) { bestp = r; besti = i0 + u * ST; }
            }
          }
        }
      }
    }
  }
  // block arg-max, ties -> lowest index; merge the (top1, top2) pairs alongside
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bestp, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > bestp || (ov == bestp && oi < besti)) { bestp = ov; besti = oi; }
    const float o1 = __shfl_xor(t1, o, 64), o2 = __shfl_xor(t2, o, 64);
    const float n1 = fmaxf(t1, o1), n2 = fmaxf(fminf(t1, o1), fmaxf(t2, o2));
    t1 = n1; t2 = n2;
  }
  __syncthreads();
  if ((tid & 63) == 0) {
    s->f32a[tid >> 6] = bestp; s->i32[tid >> 6] = besti;
    s->f32b[tid >> 6] = t1; s->f32c[tid >> 6] = t2;
  }
  __syncthreads();
  if (tid == 0) {
    float bv = s->f32a[0]; int bi = s->i32[0];
    float a1 = s->f32b[0], a2 = s->f32c[0];
    for (int w = 1; w < ST / 64; ++w) {
      const float ov = s->f32a[w]; const int oi = s->i32[w];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      const float o1 = s->f32b[w], o2 = s->f32c[w];
      const float n1 = fmaxf(a1, o1), n2 = fmaxf(fminf(a1, o1), fmaxf(a2, o2));
      a1 = n1; a2 = n2;
    }
    if (bi < 0 || bi >= V) bi = 0;  // all-NaN row: keep the index in range
    if (a2 < 0.f) a2 = 0.f;         // fewer than two non-zero probabilities: the runner-up is a zero
    float c;
    if (conf_mode == DARK_CONF_NEG_ENTROPY) c = entropy;
    else if (conf_mode == DARK_CONF_MARGIN) c = a1 - a2;
    else c = list ? 0.f : row[bi];
    conf[m] = c;
    x0[m] = (int64_t)bi;
    s->bin = bi;
  }
  if (list && conf_mode == DARK_CONF_PROB) {  // list mode: the probability of the winner lives in its owner thread
    __syncthreads();
    if (my_e > 0.f && my_idx == s->bin) conf[m] = my_p;
  }
}

// (six waves per SIMD = two workgroups per CU, which is what the row's LDS allows: the register budget is 80)
__global__ __launch_bounds__(ST) __attribute__((amdgpu_waves_per_eu(6, 6))) void sample_kernel(const float* __restrict__ logits, int64_t ldl, int V,
                                                    float temperature, float top_p, int top_k, int conf_mode,
                                                    const float* __restrict__ noise, int64_t ldn, PhiloxNoise pn,
                                                    float* __restrict__ conf, int64_t* __restrict__ x0,
                                                    const int32_t* __restrict__ row_ids,
                                                    const int32_t* __restrict__ n_dev, int second_stage) {
  // second_stage (of the two-stage sampler): only the rows sample_head_kernel marked with x0 = -1 / -2; the other workgroups
  // return at once (12 us for 16 896 of them)
  // (-2: a row the first stage found flat)
  bool flat = false;
  if (second_stage) {
    const int64_t mark = x0[blockIdx.x];
    if (mark != -1 && mark != -2) return;
    flat = mark == -2;
  }
  sample_row<false>(logits, ldl, V, temperature, top_p, top_k, conf_mode, noise, ldn, pn, conf, x0, row_ids, n_dev, nullptr,
                    flat);
}

__global__ __launch_bounds__(ST) __attribute__((amdgpu_waves_per_eu(6, 6))) void sample_bigv_kernel(const float* __restrict__ logits, int64_t ldl, int V,
                                                         float temperature, float top_p, int top_k, int conf_mode,
                                                         const float* __restrict__ noise, int64_t ldn, PhiloxNoise pn,
                                                         float* __restrict__ conf, int64_t* __restrict__ x0,
                                                         const int32_t* __restrict__ row_ids,
                                                         const int32_t* __restrict__ n_dev, float* __restrict__ row_ws) {
  sample_row<true>(logits, ldl, V, temperature, top_p, top_k, conf_mode, noise, ldn, pn, conf, x0, row_ids, n_dev, row_ws);
}

// Row workspace of the large-vocabulary sampler, one registration per device (see dark_hip.h).
namespace {
struct RowWorkspace { std::atomic<void*> ptr{nullptr}; std::atomic<size_t> bytes{0}; };
RowWorkspace g_row_ws[DARK_MAX_DEVICES];
}  // namespace

static std::atomic<int> g_sampler_stages{0};   // 0 = automatic, 1 = single-stage only (A/B and tests)
void set_sampler_stages(int v) { g_sampler_stages.store(v, std::memory_order_relaxed); }

int set_sampler_row_workspace(void* ptr, size_t bytes) {
  int dev = 0;
  DARK_CHECK_HIP(hipGetDevice(&dev));
  DARK_CHECK_ARG(dev >= 0 && dev < DARK_MAX_DEVICES, "device ordinal %d out of range", dev);
  DARK_CHECK_ARG((ptr == nullptr) == (bytes == 0) && ((uintptr_t)ptr & 15) == 0,
                 "sampler_set_row_workspace: needs a 16-byte aligned device pointer with its size, or (NULL, 0)");
  g_row_ws[dev].bytes.store(0, std::memory_order_release);
  g_row_ws[dev].ptr.store(ptr, std::memory_order_release);
  g_row_ws[dev].bytes.store(bytes, std::memory_order_release);
  return DARK_OK;
}

int launch_philox_exponential(uint64_t seed, uint64_t offset, uint32_t nthreads, uint64_t numel, float* out,
                              hipStream_t st) {
  DARK_CHECK_ARG(out && nthreads > 0 && numel > 0 && numel < (1ull << 40), "philox_exponential: bad argument");
  const PhiloxNoise pn{seed, offset, nthreads};
  hipLaunchKernelGGL(philox_exponential_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, st, pn,
                     (unsigned long long)numel, out);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int launch_sample(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p, int top_k,
                  int conf_mode, const float* noise, int64_t ldn, float* conf, int64_t* x0, hipStream_t st) {
  return launch_sample_philox(logits, ldl, M, V, temperature, top_p, top_k, conf_mode, noise, ldn, 0, 0, 0, conf, x0,
                              st);
}

int launch_sample_philox(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p, int top_k,
                         int conf_mode, const float* noise, int64_t ldn, uint64_t seed, uint64_t offset,
                         uint32_t nthreads, float* conf, int64_t* x0, hipStream_t st) {
  return launch_sample_indexed(logits, ldl, M, V, temperature, top_p, top_k, conf_mode, nullptr, noise, ldn, seed, offset,
                               nthreads, conf, x0, st, nullptr);
}

int launch_sample_indexed(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p, int top_k,
                          int conf_mode, const int32_t* row_ids, const float* noise, int64_t ldn, uint64_t seed,
                          uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0, hipStream_t st,
                          const int32_t* n_dev) {
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(V >= 2 && V <= DARK_SAMPLE_MAX_V, "sample_rows: V=%d must be in [2, %d]", V, DARK_SAMPLE_MAX_V);
  DARK_CHECK_ARG(ldl >= V, "sample_rows: ldl < V");
  DARK_CHECK_ARG(logits && conf && x0, "sample_rows: null pointer");
  DARK_CHECK_ARG(conf_mode >= 0 && conf_mode <= 2, "sample_rows: bad conf_mode %d", conf_mode);
  DARK_CHECK_ARG(!(temperature > 0.f) || nthreads > 0 || (noise && ldn >= V),
                 "sample_rows: temperature > 0 needs noise [M,V] or a Philox state");
  const PhiloxNoise pn{seed, offset, nthreads};
  DARK_CHECK_ARG(temperature >= 0.f, "sample_rows: negative temperature");
  DARK_CHECK_ARG(nthreads <= (1u << 30), "sample_rows: Philox grid of %u threads", nthreads);
  static_assert(NRAD * 12 <= NLIN * 8, "the radix histograms alias the linear histogram");
  if (V > DARK_SAMPLE_LDS_MAX_V) {  // the row does not fit in LDS: rows live in the registered workspace
    constexpr int smem_big = NLIN * 8 + CAND_CAP * 16 + (int)sizeof(Scratch);
    static SmemOptIn optin_big;   // (no opt-in needed at this size; the call checks that the stream's device is current)
    DARK_TRY(optin_big.ensure((const void*)sample_bigv_kernel, smem_big, st));
    int dev = 0;
    DARK_CHECK_HIP(hipGetDevice(&dev));
    const size_t need = (size_t)M * (size_t)((V + 3) & ~3) * 4;
    const size_t have = g_row_ws[dev].bytes.load(std::memory_order_acquire);
    float* ws = (float*)g_row_ws[dev].ptr.load(std::memory_order_acquire);
    DARK_CHECK_ARG(ws && have >= need,
                   "sample_rows: V=%d exceeds the LDS-resident sampler (V <= %d): register a row workspace of at least "
                   "%zu bytes on device %d with dark_sampler_set_row_workspace (registered: %zu)",
                   V, DARK_SAMPLE_LDS_MAX_V, need, dev, have);
    hipLaunchKernelGGL(sample_bigv_kernel, dim3(M), dim3(ST), smem_big, st, logits, ldl, V, temperature, top_p, top_k,
                       conf_mode, noise, ldn, pn, conf, x0, row_ids, n_dev, ws);
    DARK_CHECK_LAUNCH();
    return DARK_OK;
  }
  const int smem = ((V + 3) & ~3) * 4 + NLIN * 8 + CAND_CAP * 16 + (int)sizeof(Scratch);
  static SmemOptIn optin;
  DARK_TRY(optin.ensure((const void*)sample_kernel, smem, st));
  // Two-stage form (see sample_head_kernel): a plain [M, V] call with the nucleus filter on and top-k off, rows that
  // load 16 bytes a lane and fit the first stage's registers.
  if (g_sampler_stages.load(std::memory_order_relaxed) != 1 && !row_ids && !n_dev && top_p >= 0.f && top_p < 1.f &&
      !(top_k > 0 && top_k < V) && (ldl & 3) == 0 && ((uintptr_t)logits & 15) == 0 && (V >> 2) <= 26 * SH) {
    if ((V >> 2) <= 16 * SH)
      hipLaunchKernelGGL(sample_head_kernel<16>, dim3(M), dim3(SH), 0, st, logits, ldl, V, temperature, top_p, conf_mode,
                         noise, ldn, pn, conf, x0);
    else
      hipLaunchKernelGGL(sample_head_kernel<26>, dim3(M), dim3(SH), 0, st, logits, ldl, V, temperature, top_p, conf_mode,
                         noise, ldn, pn, conf, x0);
    DARK_CHECK_LAUNCH();
    hipLaunchKernelGGL(sample_kernel, dim3(M), dim3(ST), smem, st, logits, ldl, V, temperature, top_p, top_k, conf_mode,
                       noise, ldn, pn, conf, x0, row_ids, n_dev, 1);
    {  // the first stage is already enqueued and has marked its unfinished rows with x0 = -1: say so if the second is refused
      const hipError_t e2 = hipGetLastError();
      if (e2 != hipSuccess) {
        set_error("sample_rows: the second stage of the two-stage sampler could not be launched (%s): conf / x0 are "
                  "UNDEFINED for this call (rows the first stage did not finish hold x0 = -1 or -2) and must not be committed",
                  hipGetErrorString(e2));
        return DARK_ERR_HIP;
      }
    }
    return DARK_OK;
  }
  hipLaunchKernelGGL(sample_kernel, dim3(M), dim3(ST), smem, st, logits, ldl, V, temperature, top_p, top_k,
                     conf_mode, noise, ldn, pn, conf, x0, row_ids, n_dev, 0);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ================================================================== vocabulary-sharded head (BASELINE config 3)
// The LM head's columns are split over S shards (GPUs); each shard holds logits [M, Vs] for its column range and
// never sends them anywhere.  What the sampler needs from a row crosses shards in two small exchanges:
//   1. the row maximum of logits / T  (max is exact and associative: all-reduce of M floats);
//   2. per shard, relative to that GLOBAL maximum: the fixed-point probability mass of its columns (integers: sums
//      are exact in any order), and the few elements within HEAD_RANGE of the maximum (key, mass, vocabulary id).
// The owner of the row adds the masses, concatenates the candidates and finishes exactly as sample_kernel's head
// path does (list_tail_wave: crossing, kept set, softmax, confidence, Philox draw) -- bit-identical to the unsharded
// sampler.  Rows whose head set does not hold top_p of the mass, or holds more than 64 elements (flat rows), are
// flagged; the caller gathers those rows' logits and runs dark_sample_rows_indexed on them.
constexpr int SHARD_CAP = DARK_SHARD_CAND_CAP;  // candidates a shard can report per row
struct ShardPart {
  unsigned long long Z, hl;  // mass of the shard's columns; mass of the reported candidates
  int n, pad;                // candidates found (> SHARD_CAP: not all could be reported)
  unsigned long long fx[SHARD_CAP];
  uint32_t key[SHARD_CAP];
  int idx[SHARD_CAP];
};
static_assert(sizeof(ShardPart) == DARK_SHARD_PART_BYTES, "ShardPart layout is part of the C ABI");

__global__ __launch_bounds__(256) void shard_row_max_kernel(const float* __restrict__ logits, int64_t ldl, int Vs,
                                                            float temperature, float* __restrict__ mx) {
  __shared__ float wmax[4];
  const float* lr = logits + (int64_t)blockIdx.x * ldl;
  float v = -INFINITY;
  for (int i = threadIdx.x; i < Vs; i += 256) {
    float l = lr[i];
    if (temperature > 0.f) l = l / temperature;  // the sampler's own operation order (generation_utils.py:60-61)
    v = fmaxf(v, l);
  }
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) mx[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

__global__ __launch_bounds__(256) void shard_partials_kernel(const float* __restrict__ logits, int64_t ldl, int Vs,
                                                             int col0, float temperature,
                                                             const float* __restrict__ mx_all,
                                                             ShardPart* __restrict__ parts) {
  __shared__ ShardPart sp;
  __shared__ unsigned long long wsum[2][4];
  const int tid = threadIdx.x;
  const float* lr = logits + (int64_t)blockIdx.x * ldl;
  const float mx = mx_all[blockIdx.x];
  if (tid == 0) sp.n = 0;
  __syncthreads();
  unsigned long long zl = 0, hl = 0;
  for (int i = tid; i < Vs; i += 256) {
    float l = lr[i];
    if (temperature > 0.f) l = l / temperature;
    const unsigned long long fx = mass_fx(l, mx);
    zl += fx;
    if (l - mx > -HEAD_RANGE) {
      const int slot = atomicAdd(&sp.n, 1);
      if (slot < SHARD_CAP) { sp.key[slot] = fkey(l); sp.fx[slot] = fx; sp.idx[slot] = col0 + i; hl += fx; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { zl += __shfl_xor(zl, o, 64); hl += __shfl_xor(hl, o, 64); }
  if ((tid & 63) == 0) { wsum[0][tid >> 6] = zl; wsum[1][tid >> 6] = hl; }
  __syncthreads();
  ShardPart* out = parts + blockIdx.x;
  if (tid == 0) {
    out->Z = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
    out->hl = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
    out->n = sp.n;
    out->pad = 0;
  }
  if (tid < SHARD_CAP) {
    const bool have = tid < sp.n;
    out->key[tid] = have ? sp.key[tid] : 0u;
    out->fx[tid] = have ? sp.fx[tid] : 0ull;
    out->idx[tid] = have ? sp.idx[tid] : 0;
  }
}

// One wave per row: merge the S shard records of row m (parts[s * ldp + m]) and finish the row, or flag it.
__global__ __launch_bounds__(256) void sample_from_shards_kernel(const ShardPart* __restrict__ parts, int64_t ldp, int S, int M, int V,
                                                                 float temperature, float top_p, int top_k,
                                                                 int conf_mode, const float* __restrict__ mx_all,
                                                                 const float* __restrict__ noise, int64_t ldn,
                                                                 PhiloxNoise pn, float* __restrict__ conf,
                                                                 int64_t* __restrict__ x0,
                                                                 uint8_t* __restrict__ flags) {
  __shared__ unsigned long long cfx_s[4][64];
  __shared__ uint32_t cand_s[4][64];
  __shared__ int cidx_s[4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + w;
  if (m >= M) return;  // wave-uniform
  const bool nucleus = top_p >= 0.f && top_p < 1.f && !(top_k > 0 && top_k < V);
  unsigned long long Z = 0, hl = 0;
  int c = 0;
  bool over = false;
  for (int s0 = 0; s0 < S; s0 += 64) {  // S <= 64 in practice: one pass
    const int s = s0 + lane;
    unsigned long long z1 = 0, h1 = 0;
    int n1 = 0;
    if (s < S) {
      const ShardPart* p = parts + (int64_t)s * ldp + m;
      z1 = p->Z; h1 = p->hl; n1 = p->n;
    }
    const bool ov = n1 > SHARD_CAP;
    over |= __ballot(ov) != 0ull;
    // exclusive prefix of the candidate counts: where this shard's candidates go in the merged list
    int inc = ov ? 0 : n1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    const int base = c + inc - (ov ? 0 : n1);
    if (s < S && !ov && base + n1 <= 64) {
      const ShardPart* p = parts + (int64_t)s * ldp + m;
      for (int i = 0; i < n1; ++i) {
        cand_s[w][base + i] = p->key[i];
        cfx_s[w][base + i] = p->fx[i];
        cidx_s[w][base + i] = p->idx[i];
      }
    }
    c += __shfl(inc, 63, 64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { z1 += __shfl_xor(z1, o, 64); h1 += __shfl_xor(h1, o, 64); }
    Z += z1;
    hl += h1;
  }
  const unsigned long long T = (unsigned long long)((double)top_p * (double)Z);  // floor, as nucleus_crossing
  if (!nucleus || over || c > 64 || !(hl > T)) {
    if (lane == 0) flags[m] = 1;
    return;
  }
  if (lane == 0) flags[m] = 0;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the list was written by other lanes of this wave
  list_tail_wave(cand_s[w], cfx_s[w], cidx_s[w], c, T, mx_all[m], V, temperature, conf_mode,
                 noise ? noise + (int64_t)m * ldn : nullptr, pn, (unsigned long long)m * V, lane, conf + m, x0 + m);
}

int launch_shard_row_max(const float* logits, int64_t ldl, int M, int Vs, float temperature, float* mx,
                         hipStream_t st) {
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(logits && mx && Vs >= 1 && ldl >= Vs, "shard_row_max: bad argument");
  hipLaunchKernelGGL(shard_row_max_kernel, dim3(M), dim3(256), 0, st, logits, ldl, Vs, temperature, mx);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int launch_shard_partials(const float* logits, int64_t ldl, int M, int Vs, int col0, float temperature,
                          const float* mx, void* parts, hipStream_t st) {
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(logits && mx && parts && Vs >= 1 && ldl >= Vs && col0 >= 0, "shard_partials: bad argument");
  hipLaunchKernelGGL(shard_partials_kernel, dim3(M), dim3(256), 0, st, logits, ldl, Vs, col0, temperature, mx,
                     (ShardPart*)parts);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int launch_sample_from_shards(const void* parts, int64_t ldp, int S, int M, int V, float temperature, float top_p, int top_k,
                              int conf_mode, const float* mx, const float* noise, int64_t ldn, uint64_t seed,
                              uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0, uint8_t* flags,
                              hipStream_t st) {
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(parts && mx && conf && x0 && flags && S >= 1 && S <= 4096 && ldp >= M, "sample_from_shards: bad argument");
  DARK_CHECK_ARG(V >= 2 && conf_mode >= 0 && conf_mode <= 2 && temperature >= 0.f, "sample_from_shards: bad argument");
  DARK_CHECK_ARG(!(temperature > 0.f) || nthreads > 0 || (noise && ldn >= V),
                 "sample_from_shards: temperature > 0 needs noise [M,V] or a Philox state");
  const PhiloxNoise pn{seed, offset, nthreads};
  hipLaunchKernelGGL(sample_from_shards_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, (const ShardPart*)parts, ldp, S, M, V,
                     temperature, top_p, top_k, conf_mode, mx, noise, ldn, pn, conf, x0, flags);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

// ================================================================== fused LM head: finishing the rows from slice records
// gemm_bf16_s256<EPI_HEAD> leaves, per masked row and 64-column slice of the vocabulary, {max logit, its column, how
// many columns lie within the head range of it, sum over the other columns of exp((l - max) / T)} instead of logits
// (16 B per slice: 3.8 KB per row at V = 15080 against 60 KB of fp32 logits).  One wave per row turns them into what
// sample_kernel's head path computes, bit for bit:
//   * row maximum = max of the slice maxima; the head set (logit / T within HEAD_RANGE of it) can only contain the
//     maxima of "hot" slices (max within the raw-space range of the row maximum) and, where a hot slice counted more
//     than one column in range, other columns of that slice: those 64 logits are RECOMPUTED here with the GEMM's own
//     MFMA chain (same operands, same K order: the same bits) and examined one by one;
//   * head candidates get their exact 2^-40 fixed-point masses; every other column's mass is only needed inside the
//     total Z that sets the nucleus threshold T = floor(top_p * Z): slice maxima outside the head set are added
//     exactly, the rest is bounded from the slice sums, Z in [Z_lo, Z_hi], with margins that cover the float
//     evaluation (relative 2e-3 + the rounding of l / T), the summation order (2^-17 per slice) and the truncation of
//     the per-element masses (one unit each);
//   * if T(Z_lo) and T(Z_hi) give the same crossing element and the same number of kept ties, every Z in between --
//     the exact one included -- gives the kept set the unfused sampler finds; list_tail_wave then does softmax,
//     confidence and the Philox draw on that set exactly as there.  Otherwise (and for more than 64 candidates, more
//     than 8 slices to recompute, a head set that does not hold top_p of the mass) the row is appended to a list; the
//     caller runs the plain head GEMM and sample_kernel for the listed rows (dark_head_sample_rows).
namespace {
constexpr int HF_MAXRE = 8;  // hot slices with more than one column in range that a row may recompute
struct HeadFinishLds {
  unsigned long long cfx[64];
  uint32_t cand[64];
  int cidx[64];
  int re[HF_MAXRE];
};
}  // namespace

__global__ __launch_bounds__(256) void head_finish_kernel(const HeadSliceRec* __restrict__ rec, int P, int M, int V, int K,
                                                          const bf16_t* __restrict__ xn, const bf16_t* __restrict__ W,
                                                          float temperature, float top_p, int conf_mode, float c2,
                                                          float range_raw, const float* __restrict__ noise, int64_t ldn,
                                                          PhiloxNoise pn, float* __restrict__ conf,
                                                          int64_t* __restrict__ x0, int32_t* __restrict__ list,
                                                          int32_t* __restrict__ counters) {
  __shared__ HeadFinishLds lds_all[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + w;
  if (m >= M) return;  // wave-uniform
  HeadFinishLds& L = lds_all[w];
  const HeadSliceRec* r = rec + (int64_t)m * P;
  // ---- pass 1: row maximum (raw logits)
  float lmax = -INFINITY;
  for (int p = lane; p < P; p += 64) lmax = fmaxf(lmax, r[p].smax);
  lmax = wave_max(lmax);
  const float mx = lmax / temperature;  // the sampler's scaled maximum: the same division on the same value
  const float hot_thr = lmax - (range_raw + 2e-6f * fabsf(lmax));
  // ---- pass 2: classify the slices
  int c = 0, nre = 0;            // head candidates, slices to recompute (wave-uniform)
  bool fail = !(lmax > -INFINITY) || !(lmax < INFINITY);
  unsigned long long tail_exact = 0;  // masses known exactly that are not head candidates (per lane, summed later)
  double tail_f = 0.0, slack = 0.0;   // bounded part, in units of the maximum's mass
  for (int p0 = 0; p0 < P; p0 += 64) {
    const int p = p0 + lane;
    float smax = -INFINITY, tsum = 0.f;
    int nn = 0, col = 0;
    if (p < P) { const HeadSliceRec q = r[p]; smax = q.smax; tsum = q.tsum; nn = q.nn_col & 0xffff; col = q.nn_col >> 16; }
    const bool live = smax > -INFINITY;
    const bool hot = live && smax > hot_thr;
    const bool redo = hot && nn >= 2;
    if (live && !redo) {
      const float wgt = __builtin_amdgcn_exp2f((smax - lmax) * c2);
      tail_f += (double)wgt * (double)tsum;
      slack += (double)wgt * (double)(tsum + 1.0f) * (1.0 / 131072.0);
    }
    // this slice's maximum: head candidate or exact tail mass (a slice that is recomputed handles its own maximum)
    const float sc = smax / temperature;
    const bool is_head = live && !redo && (sc - mx > -HEAD_RANGE);
    if (live && !redo && !is_head) tail_exact += mass_fx(sc, mx);
    const unsigned long long hb = __ballot(is_head), rb = __ballot(redo);
    const int hpos = c + __popcll(hb & ((1ull << lane) - 1ull));
    if (is_head && hpos < 64) { L.cand[hpos] = fkey(sc); L.cfx[hpos] = mass_fx(sc, mx); L.cidx[hpos] = p * 64 + col; }
    c += __popcll(hb);
    const int rpos = nre + __popcll(rb & ((1ull << lane) - 1ull));
    if (redo && rpos < HF_MAXRE) L.re[rpos] = p;
    nre += __popcll(rb);
  }
  if (c > 64 || nre > HF_MAXRE) fail = true;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // ---- recompute the marked slices: logits[m][64 p .. 64 p + 63] = xn[m] . W[col] with the GEMM's MFMA chain
  if (!fail && nre > 0) {
    const int frow = lane & 15, fq = lane >> 4;
    const bf16_t* arow = xn + (int64_t)m * K + fq * 8;  // every MFMA row is this row: all 16 result rows are equal
    for (int i = 0; i < nre; ++i) {
      const int p = L.re[i];
      f32x4 acc[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const bf16_t* wrow = W + (int64_t)(p * 64 + frow) * K + fq * 8;
      // K-steps in groups of four: the 20 fragment loads of a group are in flight together (as a loop over single
      // K-steps every MFMA waited out a round trip to L2: 24 dependent latencies a slice, 84 us a row); the MFMA chain of
      // each accumulator still runs over ascending k, so the bits are the GEMM's
      constexpr int HU = 4;
      const int KT = K / 32;
      for (int kt0 = 0; kt0 < KT; kt0 += HU) {
        bf16x8 fa[HU], fw[HU][4];
#pragma unroll
        for (int u = 0; u < HU; ++u) {
          const int kt = min(kt0 + u, KT - 1);   // (a clamped duplicate load instead of a branch; not accumulated)
          fa[u] = *(const bf16x8*)(arow + kt * 32);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) fw[u][ni] = *(const bf16x8*)(wrow + (int64_t)ni * 16 * K + kt * 32);
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
          if (kt0 + u < KT) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[u][ni], fa[u], acc[ni], 0, 0, 0);
          }
        }
      }
      // lane (frow, fq) holds columns 64 p + 16 ni + 4 fq + e of row `frow` of the tile (= this row); lanes with
      // frow == 0 speak for the slice
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int colg = p * 64 + (k >> 2) * 16 + 4 * fq + (k & 3);
        const float sc = acc[k >> 2][k & 3] / temperature;
        const bool val = frow == 0 && colg < V;
        const bool is_head = val && (sc - mx > -HEAD_RANGE);
        if (val && !is_head) tail_exact += mass_fx(sc, mx);
        const unsigned long long hb = __ballot(is_head);
        const int hpos = c + __popcll(hb & ((1ull << lane) - 1ull));
        if (is_head && hpos < 64) { L.cand[hpos] = fkey(sc); L.cfx[hpos] = mass_fx(sc, mx); L.cidx[hpos] = colg; }
        c += __popcll(hb);
      }
    }
    if (c > 64) fail = true;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  // ---- Z interval, threshold interval, certification of the crossing
  unsigned long long T_hi = 0;
  if (!fail) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      tail_exact += __shfl_xor(tail_exact, o, 64);
      tail_f += __shfl_xor(tail_f, o, 64);
      slack += __shfl_xor(slack, o, 64);
    }
    unsigned long long hl = lane < c ? L.cfx[lane] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hl += __shfl_xor(hl, o, 64);
    const double delta = 2e-3 + 4e-7 * ((double)fabsf(mx) + 30.0);
    const double two40 = 1099511627776.0;
    const double known = (double)hl + (double)tail_exact;  // exact integers below 2^53
    double z_lo = known + (tail_f * (1.0 - delta) - slack) * two40 - (double)V;
    const double z_hi = known + (tail_f * (1.0 + delta) + slack) * two40 + 1.0;
    if (z_lo < known) z_lo = known;
    const unsigned long long T_lo = (unsigned long long)((double)top_p * z_lo);
    T_hi = (unsigned long long)((double)top_p * z_hi);
    if (!(hl > T_hi) || c < 1) fail = true;
    if (!fail) {  // tau / kept ties under both thresholds (list_tail_wave's own arithmetic)
      const bool own = lane < c;
      const uint32_t kj = own ? L.cand[lane] : 0u;
      unsigned long long S = 0;
      for (int i = 0; i < c; ++i) S += L.cand[i] >= kj ? L.cfx[i] : 0ull;
      uint32_t tau_lo = (own && S > T_lo) ? kj : 0u, tau_hi = (own && S > T_hi) ? kj : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        tau_lo = max(tau_lo, (uint32_t)__shfl_xor((int)tau_lo, o, 64));
        tau_hi = max(tau_hi, (uint32_t)__shfl_xor((int)tau_hi, o, 64));
      }
      if (tau_lo != tau_hi) fail = true;
      else {
        unsigned long long above = (own && kj > tau_hi) ? L.cfx[lane] : 0ull, nt = (own && kj == tau_hi) ? 1ull : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { above += __shfl_xor(above, o, 64); nt += __shfl_xor(nt, o, 64); }
        if (nt > 1ull) {
          const unsigned long long fx = mass_fx(fkey_inv(tau_hi), mx);
          if (fx && (T_lo - above) / fx != (T_hi - above) / fx) fail = true;
        }
      }
    }
  }
  if (fail) {
    if (lane == 0) list[atomicAdd(&counters[0], 1)] = m;
    return;
  }
  list_tail_wave(L.cand, L.cfx, L.cidx, c, T_hi, mx, V, temperature, conf_mode, noise ? noise + (int64_t)m * ldn : nullptr, pn,
                 (unsigned long long)m * V, lane, conf + m, x0 + m);
}

// rows of xn listed in `list` (count read from device memory) -> out, 16 bytes a lane
__global__ __launch_bounds__(256) void gather_listed_rows_kernel(const bf16_t* __restrict__ xn, const int32_t* __restrict__ list,
                                                                 const int32_t* __restrict__ n_dev, bf16_t* __restrict__ out,
                                                                 int d, int M) {
  const int n = min(*n_dev, M);
  const int per_row = d / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)n * per_row; i += (int64_t)gridDim.x * 256) {
    const int rrow = (int)(i / per_row), ch = (int)(i - (int64_t)rrow * per_row);
    int src = list[rrow];
    src = src < 0 ? 0 : (src >= M ? M - 1 : src);
    *(uint4*)(out + (int64_t)rrow * d + ch * 8) = *(const uint4*)(xn + (int64_t)src * d + ch * 8);
  }
}

int launch_head_finish(const void* records, int P, int M, int V, int K, const void* xn, const void* W, float temperature,
                       float top_p, int conf_mode, const float* noise, int64_t ldn, uint64_t seed, uint64_t offset,
                       uint32_t nthreads, float* conf, int64_t* x0, int32_t* list, int32_t* counters, hipStream_t st) {
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(records && xn && W && conf && x0 && list && counters, "head_finish: null pointer");
  DARK_CHECK_ARG(temperature > 0.f && top_p >= 0.f && top_p < 1.f && conf_mode >= 0 && conf_mode <= 2 && K % 32 == 0,
                 "head_finish: needs temperature > 0 and an active nucleus filter");
  DARK_CHECK_ARG(nthreads > 0 || (noise && ldn >= V), "head_finish: needs noise [M,V] or a Philox state");
  const PhiloxNoise pn{seed, offset, nthreads};
  hipLaunchKernelGGL(head_finish_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, (const HeadSliceRec*)records, P, M, V, K,
                     (const bf16_t*)xn, (const bf16_t*)W, temperature, top_p, conf_mode, head_c2(temperature),
                     head_range_raw(temperature), noise, ldn, pn, conf, x0, list, counters);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int launch_gather_listed_rows(const void* xn, const int32_t* list, const int32_t* n_dev, void* out, int d, int M,
                              hipStream_t st) {
  DARK_CHECK_ARG(xn && list && n_dev && out && d % 8 == 0 && M > 0, "gather_listed_rows: bad argument");
  hipLaunchKernelGGL(gather_listed_rows_kernel, dim3(512), dim3(256), 0, st, (const bf16_t*)xn, list, n_dev, (bf16_t*)out, d, M);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

}  // namespace dark
