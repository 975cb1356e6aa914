// Shared device/host helpers for the DARK denoise hot path on gfx950 (MI355X, CDNA4).
// Wave = 64 lanes everywhere in this directory.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dark_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DARK_WAVE 64

namespace dark {

void set_error(const char* fmt, ...);

#define DARK_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      ::dark::set_error(__VA_ARGS__);             \
      return DARK_ERR_ARG;                        \
    }                                             \
  } while (0)

#define DARK_CHECK_HIP(expr)                                                        \
  do {                                                                              \
    hipError_t e__ = (expr);                                                        \
    if (e__ != hipSuccess) {                                                        \
      ::dark::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                        __FILE__, __LINE__);                                        \
      return DARK_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

#define DARK_CHECK_LAUNCH()                                                         \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      ::dark::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                        \
      return DARK_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

#define DARK_TRY(expr)            \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != DARK_OK) return rc__; \
  } while (0)

// Opt-in to more than 64 KiB of dynamic LDS.  hipFuncSetAttribute applies to the kernel ON THE CURRENT DEVICE, so
// the "already done" cache is one word per device (atomic: entry points may be called from several host threads).
// The same call refuses a stream that belongs to another device than the current one: the launch would otherwise
// run there without the attribute (and without this library's per-device state).
#define DARK_MAX_DEVICES 64
struct SmemOptIn {
  std::atomic<int> have[DARK_MAX_DEVICES] = {};
  int ensure(const void* fn, int bytes, hipStream_t st) {
    int dev = 0;
    DARK_CHECK_HIP(hipGetDevice(&dev));
    DARK_CHECK_ARG(dev >= 0 && dev < DARK_MAX_DEVICES, "device ordinal %d out of range", dev);
    if (st != nullptr) {
      hipDevice_t sdev = dev;
      DARK_CHECK_HIP(hipStreamGetDevice(st, &sdev));
      DARK_CHECK_ARG((int)sdev == dev, "stream belongs to device %d but device %d is current (make the model's "
                     "device current before calling)", (int)sdev, dev);
    }
    if (have[dev].load(std::memory_order_acquire) >= bytes) return DARK_OK;
    DARK_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    int cur = have[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !have[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
    return DARK_OK;
  }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Total order of confidences as torch.topk / torch.sort see them: -inf < finite < +inf < NaN (a NaN counts as the
// LARGEST value, so a row with NaN confidences still transfers exactly k positions); -0.0 == +0.0 as in a float
// compare.  Ranking on these keys keeps "ties -> lower index" well defined for every input.
__device__ __forceinline__ uint32_t order_key(float f) {
  if (f != f) return 0xffffffffu;
  const uint32_t u = __float_as_uint(f + 0.0f);  // -0.0 + 0.0 = +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kDtype = DARK_F32;
  __device__ static __forceinline__ float to_f32(float v) { return v; }
  __device__ static __forceinline__ float from_f32(float v) { return v; }
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int kDtype = DARK_BF16;
  __device__ static __forceinline__ float to_f32(bf16_t v) { return (float)v; }
  __device__ static __forceinline__ bf16_t from_f32(float v) { return (bf16_t)v; }
};

// The residual stream is fp32 or, for bf16 models with dark_model_desc.resid_dtype = DARK_BF16, bf16 (what the
// reference's bf16 model carries: `hidden_states = residual + hidden_states` on bf16 tensors,
// modeling_dream_length.py:508, :514).  TH = its element type; four consecutive elements move as one access.
template <typename TH>
__device__ __forceinline__ float4 load4(const TH* p) {
  if constexpr (sizeof(TH) == 4) {
    return *(const float4*)p;
  } else {
    const bf16x4 b = *(const bf16x4*)p;
    return make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]);
  }
}
template <typename TH>
__device__ __forceinline__ void store4(TH* p, const float4 v) {
  if constexpr (sizeof(TH) == 4) {
    *(float4*)p = v;
  } else {
    bf16x4 b;
    b[0] = (bf16_t)v.x; b[1] = (bf16_t)v.y; b[2] = (bf16_t)v.z; b[3] = (bf16_t)v.w;
    *(bf16x4*)p = b;
  }
}

// XCD-aware, bijective remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give every XCD one contiguous chunk of the tile space for L2 reuse.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---- launchers implemented across the .hip files --------------------------
// GEMM epilogues: C[M,N] = A[M,K] * W[N,K]^T (+...)
enum Epi { EPI_BIAS = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_F32 = 3, EPI_LSE = 4, EPI_HEAD = 5, EPI_BIAS_GELU = 6,
           EPI_RESID_BIAS = 7, EPI_RESID_H = 8, EPI_RESID_BIAS_H = 9 };

// A: [M,K] dtype, W: [Npad,K] dtype (rows padded to a multiple of 128 with zeros).
// EPI_BIAS  : out (dtype) [M,ldo] = acc + bias[n]
// EPI_RESID : resid (fp32) [M,ldo] += acc
// EPI_SWIGLU: W rows interleaved in groups of 32 (gate|up); out (dtype) [M,ldo], ldo cols = N/2
// EPI_F32   : out (fp32) [M,ldo] = acc, columns n < N only
// EPI_BIAS_GELU : out (dtype) [M,ldo] = gelu_new(acc + bias[n])            (GPT-2 backbone: mlp.c_fc + act)
// EPI_RESID_BIAS: resid (fp32) [M,ldo] += acc + bias[n]                     (GPT-2 backbone: attn / mlp c_proj)
// EPI_RESID_H / EPI_RESID_BIAS_H: the two residual flavours on a bf16 residual stream (bf16 only):
//             resid (bf16) [M,ldo] = bf16(float(resid) + acc (+ bias[n])), one rounding
// EPI_LSE   : bf16 only; bias = int64 targets [M]; out = float2 part[M][ldo] (max, sum exp(acc - max)) per
//             64-column wave slice, ldo = 2*ceil(N/128), followed by float tgt[M] = acc[m][target[m]]
int launch_gemm(int dtype, int epi, const void* A, const void* W, const float* bias, void* out,
                int M, int N, int K, int64_t ldo, hipStream_t st);

// gemm256.hip: persistent 256 x 256 bf16 kernel (EPI_BIAS / EPI_SWIGLU / EPI_F32); w_rows = allocated rows of W
bool p256_supports(int epi, int M, int N, int K, int64_t ldo, const void* out, const void* bias);
int launch_gemm_p256(int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                     int64_t ldo, int w_rows, hipStream_t st);
bool s256_supports(int epi, int M, int N, int K, int64_t ldo, const void* out, const void* bias);
int launch_gemm_s256(int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                     int64_t ldo, int w_rows, hipStream_t st);
// EPI_HEAD (gemm_bf16_s256 only): no logits are written; per (row, 64-column slice) one 16-byte record
// {max logit, sum over the slice's OTHER columns of exp2((l - max) * c2), count of columns within `range` of the max
// | column of the max << 16, 0} into records[M][P], P = 4 * ceil(N / 256).  m_dev (nullable): the row count is read
// from device memory (min(M, *m_dev)); launched for at most M rows.
struct HeadSliceRec { float smax, tsum; int nn_col, pad; };
// temperature -> the two epilogue constants: log2(e) / T, and the raw-logit width that is sure to cover the sampler's
// head range (6.0 in units of logit / T) with every rounding of l / T on the safe side
static inline float head_c2(float T) { return 1.4426950408889634f / T; }
static inline float head_range_raw(float T) { return 6.0f * T * 1.001f; }
bool head_gemm_supports(int M, int N, int K);
int launch_gemm_head(const void* A, const void* W, void* records, int M, int N, int K, float c2, float range,
                     const int* m_dev, int w_rows, hipStream_t st);
// the EPI_F32 flavour of the same kernel with the row count read from device memory (fallback rows of the fused head)
int launch_gemm_s256_devrows(const void* A, const void* W, void* out, int M, int N, int K, int64_t ldo, const int* m_dev,
                             int w_rows, hipStream_t st);
void set_gemm_variant(int v);
int gemm_variant();
// gemm_rowpanel.hip: hidden[M,768] += A[M,K] W[768,K]^T (fp32 residual, in place) and, with ln_w / xn non-null,
// xn = bf16(ln_w * hidden / rms(hidden)) from the same registers (bf16 operands, N = 768 only)
bool rowpanel_supports(int M, int N, int K, const void* A, const void* W, const void* hidden, const void* ln_w,
                       const void* xn);
int launch_gemm_resid_norm(const void* A, const void* W, float* hidden, const float* ln_w, void* xn, int M, int K,
                           float eps, hipStream_t st);
int launch_gemm_resid_norm_h(const void* A, const void* W, void* hidden, const float* ln_w, void* xn, int M, int K,
                             float eps, hipStream_t st);   // the same on a bf16 residual stream

int prof_select(int epi, int rows = 0);
int prof_read(double* total_ms, double* total_flops, int* launches);

// `xdt` below = element type of the residual stream: DARK_F32, or DARK_BF16 (bf16 models only)
int launch_embed(int dtype, int xdt, const int64_t* ids, const void* table, void* out, int M, int d, int vocab,
                 hipStream_t st);
// out[m] = w * (x[src(m)] * rsqrt(mean(x^2) + eps)); src_rows nullable (identity), else clamped to [0,n_src)
int launch_embed_norm(int dtype, int xdt, const int64_t* ids, const void* table, void* out, int M, int d, int vocab,
                      const float* ln_w, float eps, void* xn, hipStream_t st);
int launch_rmsnorm(int dtype, int xdt, const void* x, const float* w, void* out, const int32_t* src_rows, int n_src,
                   int M, int d, float eps, hipStream_t st);
// att_out[r] = att[rows[r]] (model dtype; both nullable together) and x_out[r] = x[rows[r]] (stream dtype), rows
// clamped to [0, n_src)
int launch_gather_rows(int dtype, int xdt, const void* att, void* att_out, const void* x, void* x_out, const int32_t* rows,
                       int n_src, int R, int d, hipStream_t st);
int launch_attention(int dtype, const void* qkv, const float* cosT, const float* sinT, const int32_t* pos,
                     const uint8_t* kmask, void* out, int B, int N, int H, int HD, int max_pos,
                     hipStream_t st, bool causal = false);
// gpt2.hip: torch.nn.LayerNorm over gathered rows (src_rows nullable = identity, clamped to [0, n_src))
int launch_layernorm(int dtype, int xdt, const void* x, const float* w, const float* b, void* out, const int32_t* src_rows,
                     int n_src, int M, int d, float eps, hipStream_t st);
int launch_mask_rows(const int64_t* x, int B, int N, int64_t mask_id, int32_t* row_index,
                     int32_t* src_rows, int32_t* count, hipStream_t st);
int launch_sample(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p,
                  int top_k, int conf_mode, const float* noise, int64_t ldn, float* conf, int64_t* x0,
                  hipStream_t st);
// noise == NULL and nthreads > 0: the Exp(1) draws are computed in place from torch's Philox stream (sample.hip)
int launch_sample_philox(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p, int top_k,
                         int conf_mode, const float* noise, int64_t ldn, uint64_t seed, uint64_t offset,
                         uint32_t nthreads, float* conf, int64_t* x0, hipStream_t st);
// `row_ids` nullable: logits is a compact list of rows row_ids[i] of the [M, V] problem (noise / Philox index / outputs)
int launch_sample_indexed(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p, int top_k,
                          int conf_mode, const int32_t* row_ids, const float* noise, int64_t ldn, uint64_t seed,
                          uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0, hipStream_t st,
                          const int32_t* n_dev = nullptr);
// fused head (sample.hip): finish the rows from the slice records of launch_gemm_head; rows that cannot be finished are
// appended to list[counters[0]++]
int launch_head_finish(const void* records, int P, int M, int V, int K, const void* xn, const void* W, float temperature,
                       float top_p, int conf_mode, const float* noise, int64_t ldn, uint64_t seed, uint64_t offset,
                       uint32_t nthreads, float* conf, int64_t* x0, int32_t* list, int32_t* counters, hipStream_t st);
int launch_gather_listed_rows(const void* xn, const int32_t* list, const int32_t* n_dev, void* out, int d, int M,
                              hipStream_t st);
// vocabulary-sharded head (sample.hip)
int launch_shard_row_max(const float* logits, int64_t ldl, int M, int Vs, float temperature, float* mx, hipStream_t st);
int launch_shard_partials(const float* logits, int64_t ldl, int M, int Vs, int col0, float temperature,
                          const float* mx, void* parts, hipStream_t st);
int set_sampler_row_workspace(void* ptr, size_t bytes);
void set_sampler_stages(int v);
int launch_sample_from_shards(const void* parts, int64_t ldp, int S, int M, int V, float temperature, float top_p, int top_k,
                              int conf_mode, const float* mx, const float* noise, int64_t ldn, uint64_t seed,
                              uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0, uint8_t* flags,
                              hipStream_t st);
int launch_philox_exponential(uint64_t seed, uint64_t offset, uint32_t nthreads, uint64_t numel, float* out,
                              hipStream_t st);
// alg_temp > 0: multinomial ranking with the [B,N] Exp(1) draws `qnoise` (generation_utils.py:527-530)
int launch_commit(int64_t* x, int B, int N, int64_t mask_id, const int32_t* row_index, int M,
                  const float* conf, const int64_t* x0, int k, hipStream_t st, float alg_temp = 0.f,
                  const float* qnoise = nullptr);
int launch_scatter_tokens(int64_t* x, int64_t total, const int32_t* row_index, int M, const int64_t* sel,
                          const int64_t* tokens, int n, hipStream_t st);

}  // namespace dark
