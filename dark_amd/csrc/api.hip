// extern "C" entry points of libdark_hip.so (declared in include/dark_hip.h) and the per-step
// launch sequence.  Nothing here allocates or synchronises; every call only enqueues on `stream`.
#include <stdarg.h>

#include "common.h"

namespace dark {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static size_t esize(int dtype) { return dtype == DARK_BF16 ? 2 : 4; }

static int check_model(const dark_model_desc* m) {
  DARK_CHECK_ARG(m != nullptr, "null model descriptor");
  DARK_CHECK_ARG(m->dtype == DARK_F32 || m->dtype == DARK_BF16, "bad dtype %d", m->dtype);
  DARK_CHECK_ARG(m->d > 0 && m->d % 64 == 0, "hidden_size=%d must be a positive multiple of 64", m->d);
  DARK_CHECK_ARG(m->ff > 0 && m->ff % 64 == 0, "intermediate_size=%d must be a positive multiple of 64", m->ff);
  DARK_CHECK_ARG(m->n_heads > 0 && m->n_heads * m->head_dim == m->d, "heads %d x head_dim %d != hidden %d",
                 m->n_heads, m->head_dim, m->d);
  DARK_CHECK_ARG(m->head_dim == 64 || m->head_dim == 96, "head_dim=%d must be 64 or 96", m->head_dim);
  DARK_CHECK_ARG(m->vocab >= 2 && m->vocab_pad >= m->vocab && m->vocab_pad % 128 == 0,
                 "vocab=%d vocab_pad=%d (pad must be a multiple of 128)", m->vocab, m->vocab_pad);
  DARK_CHECK_ARG(m->n_layers >= 0 && (m->n_layers == 0 || m->layers), "layers missing");
  DARK_CHECK_ARG(m->embed && m->final_norm && m->lm_head && m->rope_cos && m->rope_sin, "null weight pointer");
  DARK_CHECK_ARG(m->max_pos >= 1, "max_pos=%d", m->max_pos);
  DARK_CHECK_ARG(m->resid_dtype == DARK_F32 || (m->resid_dtype == DARK_BF16 && m->dtype == DARK_BF16),
                 "resid_dtype %d with model dtype %d (fp32 always, bf16 for bf16 models)", m->resid_dtype, m->dtype);
  return DARK_OK;
}

struct FwdWs {
  char *xn, *qkv, *att, *act;
  size_t total;
};
static FwdWs carve_fwd(const dark_model_desc* m, size_t rows, char* base) {
  const size_t es = esize(m->dtype);
  FwdWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };  // (null base: sizes only)
  w.xn = take(rows * m->d * es);
  w.qkv = take(rows * 3 * m->d * es);
  w.att = take(rows * m->d * es);
  w.act = take(rows * m->ff * es);
  w.total = off;
  return w;
}

// hidden += A * W^T (o_proj / down_proj), then -- when `ln` is given -- xn = RMSNorm(hidden) * ln for the next
// projection: the residual GEMM and the norm as two launches (the GEMM flavour follows the residual stream's dtype).
// Variant 5 (dark_dbg_set_gemm_variant) routes bf16 models of width 768 WITH AN FP32 STREAM through the row-panel kernel that does both from one set of registers (gemm_rowpanel.hip);
// it is NOT the default: measured in the denoise loop it is 6 % slower end to end (2675 vs 2835 sequences/s,
// profiles/r02q_rowpanel_ab.txt) -- its 246 workgroups run in lockstep, so the 180 MB residual read-modify-write of all
// epilogues hits HBM at once with no MFMA work left to hide it, where the multi-round 128 x 192 kernel overlaps
// one tile's epilogue with its neighbours' main loops.
static int resid_gemm_norm(const dark_model_desc* m, const void* A, const void* W, void* hidden, const float* ln,
                           void* xn, int M, int K, hipStream_t st) {
  const int d = m->d;
  if (m->dtype == DARK_BF16 && m->resid_dtype == DARK_F32 && gemm_variant() == 5 &&
      rowpanel_supports(M, d, K, A, W, hidden, ln, ln ? xn : nullptr))
    return launch_gemm_resid_norm(A, W, (float*)hidden, ln, ln ? xn : nullptr, M, K, m->rms_eps, st);
  // bf16 stream (round 3): the row-panel kernel gives the same bits as the two launches below (hidden AND xn), so it
  // can be chosen per call: variant 5 = for both residual projections, 13 = for o_proj only (K = d)
  if (m->dtype == DARK_BF16 && m->resid_dtype == DARK_BF16 && (gemm_variant() == 5 || (gemm_variant() == 13 && K == d)) &&
      rowpanel_supports(M, d, K, A, W, hidden, ln, ln ? xn : nullptr))
    return launch_gemm_resid_norm_h(A, W, hidden, ln, ln ? xn : nullptr, M, K, m->rms_eps, st);
  const int epi = m->resid_dtype == DARK_BF16 ? EPI_RESID_H : EPI_RESID;
  DARK_TRY(launch_gemm(m->dtype, epi, A, W, nullptr, hidden, M, d, K, d, st));
  if (ln) DARK_TRY(launch_rmsnorm(m->dtype, m->resid_dtype, hidden, ln, xn, nullptr, M, M, d, m->rms_eps, st));
  return DARK_OK;
}

// `tail_rows` = false: the whole body.  true: stop after the LAST layer's attention -- what is left of that layer
// (o_proj, residual, norm, MLP, residual) only matters for the rows the head will read and is done by forward_tail.
static int forward_hidden(const dark_model_desc* m, const int64_t* ids, const int32_t* pos,
                          const uint8_t* key_mask, int B, int N, void* workspace, size_t ws_bytes,
                          void* hidden, hipStream_t st, bool tail_rows = false) {
  DARK_TRY(check_model(m));
  DARK_CHECK_ARG(B > 0 && N > 0 && N <= 128, "forward: B=%d N=%d (N must be in [1,128])", B, N);
  DARK_CHECK_ARG((int64_t)B * N < (1ll << 30), "forward: B*N too large");
  DARK_CHECK_ARG(ids && hidden && workspace, "forward: null pointer");
  DARK_CHECK_ARG(pos != nullptr || N <= m->max_pos, "forward: N=%d exceeds the RoPE table (%d rows)", N, m->max_pos);
  const int M = B * N, d = m->d;
  FwdWs w = carve_fwd(m, (size_t)M, (char*)workspace);
  DARK_CHECK_ARG(ws_bytes >= w.total, "forward: workspace %zu < required %zu bytes", ws_bytes, w.total);

  if (m->n_layers > 0) {  // embedding + the first layer's input norm in one pass over the gathered rows
    DARK_CHECK_ARG(m->layers[0].ln1, "layer 0: null weight");
    DARK_TRY(launch_embed_norm(m->dtype, m->resid_dtype, ids, m->embed, hidden, M, d, m->vocab, m->layers[0].ln1, m->rms_eps, w.xn, st));
  } else {
    DARK_TRY(launch_embed(m->dtype, m->resid_dtype, ids, m->embed, hidden, M, d, m->vocab, st));
  }
  for (int l = 0; l < m->n_layers; ++l) {
    const dark_layer_weights& L = m->layers[l];
    DARK_CHECK_ARG(L.w_qkv && L.b_qkv && L.w_o && L.w_gu && L.w_down && L.ln1 && L.ln2, "layer %d: null weight", l);
    // (w.xn = this layer's normalised input: from embed_norm for layer 0, from the previous down_proj's epilogue after)
    DARK_TRY(launch_gemm(m->dtype, EPI_BIAS, w.xn, L.w_qkv, L.b_qkv, w.qkv, M, 3 * d, d, 3 * d, st));
    DARK_TRY(launch_attention(m->dtype, w.qkv, m->rope_cos, m->rope_sin, pos, key_mask, w.att, B, N, m->n_heads,
                              m->head_dim, m->max_pos, st));
    if (tail_rows && l == m->n_layers - 1) break;
    DARK_TRY(resid_gemm_norm(m, w.att, L.w_o, hidden, L.ln2, w.xn, M, d, st));
    DARK_TRY(launch_gemm(m->dtype, EPI_SWIGLU, w.xn, L.w_gu, nullptr, w.act, M, 2 * m->ff, d, m->ff, st));
    const float* ln_next = l + 1 < m->n_layers ? m->layers[l + 1].ln1 : nullptr;  // the final norm runs on gathered rows
    DARK_CHECK_ARG(l + 1 >= m->n_layers || ln_next, "layer %d: null weight", l + 1);
    DARK_TRY(resid_gemm_norm(m, w.act, L.w_down, hidden, ln_next, w.xn, M, m->ff, st));
  }
  return DARK_OK;
}

// The rest of the last layer for the R rows in `rows` only (bidirectional attention needs every position's K/V in
// every layer, but nothing reads the last layer's output except the LM head, which reads R << B*N rows):
// hidden_rows[r] = what the full body leaves in hidden[rows[r]], bit for bit -- every GEMM output element is the same
// K-ordered dot product whichever rows share its tile.  The compact operands reuse the (now dead) xn / qkv / act regions.
static int forward_tail(const dark_model_desc* m, int B, int N, const int32_t* rows, int R, void* workspace,
                        size_t ws_bytes, const void* hidden, void* hidden_rows, hipStream_t st) {
  DARK_TRY(check_model(m));
  DARK_CHECK_ARG(B > 0 && N > 0 && R >= 0 && R <= B * N, "forward_tail: bad B=%d N=%d rows=%d", B, N, R);
  if (R == 0) return DARK_OK;
  DARK_CHECK_ARG(rows && workspace && hidden && hidden_rows, "forward_tail: null pointer");
  const int d = m->d, n_src = B * N;
  FwdWs w = carve_fwd(m, (size_t)n_src, (char*)workspace);
  DARK_CHECK_ARG(ws_bytes >= w.total, "forward_tail: workspace %zu < required %zu bytes", ws_bytes, w.total);
  if (m->n_layers == 0) return launch_gather_rows(m->dtype, m->resid_dtype, nullptr, nullptr, hidden, hidden_rows, rows, n_src, R, d, st);
  const dark_layer_weights& L = m->layers[m->n_layers - 1];
  char *att_c = w.xn, *xn_c = w.qkv;
  DARK_TRY(launch_gather_rows(m->dtype, m->resid_dtype, w.att, att_c, hidden, hidden_rows, rows, n_src, R, d, st));
  DARK_TRY(resid_gemm_norm(m, att_c, L.w_o, hidden_rows, L.ln2, xn_c, R, d, st));
  DARK_TRY(launch_gemm(m->dtype, EPI_SWIGLU, xn_c, L.w_gu, nullptr, w.act, R, 2 * m->ff, d, m->ff, st));
  DARK_TRY(resid_gemm_norm(m, w.act, L.w_down, hidden_rows, nullptr, nullptr, R, m->ff, st));
  return DARK_OK;
}

static int lm_head_rows(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int n_src, int M,
                        void* workspace, size_t ws_bytes, float* logits, int64_t ldl, hipStream_t st) {
  DARK_TRY(check_model(m));
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(hidden && workspace && logits, "lm_head_rows: null pointer");
  DARK_CHECK_ARG(ldl >= m->vocab, "lm_head_rows: ldl=%lld < vocab=%d", (long long)ldl, m->vocab);
  const size_t need = align_up((size_t)M * m->d * esize(m->dtype), 256);
  DARK_CHECK_ARG(ws_bytes >= need, "lm_head_rows: workspace %zu < required %zu bytes", ws_bytes, need);
  DARK_TRY(launch_rmsnorm(m->dtype, m->resid_dtype, hidden, m->final_norm, workspace, src_rows, n_src, M, m->d, m->rms_eps, st));
  // A vocabulary that is not a multiple of four (DBpedia50: 25 427, WN18RR: 41 066) on a row stride that leaves room:
  // the GEMM runs over the next multiple of four -- the extra columns come from the zero rows that pad lm_head and land
  // in the row's own padding -- so that the persistent kernel's 16-byte logit stores apply (they need whole pieces).
  const int v4 = (m->vocab + 3) & ~3;
  const int n_gemm = (ldl >= v4 && m->vocab_pad >= v4) ? v4 : m->vocab;
  DARK_TRY(launch_gemm(m->dtype, EPI_F32, workspace, m->lm_head, nullptr, logits, M, n_gemm, m->d, ldl, st));
  return DARK_OK;
}

// ---- scratch of the allocation-free entry point (dark_lm_head_rows_noalloc): the gathered, normalised rows that
// dark_lm_head_rows keeps in its caller-provided workspace live here instead.  Every GPU the library runs on holds
// its own copy of the array until the process ends; the name below is only a host-side handle, the device address is
// resolved per device with hipGetSymbolAddress and that pointer is what the kernels receive.
constexpr size_t kNoAllocScratchBytes = DARK_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
static std::atomic<void*> g_scratch_ptr[DARK_MAX_DEVICES] = {};
static std::atomic<int> g_scratch_ready[DARK_MAX_DEVICES] = {};

// Device address of `scratch` on the CURRENT device (resolved once per device, with that device current).
static int scratch_on_current_device(void** out) {
  int dev = 0;
  DARK_CHECK_HIP(hipGetDevice(&dev));
  DARK_CHECK_ARG(dev >= 0 && dev < DARK_MAX_DEVICES, "device ordinal %d out of range", dev);
  if (!g_scratch_ready[dev].load(std::memory_order_acquire)) {
    void* p = nullptr;
    DARK_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    DARK_CHECK_ARG(p != nullptr, "hipGetSymbolAddress returned a null scratch address on device %d", dev);
    g_scratch_ptr[dev].store(p, std::memory_order_relaxed);  // (racing threads resolve the same address)
    g_scratch_ready[dev].store(1, std::memory_order_release);
  }
  *out = g_scratch_ptr[dev].load(std::memory_order_relaxed);
  return DARK_OK;
}

struct StepWs {
  char* fwd;
  char* hidden;
  int32_t *row_index, *src_rows, *count;
  char* head;
  float* logits;
  float* conf;
  int64_t* x0;
  char* hidden_rows;
  size_t off[7];   // byte offsets of {hidden, row_index, src_rows, count, logits, conf, x0} (dark_step_workspace_offsets)
  size_t fwd_bytes, head_bytes, total;
};
static StepWs carve_step(const dark_model_desc* m, int B, int N, int M, char* base) {
  StepWs w;
  size_t off = 0;
  // (offsets first, pointers from them: `base` may be null when only sizes / offsets are wanted)
  auto take_off = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  auto at = [&](size_t o) { return base ? base + o : nullptr; };
  const size_t rows = (size_t)B * N;
  w.fwd_bytes = carve_fwd(m, rows, nullptr).total;
  w.fwd = at(take_off(w.fwd_bytes));
  const size_t hs = esize(m->resid_dtype);   // the residual stream: fp32, or bf16 for bf16 models that ask for it
  w.off[0] = take_off(rows * m->d * hs);
  w.off[1] = take_off(rows * 4);
  w.off[2] = take_off(rows * 4);
  w.off[3] = take_off(16);
  w.head_bytes = align_up((size_t)M * m->d * esize(m->dtype), 256);
  w.head = at(take_off(w.head_bytes));
  w.off[4] = take_off((size_t)M * m->vocab * 4);
  w.off[5] = take_off((size_t)M * 4);
  w.off[6] = take_off((size_t)M * 8);
  w.hidden = at(w.off[0]);
  w.row_index = (int32_t*)at(w.off[1]);
  w.src_rows = (int32_t*)at(w.off[2]);
  w.count = (int32_t*)at(w.off[3]);
  w.logits = (float*)at(w.off[4]);
  w.conf = (float*)at(w.off[5]);
  w.x0 = (int64_t*)at(w.off[6]);
  w.hidden_rows = at(take_off((size_t)M * m->d * hs));
  w.total = off;
  return w;
}

}  // namespace dark

using namespace dark;

extern "C" {

int dark_hip_abi_version(void) { return DARK_HIP_ABI_VERSION; }
const char* dark_hip_last_error(void) { return g_err; }

size_t dark_forward_workspace_bytes(const dark_model_desc* m, int B, int N) {
  if (!m || B <= 0 || N <= 0) return 0;
  return carve_fwd(m, (size_t)B * N, nullptr).total;
}
size_t dark_head_workspace_bytes(const dark_model_desc* m, int M) {
  if (!m || M <= 0) return 0;
  return align_up((size_t)M * m->d * esize(m->dtype), 256);
}
size_t dark_step_workspace_bytes(const dark_model_desc* m, int B, int N, int M) {
  if (!m || B <= 0 || N <= 0 || M < 0) return 0;
  return carve_step(m, B, N, M, nullptr).total;
}

int dark_step_workspace_offsets(const dark_model_desc* m, int B, int N, int M, size_t* out7) {
  DARK_CHECK_ARG(m && out7 && B > 0 && N > 0 && M >= 0, "step_workspace_offsets: bad argument");
  StepWs w = carve_step(m, B, N, M, nullptr);
  for (int i = 0; i < 7; ++i) out7[i] = w.off[i];
  return DARK_OK;
}

int dark_forward_hidden(const dark_model_desc* m, const int64_t* ids, const int32_t* pos, const uint8_t* key_mask,
                        int B, int N, void* workspace, size_t workspace_bytes, void* hidden, void* stream) {
  return forward_hidden(m, ids, pos, key_mask, B, N, workspace, workspace_bytes, hidden, (hipStream_t)stream);
}

int dark_forward_hidden_begin(const dark_model_desc* m, const int64_t* ids, const int32_t* pos,
                              const uint8_t* key_mask, int B, int N, void* workspace, size_t workspace_bytes,
                              void* hidden, void* stream) {
  return forward_hidden(m, ids, pos, key_mask, B, N, workspace, workspace_bytes, hidden, (hipStream_t)stream, true);
}

int dark_forward_hidden_rows(const dark_model_desc* m, int B, int N, const int32_t* rows, int n_rows, void* workspace,
                             size_t workspace_bytes, const void* hidden, void* hidden_rows, void* stream) {
  return forward_tail(m, B, N, rows, n_rows, workspace, workspace_bytes, hidden, hidden_rows, (hipStream_t)stream);
}

int dark_lm_head_rows(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int M, void* workspace,
                      size_t workspace_bytes, float* logits, int64_t ldl, void* stream) {
  // src_rows entries are clamped to < 2^30 rows; callers pass indices produced by dark_mask_rows
  return lm_head_rows(m, hidden, src_rows, 1 << 30, M, workspace, workspace_bytes, logits, ldl, (hipStream_t)stream);
}

// ---- fused LM head + sampler (sample.hip "fused LM head", gemm256.hip EPI_HEAD)
struct HeadFuseWs {
  char* xn;        // [M, d] model dtype: normalised rows
  char* records;   // [M, P] 16-byte slice records
  int32_t* list;   // [M] rows the records could not finish
  int32_t* counters;  // [4]: [0] = length of list
  char* xn_f;      // [M, d]: the listed rows, gathered
  int P;
  size_t total;
};
static HeadFuseWs carve_head_fuse(const dark_model_desc* m, size_t M, char* base) {
  HeadFuseWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };  // (null base: sizes only)
  w.P = 4 * cdiv(m->vocab, 256);
  w.xn = take(M * m->d * esize(m->dtype));
  w.records = take(M * (size_t)w.P * 16);
  w.list = (int32_t*)take(M * 4);
  w.counters = (int32_t*)take(16);
  w.xn_f = take(M * m->d * esize(m->dtype));
  w.total = off;
  return w;
}

// Can (and should) this call take the fused path?  bf16 model, temperature > 0, nucleus filter on, no top-k, enough
// rows to fill the persistent GEMM; everything else runs dark_lm_head_rows + dark_sample_rows.
static bool head_fuse_applies(const dark_model_desc* m, int M, float temperature, float top_p, int top_k,
                              const float* logits, int64_t ldl) {
  if (ldl % 4 != 0 || ((uintptr_t)logits & 15) != 0) return false;  // the fallback GEMM stores 16 bytes a lane
  if (gemm_variant() == 1) return false;  // A/B hook: variant 1 = the unfused head
  if (m->dtype != DARK_BF16 || !(temperature > 0.f) || !(top_p >= 0.f && top_p < 1.f)) return false;
  if (top_k > 0 && top_k < m->vocab) return false;
  if (M < 1024 || m->vocab > DARK_SAMPLE_LDS_MAX_V || m->vocab % 4 != 0 || m->d % 64 != 0) return false;  // (16-byte logit stores)
  // (the fallback GEMM addresses the logits with 32-bit byte offsets: very large M x V takes the two-call form)
  return head_gemm_supports(M, m->vocab, m->d) && s256_supports(EPI_F32, M, m->vocab, m->d, ldl, logits, nullptr);
}

__global__ void head_stats_kernel(const int32_t* counters, int32_t* stats, int M) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { stats[0] += M; stats[1] += counters[0]; }
}

static int head_sample_rows(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int n_src, int M,
                            void* workspace, size_t ws_bytes, float* logits, int64_t ldl, float temperature, float top_p,
                            int top_k, int conf_mode, const float* noise, int64_t ldn, uint64_t seed, uint64_t offset,
                            uint32_t nthreads, float* conf, int64_t* x0, int32_t* stats, hipStream_t st) {
  DARK_TRY(check_model(m));
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(hidden && workspace && logits && conf && x0, "head_sample_rows: null pointer");
  DARK_CHECK_ARG(ldl >= m->vocab, "head_sample_rows: ldl=%lld < vocab=%d", (long long)ldl, m->vocab);
  const int V = m->vocab;
  if (!head_fuse_applies(m, M, temperature, top_p, top_k, logits, ldl)) {
    DARK_TRY(lm_head_rows(m, hidden, src_rows, n_src, M, workspace, ws_bytes, logits, ldl, st));
    return launch_sample_indexed(logits, ldl, M, V, temperature, top_p, top_k, conf_mode, nullptr, noise, ldn, seed, offset,
                                 nthreads, conf, x0, st);
  }
  HeadFuseWs w = carve_head_fuse(m, (size_t)M, (char*)workspace);
  DARK_CHECK_ARG(ws_bytes >= w.total, "head_sample_rows: workspace %zu < required %zu bytes", ws_bytes, w.total);
  const int w_rows = cdiv(V, 128) * 128;
  DARK_CHECK_HIP(hipMemsetAsync(w.counters, 0, 16, st));
  DARK_TRY(launch_rmsnorm(m->dtype, m->resid_dtype, hidden, m->final_norm, w.xn, src_rows, n_src, M, m->d, m->rms_eps, st));
  DARK_TRY(launch_gemm_head(w.xn, m->lm_head, w.records, M, V, m->d, head_c2(temperature), head_range_raw(temperature),
                            nullptr, w_rows, st));
  DARK_TRY(launch_head_finish(w.records, w.P, M, V, m->d, w.xn, m->lm_head, temperature, top_p, conf_mode, noise, ldn, seed,
                              offset, nthreads, conf, x0, w.list, w.counters, st));
  // the rows on the list: plain head GEMM + full sampler, sized on the device (no host round trip)
  DARK_TRY(launch_gather_listed_rows(w.xn, w.list, w.counters, w.xn_f, m->d, M, st));
  DARK_TRY(launch_gemm_s256_devrows(w.xn_f, m->lm_head, logits, M, V, m->d, ldl, w.counters, w_rows, st));
  DARK_TRY(launch_sample_indexed(logits, ldl, M, V, temperature, top_p, top_k, conf_mode, w.list, noise, ldn, seed, offset,
                                 nthreads, conf, x0, st, w.counters));
  if (stats) {
    hipLaunchKernelGGL(head_stats_kernel, dim3(1), dim3(64), 0, st, w.counters, stats, M);
    DARK_CHECK_LAUNCH();
  }
  return DARK_OK;
}

size_t dark_head_sample_workspace_bytes(const dark_model_desc* m, int M) {
  if (!m || M <= 0 || m->d <= 0 || m->vocab <= 0) return 0;
  return carve_head_fuse(m, (size_t)M, nullptr).total;
}

int dark_head_sample_rows(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int M, void* workspace,
                          size_t workspace_bytes, float* logits, int64_t ldl, float temperature, float top_p, int top_k,
                          int conf_mode, const float* noise, int64_t ldn, uint64_t seed, uint64_t offset,
                          uint32_t nthreads, float* conf, int64_t* x0, int32_t* stats, void* stream) {
  DARK_CHECK_ARG(nthreads == 0 || nthreads % 256 == 0, "head_sample_rows: nthreads=%u must be a multiple of 256", nthreads);
  return head_sample_rows(m, hidden, src_rows, 1 << 30, M, workspace, workspace_bytes, logits, ldl, temperature, top_p,
                          top_k, conf_mode, noise, ldn, seed, offset, nthreads, conf, x0, stats, (hipStream_t)stream);
}

int dark_noalloc_init(void) {
  void* p = nullptr;
  return scratch_on_current_device(&p);
}

size_t dark_noalloc_max_rows(const dark_model_desc* m) {
  if (!m || m->d <= 0) return 0;
  return kNoAllocScratchBytes / ((size_t)m->d * esize(m->dtype));
}

int dark_lm_head_rows_noalloc(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int M,
                              float* logits, int64_t ldl, void* stream) {
  DARK_TRY(check_model(m));
  if (M <= 0) return DARK_OK;
  const size_t need = align_up((size_t)M * m->d * esize(m->dtype), 256);
  DARK_CHECK_ARG(need <= kNoAllocScratchBytes,
                 "lm_head_rows_noalloc: M=%d rows of %d %s values need %zu bytes of scratch, the static buffer holds "
                 "%zu (at most %zu rows); use dark_lm_head_rows with a workspace for larger batches",
                 M, m->d, m->dtype == DARK_BF16 ? "bf16" : "fp32", need, kNoAllocScratchBytes,
                 kNoAllocScratchBytes / ((size_t)m->d * esize(m->dtype)));
  // the scratch array resolved below is the CURRENT device's copy: refuse a stream of another device before anything
  // is enqueued on it (hipStreamGetDevice; the null stream belongs to the current device)
  if (stream) {
    int cur = 0;
    hipDevice_t sdev = 0;
    DARK_CHECK_HIP(hipGetDevice(&cur));
    DARK_CHECK_HIP(hipStreamGetDevice((hipStream_t)stream, &sdev));
    DARK_CHECK_ARG((int)sdev == cur, "lm_head_rows_noalloc: the stream belongs to device %d, the current device is %d", (int)sdev, cur);
  }
  void* ws = nullptr;
  DARK_TRY(scratch_on_current_device(&ws));
  // from here on only `ws` (the resolved device pointer) is used, never the array's name
  return lm_head_rows(m, hidden, src_rows, 1 << 30, M, ws, kNoAllocScratchBytes, logits, ldl, (hipStream_t)stream);
}

// ---- vocabulary-sharded head (BASELINE config 3)
int dark_final_norm_rows(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int M, void* out,
                         void* stream) {
  DARK_TRY(check_model(m));
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(hidden && out, "final_norm_rows: null pointer");
  return launch_rmsnorm(m->dtype, m->resid_dtype, hidden, m->final_norm, out, src_rows, 1 << 30, M, m->d, m->rms_eps, (hipStream_t)stream);
}

int dark_lm_head_shard(const dark_model_desc* m, const void* xn, int M, int v0, int Vs, float* logits, int64_t ldl,
                       void* stream) {
  DARK_TRY(check_model(m));
  if (M <= 0) return DARK_OK;
  DARK_CHECK_ARG(xn && logits, "lm_head_shard: null pointer");
  DARK_CHECK_ARG(v0 >= 0 && v0 % 128 == 0 && Vs >= 1 && v0 + Vs <= m->vocab && ldl >= Vs,
                 "lm_head_shard: columns [%d, %d) of a vocabulary of %d (the first column must be a multiple of 128)", v0,
                 v0 + Vs, m->vocab);
  const char* w = (const char*)m->lm_head + (size_t)v0 * m->d * esize(m->dtype);
  return launch_gemm(m->dtype, EPI_F32, xn, w, nullptr, logits, M, Vs, m->d, ldl, (hipStream_t)stream);
}

int dark_shard_row_max(const float* logits, int64_t ldl, int M, int Vs, float temperature, float* row_max,
                       void* stream) {
  return launch_shard_row_max(logits, ldl, M, Vs, temperature, row_max, (hipStream_t)stream);
}

int dark_shard_partials(const float* logits, int64_t ldl, int M, int Vs, int v0, float temperature,
                        const float* row_max, void* parts, void* stream) {
  return launch_shard_partials(logits, ldl, M, Vs, v0, temperature, row_max, parts, (hipStream_t)stream);
}

int dark_sample_from_shards(const void* parts, int64_t ld_parts, int n_shards, int M, int V, float temperature, float top_p, int top_k,
                            int conf_mode, const float* row_max, const float* noise, int64_t ldn, uint64_t seed,
                            uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0, uint8_t* fallback,
                            void* stream) {
  return launch_sample_from_shards(parts, ld_parts, n_shards, M, V, temperature, top_p, top_k, conf_mode, row_max, noise, ldn,
                                   seed, offset, nthreads, conf, x0, fallback, (hipStream_t)stream);
}

int dark_sample_rows_indexed(const float* logits, int64_t ldl, int n, int V, float temperature, float top_p, int top_k,
                             int conf_mode, const int32_t* row_ids, const float* noise, int64_t ldn, uint64_t seed,
                             uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0, void* stream) {
  DARK_CHECK_ARG(n <= 0 || row_ids, "sample_rows_indexed: null row_ids");
  return launch_sample_indexed(logits, ldl, n, V, temperature, top_p, top_k, conf_mode, row_ids, noise, ldn, seed,
                               offset, nthreads, conf, x0, (hipStream_t)stream);
}

int dark_mask_rows(const int64_t* x, int B, int N, int64_t mask_id, int32_t* row_index, int32_t* src_rows,
                   int32_t* count, void* stream) {
  DARK_CHECK_ARG(x && row_index && src_rows && count, "mask_rows: null pointer");
  return launch_mask_rows(x, B, N, mask_id, row_index, src_rows, count, (hipStream_t)stream);
}

int dark_sample_rows(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p, int top_k,
                     int conf_mode, const float* noise, int64_t ldn, float* conf, int64_t* x0, void* stream) {
  return launch_sample(logits, ldl, M, V, temperature, top_p, top_k, conf_mode, noise, ldn, conf, x0,
                       (hipStream_t)stream);
}

int dark_sample_rows_philox(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p, int top_k,
                            int conf_mode, uint64_t seed, uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0,
                            void* stream) {
  DARK_CHECK_ARG(nthreads > 0 && nthreads % 256 == 0, "sample_rows_philox: nthreads=%u must be a positive multiple of 256",
                 nthreads);
  return launch_sample_philox(logits, ldl, M, V, temperature, top_p, top_k, conf_mode, nullptr, 0, seed, offset,
                              nthreads, conf, x0, (hipStream_t)stream);
}

int dark_sampler_set_row_workspace(void* ptr, size_t bytes) { return set_sampler_row_workspace(ptr, bytes); }
int dark_dbg_set_sampler_stages(int stages) {
  DARK_CHECK_ARG(stages == 0 || stages == 1, "set_sampler_stages: %d (0 = automatic, 1 = single-stage)", stages);
  set_sampler_stages(stages);
  return DARK_OK;
}

int dark_philox_exponential(uint64_t seed, uint64_t offset, uint32_t nthreads, uint64_t numel, float* out,
                            void* stream) {
  return launch_philox_exponential(seed, offset, nthreads, numel, out, (hipStream_t)stream);
}

int dark_commit_topk(int64_t* x, int B, int N, int64_t mask_id, const int32_t* row_index, int M, const float* conf,
                     const int64_t* x0, int k, void* stream) {
  DARK_CHECK_ARG(x && (M == 0 || (row_index && conf && x0)), "commit_topk: null pointer");
  return launch_commit(x, B, N, mask_id, row_index, M, conf, x0, k, (hipStream_t)stream);
}

int dark_commit_sampled(int64_t* x, int B, int N, int64_t mask_id, const int32_t* row_index, int M, const float* conf,
                        const int64_t* x0, int k, float alg_temp, const float* exp_noise, void* stream) {
  DARK_CHECK_ARG(x && (M == 0 || (row_index && conf && x0)), "commit_sampled: null pointer");
  DARK_CHECK_ARG(alg_temp > 0.f && exp_noise, "commit_sampled: needs alg_temp > 0 and the [B,N] Exp(1) draws");
  return launch_commit(x, B, N, mask_id, row_index, M, conf, x0, k, (hipStream_t)stream, alg_temp, exp_noise);
}

int dark_scatter_tokens(int64_t* x, int B, int N, const int32_t* row_index, int M, const int64_t* sel,
                        const int64_t* tokens, int n, void* stream) {
  DARK_CHECK_ARG(B > 0 && N > 0, "scatter_tokens: bad B=%d N=%d", B, N);
  return launch_scatter_tokens(x, (int64_t)B * N, row_index, M, sel, tokens, n, (hipStream_t)stream);
}

int dark_denoise_step(const dark_model_desc* m, int64_t* x, const int32_t* pos, const uint8_t* key_mask, int B, int N,
                      int M, int64_t mask_id, float temperature, float top_p, int top_k, int conf_mode,
                      const float* noise, int k, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DARK_TRY(check_model(m));
  DARK_CHECK_ARG(x && workspace, "denoise_step: null pointer");
  DARK_CHECK_ARG(B > 0 && N > 0 && M >= 0 && M <= B * N, "denoise_step: bad B=%d N=%d M=%d", B, N, M);
  StepWs w = carve_step(m, B, N, M, (char*)workspace);
  DARK_CHECK_ARG(workspace_bytes >= w.total, "denoise_step: workspace %zu < required %zu bytes", workspace_bytes,
                 w.total);
  if (M == 0) return forward_hidden(m, x, pos, key_mask, B, N, w.fwd, w.fwd_bytes, w.hidden, st);
  // the last layer's o_proj / MLP only for the M rows the head reads (dark_forward_hidden_begin / _rows)
  DARK_TRY(forward_hidden(m, x, pos, key_mask, B, N, w.fwd, w.fwd_bytes, w.hidden, st, true));
  DARK_TRY(launch_mask_rows(x, B, N, mask_id, w.row_index, w.src_rows, w.count, st));
  DARK_TRY(forward_tail(m, B, N, w.src_rows, M, w.fwd, w.fwd_bytes, w.hidden, w.hidden_rows, st));
  DARK_TRY(lm_head_rows(m, w.hidden_rows, nullptr, M, M, w.head, w.head_bytes, w.logits, m->vocab, st));
  DARK_TRY(launch_sample(w.logits, m->vocab, M, m->vocab, temperature, top_p, top_k, conf_mode, noise, m->vocab,
                         w.conf, w.x0, st));
  DARK_TRY(launch_commit(x, B, N, mask_id, w.row_index, M, w.conf, w.x0, k, st));
  return DARK_OK;
}

int dark_prof_select_gemm(int epi) { return prof_select(epi); }
int dark_prof_select_gemm_rows(int epi, int rows) { return prof_select(epi, rows); }
int dark_prof_read(double* total_ms, double* total_flops, int* launches) {
  DARK_CHECK_ARG(total_ms && total_flops && launches, "prof_read: null pointer");
  return prof_read(total_ms, total_flops, launches);
}

int dark_dbg_set_gemm_variant(int variant) {
  DARK_CHECK_ARG((variant >= 0 && variant <= 14) || (variant >= 16 && variant <= 16 + 255) || (variant >= 1000 && variant < 1128),
                 "set_gemm_variant: %d", variant);
  set_gemm_variant(variant);
  return DARK_OK;
}

// ---- unit-test hooks: single-kernel launches (tests/test_hip_kernels.py) ----
int dark_dbg_gemm(int dtype, int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                  int64_t ldo, void* stream) {
  return launch_gemm(dtype, epi, A, W, bias, out, M, N, K, ldo, (hipStream_t)stream);
}
int dark_dbg_gemm_head(const void* A, const void* W, void* records, int M, int N, int K, float temperature,
                       const int32_t* m_dev, void* stream) {
  DARK_CHECK_ARG(temperature > 0.f, "dbg_gemm_head: temperature must be positive");
  return launch_gemm_head(A, W, records, M, N, K, head_c2(temperature), head_range_raw(temperature), m_dev,
                          cdiv(N, 128) * 128, (hipStream_t)stream);
}
int dark_dbg_gemm_resid_norm(const void* A, const void* W, float* hidden, const float* ln_w, void* xn, int M, int K,
                             float eps, void* stream) {
  return launch_gemm_resid_norm(A, W, hidden, ln_w, xn, M, K, eps, (hipStream_t)stream);
}
int dark_dbg_gemm_resid_norm_h(const void* A, const void* W, void* hidden, const float* ln_w, void* xn, int M, int K,
                               float eps, void* stream) {
  return launch_gemm_resid_norm_h(A, W, hidden, ln_w, xn, M, K, eps, (hipStream_t)stream);
}
int dark_dbg_attention(int dtype, const void* qkv, const float* cosT, const float* sinT, const int32_t* pos,
                       const uint8_t* key_mask, void* out, int B, int N, int H, int HD, int max_pos, void* stream) {
  DARK_CHECK_ARG(qkv && cosT && sinT && out && B > 0 && H > 0, "dbg_attention: bad argument");
  return launch_attention(dtype, qkv, cosT, sinT, pos, key_mask, out, B, N, H, HD, max_pos, (hipStream_t)stream);
}
int dark_dbg_rmsnorm(int dtype, int x_dtype, const void* x, const float* w, void* out, const int32_t* src_rows, int n_src,
                     int M, int d, float eps, void* stream) {
  DARK_CHECK_ARG(x && w && out, "dbg_rmsnorm: null pointer");
  return launch_rmsnorm(dtype, x_dtype, x, w, out, src_rows, n_src, M, d, eps, (hipStream_t)stream);
}
int dark_dbg_embed(int dtype, int x_dtype, const int64_t* ids, const void* table, void* out, int M, int d, int vocab,
                   void* stream) {
  DARK_CHECK_ARG(ids && table && out && M > 0, "dbg_embed: bad argument");
  return launch_embed(dtype, x_dtype, ids, table, out, M, d, vocab, (hipStream_t)stream);
}

}  // extern "C"
