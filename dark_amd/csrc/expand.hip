// Variable-length (EXPAND / DELETE) sampler, batched: every sequence carries its own window length, expansion
// budget and step limit, so B independent runs of the reference's batch-1 loop
// (akgr/abduction_model/model/custom_mydream/generation_expand_utils.py:358-503) advance in one launch.
// Integer work on [B, N] ids with N <= 256: latency-bound, one workgroup (prepare) or one wave per
// sequence (commit + edits).
#include "common.h"

namespace dark {

// ------------------------------------------------------------------ prepare (:397-412)
// key_mask[b,j] = j < win[b]; pos[b,j] = j inside the window, 1 outside (tok_idx, :407-408);
// masked rows = (x == mask) & window & (step < limit[b]) in row-major order (:403, :418) with the logit shift
// (:417) folded into src_rows; active[b] = the sequence still has a masked position (:404-405);
// count[0] = masked rows, count[1] = active sequences.
__global__ __launch_bounds__(1024) void expand_prepare_kernel(const int64_t* __restrict__ x,
                                                              const int32_t* __restrict__ win,
                                                              const int32_t* __restrict__ limit, int step, int B,
                                                              int N, int64_t mask_id, uint8_t* __restrict__ key_mask,
                                                              int32_t* __restrict__ pos,
                                                              int32_t* __restrict__ row_index,
                                                              int32_t* __restrict__ src_rows,
                                                              uint8_t* __restrict__ active,
                                                              int32_t* __restrict__ count) {
  __shared__ int wave_tot[16];
  __shared__ int n_active;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = B * N;
  const int chunk = (total + 1023) / 1024;
  const int lo = min(tid * chunk, total), hi = min(lo + chunk, total);
  if (tid == 0) n_active = 0;
  for (int b = tid; b < B; b += 1024) active[b] = 0;
  __syncthreads();  // one workgroup: the zeroes above are visible to its own later stores
  int cnt = 0;
  for (int f = lo; f < hi; ++f) {
    const int b = f / N, j = f - b * N;
    const bool in_win = j < win[b];
    key_mask[f] = in_win ? 1 : 0;
    pos[f] = in_win ? j : 1;
    cnt += (in_win && step < limit[b] && x[f] == mask_id);
  }
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
    const int b = f / N, j = f - b * N;
    if (j < win[b] && step < limit[b] && x[f] == mask_id) {
      row_index[off] = f;
      src_rows[off] = (j == 0) ? f : f - 1;
      active[b] = 1;  // same value from every writer
      ++off;
    }
  }
  __syncthreads();
  int mine = 0;
  for (int b = tid; b < B; b += 1024) mine += active[b];
  if (mine) atomicAdd(&n_active, mine);
  __syncthreads();
  if (tid == 0) {
    count[0] = all;
    count[1] = n_active;
  }
}

// ------------------------------------------------------------------ EXPAND-logit blocking (:421-422)
__global__ __launch_bounds__(256) void expand_block_kernel(float* __restrict__ logits, int64_t ldl,
                                                           const int32_t* __restrict__ row_index, int M, int N,
                                                           const int32_t* __restrict__ win,
                                                           const int32_t* __restrict__ budget, int64_t expand_id) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int b = row_index[m] / N;
  if (win[b] == N || budget[b] == 0) logits[(int64_t)m * ldl + expand_id] -= 1e9f;
}

// ------------------------------------------------------------------ commit + DELETE + EXPAND edits (:434-495)
// One wave per active sequence; the row lives in LDS.  Snapshots mirror the three `histories.append`
// sites (:444-445, :474-476, :493-495): snaps[s][b][0..N) with snap_len[s][b] = visible length or -1.
constexpr int XN = 256;  // max_length supported by the edit kernel

__global__ __launch_bounds__(64) void expand_commit_edit_kernel(
    int64_t* __restrict__ x, int B, int N, const int32_t* __restrict__ row_index, int M,
    const float* __restrict__ conf, const int64_t* __restrict__ x0, int k, const uint8_t* __restrict__ active,
    int32_t* __restrict__ win, int32_t* __restrict__ budget, int64_t mask_id, int64_t delete_id,
    int64_t expand_id, int pad_delete_to_right, int64_t* __restrict__ snaps, int32_t* __restrict__ snap_len,
    float alg_temp, const float* __restrict__ qnoise) {
  __shared__ int64_t row[XN], tmp[XN + 8];
  __shared__ float cf[XN];
  __shared__ int64_t cand[XN];
  __shared__ uint8_t was[XN];
  __shared__ int sh_first, sh_nd, sh_ne;
  const int b = blockIdx.x, lane = threadIdx.x;
  if (snap_len && lane == 0) {
    for (int s = 0; s < 3; ++s) snap_len[s * B + b] = -1;
  }
  if (!active[b]) return;
  int w = win[b];
  const int wv = min(w, N);
  const int key = b * N;
  int lo = 0, hi = M;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (row_index[mid] < key) lo = mid + 1; else hi = mid;
  }
  // ---- load; masked positions take (conf, x0) in row-major order from m = lo
  int m_next = lo;
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    const bool valid = j < N;
    const int64_t t = valid ? x[(int64_t)key + j] : 0;
    const bool is_mask = valid && j < wv && t == mask_id;
    const unsigned long long bal = __ballot(is_mask);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (valid) {
      row[j] = t;
      was[j] = is_mask;
      const int m = min(m_next + rank, M - 1);  // == m_next + rank when row_index describes this x
      cf[j] = is_mask ? conf[m] : -INFINITY;
      cand[j] = is_mask ? x0[m] : mask_id;
      if (alg_temp > 0.f) tmp[j] = is_mask ? (int64_t)__float_as_uint(qnoise[m]) : 0;  // parked until the softmax below
    }
    m_next += __popcll(bal);
  }
  __syncthreads();
  if (alg_temp > 0.f) {
    // :436-439: torch.multinomial(softmax(confidence / alg_temp), k) over this sequence's masked positions ==
    // topk(p / q, k) with q the Exp(1) draws of ATen's multinomial fast path (one fill over the masked rows)
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64)
      if (was[j]) mx = fmaxf(mx, cf[j] / alg_temp);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64)
      if (was[j]) sum += expf(cf[j] / alg_temp - mx);
    sum = wave_sum(sum);
    for (int j = lane; j < N; j += 64)
      if (was[j]) cf[j] = (expf(cf[j] / alg_temp - mx) / sum) / __uint_as_float((uint32_t)tmp[j]);
    __syncthreads();
  }
  // ---- torch.topk(confidence, k) over the masked positions (ties -> lower index), :434-442
  for (int j = lane; j < N; j += 64) {
    if (!was[j]) continue;
    const uint32_t c = order_key(cf[j]);  // NaN ranks first, as in torch.topk
    int rank = 0;
    for (int j2 = 0; j2 < N; ++j2) {
      if (!was[j2]) continue;
      const uint32_t c2 = order_key(cf[j2]);
      rank += (c2 > c) || (c2 == c && j2 < j);
    }
    if (rank < k) row[j] = cand[j];
  }
  __syncthreads();
  if (snaps) {
    for (int j = lane; j < N; j += 64) snaps[((int64_t)0 * B + b) * N + j] = row[j];
    if (lane == 0) snap_len[0 * B + b] = wv;
  }
  // ---- pad_delete_to_right (:449-462): from the first DELETE anywhere in the row, this step's masked
  // positions all become DELETE
  if (pad_delete_to_right) {
    if (lane == 0) sh_first = N;
    __syncthreads();
    int first = N;
    for (int j = lane; j < N; j += 64)
      if (row[j] == delete_id) { first = j; break; }
    if (first < N) atomicMin(&sh_first, first);
    __syncthreads();
    first = sh_first;
    for (int j = lane; j < N; j += 64)
      if (j >= first && was[j]) row[j] = delete_id;
    __syncthreads();
  }
  // ---- delete (:465-476) then expand (:478-495): two order-preserving rewrites of a <= 256-token row,
  // done by one lane (a few hundred LDS accesses)
  if (lane == 0) {
    int nd = 0, o = 0;
    for (int j = 0; j < N; ++j) {
      if (was[j] && row[j] == delete_id) ++nd; else tmp[o++] = row[j];
    }
    for (; o < N; ++o) tmp[o] = mask_id;
    sh_nd = nd;
  }
  __syncthreads();
  const int nd = sh_nd;
  if (nd > 0) {
    w -= nd;
    for (int j = lane; j < N; j += 64) row[j] = tmp[j];
    __syncthreads();
    if (snaps) {
      for (int j = lane; j < N; j += 64) snaps[((int64_t)1 * B + b) * N + j] = row[j];
      if (lane == 0) snap_len[1 * B + b] = max(min(w, N), 0);
    }
  }
  if (lane == 0) {
    int ne = 0, o = 0;
    for (int j = 0; j < N; ++j) {
      if (row[j] == expand_id) {
        ++ne;
        if (o < N) tmp[o] = mask_id;
        if (o + 1 < N) tmp[o + 1] = mask_id;
        o += 2;
      } else {
        if (o < N) tmp[o] = row[j];
        ++o;
      }
    }
    sh_ne = ne;
  }
  __syncthreads();
  const int ne = sh_ne;
  if (ne > 0) {
    w += ne;
    for (int j = lane; j < N; j += 64) row[j] = tmp[j];
    __syncthreads();
    if (snaps) {
      for (int j = lane; j < N; j += 64) snaps[((int64_t)2 * B + b) * N + j] = row[j];
      if (lane == 0) snap_len[2 * B + b] = max(min(w, N), 0);
    }
  }
  for (int j = lane; j < N; j += 64) x[(int64_t)key + j] = row[j];
  if (lane == 0) {
    win[b] = w;
    budget[b] -= ne;
  }
}

}  // namespace dark

using namespace dark;

extern "C" {

int dark_expand_prepare(const int64_t* x, const int32_t* win, const int32_t* limit, int step, int B, int N,
                        int64_t mask_id, uint8_t* key_mask, int32_t* pos, int32_t* row_index, int32_t* src_rows,
                        uint8_t* active, int32_t* count, void* stream) {
  DARK_CHECK_ARG(x && win && limit && key_mask && pos && row_index && src_rows && active && count,
                 "expand_prepare: null pointer");
  DARK_CHECK_ARG(B > 0 && N > 0 && (int64_t)B * N < (1ll << 30), "expand_prepare: bad B=%d N=%d", B, N);
  hipLaunchKernelGGL(expand_prepare_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, win, limit, step, B, N,
                     mask_id, key_mask, pos, row_index, src_rows, active, count);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int dark_expand_block_logit(float* logits, int64_t ldl, int V, const int32_t* row_index, int M, int B, int N,
                            const int32_t* win, const int32_t* budget, int64_t expand_id, void* stream) {
  DARK_CHECK_ARG(M >= 0 && B > 0 && N > 0 && V > 0 && ldl >= V, "expand_block_logit: bad M=%d B=%d N=%d V=%d", M, B,
                 N, V);
  if (M == 0) return DARK_OK;
  DARK_CHECK_ARG(logits && row_index && win && budget, "expand_block_logit: null pointer");
  DARK_CHECK_ARG(expand_id >= 0 && expand_id < V, "expand_block_logit: expand id %lld outside the vocabulary",
                 (long long)expand_id);
  hipLaunchKernelGGL(expand_block_kernel, dim3(cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, logits, ldl,
                     row_index, M, N, win, budget, expand_id);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int dark_expand_commit_edit(int64_t* x, int B, int N, const int32_t* row_index, int M, const float* conf,
                            const int64_t* x0, int k, const uint8_t* active, int32_t* win, int32_t* budget,
                            int64_t mask_id, int64_t delete_id, int64_t expand_id, int pad_delete_to_right,
                            int64_t* snaps, int32_t* snap_len, void* stream) {
  return dark_expand_commit_edit_sampled(x, B, N, row_index, M, conf, x0, k, active, win, budget, mask_id, delete_id,
                                         expand_id, pad_delete_to_right, snaps, snap_len, 0.f, nullptr, stream);
}

int dark_expand_commit_edit_sampled(int64_t* x, int B, int N, const int32_t* row_index, int M, const float* conf,
                                    const int64_t* x0, int k, const uint8_t* active, int32_t* win, int32_t* budget,
                                    int64_t mask_id, int64_t delete_id, int64_t expand_id, int pad_delete_to_right,
                                    int64_t* snaps, int32_t* snap_len, float alg_temp, const float* exp_noise,
                                    void* stream) {
  DARK_CHECK_ARG(!(alg_temp > 0.f) || exp_noise || M == 0, "expand_commit_edit: alg_temp > 0 needs the [M] Exp(1) draws");
  DARK_CHECK_ARG(x && active && win && budget && (M == 0 || (row_index && conf && x0)),
                 "expand_commit_edit: null pointer");
  DARK_CHECK_ARG(B > 0 && N > 0 && N <= XN, "expand_commit_edit: N=%d must be in [1,%d]", N, XN);
  DARK_CHECK_ARG(M >= 0 && M <= B * N && k >= 0, "expand_commit_edit: bad M=%d k=%d", M, k);
  DARK_CHECK_ARG((snaps == nullptr) == (snap_len == nullptr), "expand_commit_edit: snaps and snap_len go together");
  hipLaunchKernelGGL(expand_commit_edit_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, B, N, row_index, M,
                     conf, x0, k, active, win, budget, mask_id, delete_id, expand_id, pad_delete_to_right, snaps,
                     snap_len, alg_temp, exp_noise);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

}  // extern "C"
