// Host pre / post steps of the batched evaluation as device-side integer kernels:
// the fixed-layout builder of akgr/tokenizer.py:318-335 (after the tokenizer has produced the padded id matrices)
// and the SEP split + special-token strip of akgr/abduction_model/main.py:409-421 (what
// `tokenizer.decode(part, skip_special_tokens=True)` keeps, as ids).  Byte-sized work; one wave per row.
#include "common.h"

namespace dark {

// input_ids = [bos | query or MASK | sep | observation or MASK | eos]; need_to_mask as the reference's float map
__global__ __launch_bounds__(256) void build_layout_kernel(const int64_t* __restrict__ q,
                                                           const int64_t* __restrict__ o, int B, int Q, int A,
                                                           int mode, int64_t bos, int64_t sep, int64_t eos,
                                                           int64_t mask, int64_t* __restrict__ ids,
                                                           float* __restrict__ need) {
  const int N = Q + A + 3;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= B * N) return;
  const int b = f / N, j = f - b * N;
  int64_t v;
  float m = 0.f;
  if (j == 0) v = bos;
  else if (j <= Q) {
    v = mode == 1 ? mask : q[(int64_t)b * Q + (j - 1)];
    m = (mode == 0 || mode == 1) ? 1.f : 0.f;
  } else if (j == Q + 1) v = sep;
  else if (j <= Q + 1 + A) {
    v = mode == 2 ? mask : o[(int64_t)b * A + (j - Q - 2)];
    m = (mode == 0 || mode == 2) ? 1.f : 0.f;
  } else v = eos;
  ids[f] = v;
  if (need) need[f] = m;
}

// one wave per row: sep_pos = first SEP of ROW 0 (main.py:411-414); left = ids[b, :sep_pos], right = ids[b, sep_pos+1:],
// each with the special ids removed, order kept, padded with -1
__global__ __launch_bounds__(256) void split_strip_kernel(const int64_t* __restrict__ ids, int B, int N,
                                                          int64_t sep_id, unsigned long long special_lo,
                                                          int64_t* __restrict__ left, int32_t* __restrict__ left_len,
                                                          int64_t* __restrict__ right,
                                                          int32_t* __restrict__ right_len,
                                                          int32_t* __restrict__ sep_pos) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  int sp = N;
  for (int j0 = 0; j0 < N && sp == N; j0 += 64) {
    const int j = j0 + lane;
    const unsigned long long bal = __ballot(j < N && ids[j] == sep_id);
    if (bal) sp = j0 + __ffsll((long long)bal) - 1;
  }
  if (b == 0 && lane == 0) *sep_pos = sp < N ? sp : -1;
  if (sp == N) {  // the reference raises IndexError here; the host wrapper does too
    if (lane == 0) { left_len[b] = -1; right_len[b] = -1; }
    return;
  }
  int nl = 0, nr = 0;
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    const int64_t t = j < N ? ids[(int64_t)b * N + j] : 0;
    const bool special = t >= 0 && t < 64 && ((special_lo >> t) & 1ull);
    const bool keep_l = j < sp && !special, keep_r = j > sp && j < N && !special;
    const unsigned long long bl = __ballot(keep_l), br = __ballot(keep_r);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (keep_l) left[(int64_t)b * N + nl + __popcll(bl & below)] = t;
    if (keep_r) right[(int64_t)b * N + nr + __popcll(br & below)] = t;
    nl += __popcll(bl);
    nr += __popcll(br);
  }
  for (int j = nl + lane; j < N; j += 64) left[(int64_t)b * N + j] = -1;
  for (int j = nr + lane; j < N; j += 64) right[(int64_t)b * N + j] = -1;
  if (lane == 0) { left_len[b] = nl; right_len[b] = nr; }
}

}  // namespace dark

using namespace dark;

extern "C" {

int dark_build_layout(const int64_t* query_ids, const int64_t* observation_ids, int B, int Q, int A, int mode,
                      int64_t bos_id, int64_t sep_id, int64_t eos_id, int64_t mask_id, int64_t* input_ids,
                      float* need_to_mask, void* stream) {
  DARK_CHECK_ARG(input_ids && B > 0 && Q > 0 && A > 0 && (int64_t)B * (Q + A + 3) < (1ll << 30),
                 "build_layout: bad B=%d Q=%d A=%d", B, Q, A);
  DARK_CHECK_ARG(mode >= 0 && mode <= 2, "build_layout: mode %d (0 unify, 1 sft_abd, 2 sft_ded)", mode);
  DARK_CHECK_ARG((mode == 1 || query_ids) && (mode == 2 || observation_ids), "build_layout: null id matrix");
  const int total = B * (Q + A + 3);
  hipLaunchKernelGGL(build_layout_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, query_ids,
                     observation_ids, B, Q, A, mode, bos_id, sep_id, eos_id, mask_id, input_ids, need_to_mask);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

int dark_split_strip(const int64_t* ids, int B, int N, int64_t sep_id, uint64_t special_lo, int64_t* left,
                     int32_t* left_len, int64_t* right, int32_t* right_len, int32_t* sep_pos, void* stream) {
  DARK_CHECK_ARG(ids && left && left_len && right && right_len && sep_pos, "split_strip: null pointer");
  DARK_CHECK_ARG(B > 0 && N > 0 && (int64_t)B * N < (1ll << 30), "split_strip: bad B=%d N=%d", B, N);
  hipLaunchKernelGGL(split_strip_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, ids, B, N, sep_id,
                     (unsigned long long)special_lo, left, left_len, right, right_len, sep_pos);
  DARK_CHECK_LAUNCH();
  return DARK_OK;
}

}  // extern "C"
