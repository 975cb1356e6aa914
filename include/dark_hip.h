/* dark_hip.h - C ABI of the MI355X-native DARK denoise / unmask hot path.
 *
 * The reference (HKUST-KnowComp/DARK) is pure Python: its "FFI" for this path is
 * the set of ATen ops that `DreamModel_length.forward` and
 * `DreamGenerationMixin._sample` launch per denoise step.  Each entry point below
 * replaces one group of those ops and cites the reference lines it stands in for
 * (paths relative to the reference root).  All pointers are DEVICE pointers unless
 * stated otherwise; `stream` is a `hipStream_t` passed as `void*` (NULL = default
 * stream).  Every function only enqueues work on `stream`: no allocation, no
 * synchronisation, no host copies.  Return value: DARK_OK or a DARK_ERR_* code;
 * `dark_hip_last_error()` returns a thread-local message for the last failure.
 *
 * Python binding (ctypes): dark_amd/_hip.py.  INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 */
#ifndef DARK_HIP_H_
#define DARK_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DARK_HIP_ABI_VERSION 7

#define DARK_OK 0
#define DARK_ERR_ARG 1         /* bad shape / pointer / unsupported size */
#define DARK_ERR_HIP 2         /* a HIP runtime call failed */
#define DARK_ERR_UNSUPPORTED 3 /* valid request this build has no kernel for */

/* arithmetic type of weights and GEMM/attention operands (accumulation and logits are always fp32; the residual
 * stream is fp32 or -- bf16 models only -- bf16, see dark_model_desc.resid_dtype) */
#define DARK_F32 0
#define DARK_BF16 1

/* confidence definitions of generation_utils.py:505-517 */
#define DARK_CONF_PROB 0        /* probs[x0]           ('maskgit_plus', 'origin') */
#define DARK_CONF_MARGIN 1      /* top1 - top2         ('topk_margin')            */
#define DARK_CONF_NEG_ENTROPY 2 /* sum p*log(p+1e-10)  ('entropy')                */

/* One decoder layer, packed by dark_amd/modeling_dream_length.py:pack_weights().
 * `w_*` are row-major [out_features_padded, in_features] in the model dtype (rows padded with
 * zeros to a multiple of 128).  Replaces the nn.Linear / DreamRMSNorm parameters of
 * modeling_dream_length.py:270-273 (q/k/v/o), :216-218 (gate/up/down), :456-457 (norms). */
typedef struct dark_layer_weights {
  const void* w_qkv;   /* [3d, d]: q rows, then k rows, then v rows                        */
  const float* b_qkv;  /* [3d] fp32                                                        */
  const void* w_o;     /* [d, d]                                                           */
  const void* w_gu;    /* [2ff, d]: gate/up interleaved in groups of G output columns (G = 16 for bf16, 32  *
                        * for fp32): rows [2Gj, 2Gj+G) = gate[Gj..], rows [2Gj+G, 2Gj+2G) = up[Gj..]        */
  const void* w_down;  /* [d, ff]                                                          */
  const float* ln1;    /* input_layernorm.weight [d] fp32                                  */
  const float* ln2;    /* post_attention_layernorm.weight [d] fp32                         */
} dark_layer_weights;

/* Whole model (DreamConfig shape contract, configuration_dream.py:29-50). */
typedef struct dark_model_desc {
  int32_t dtype;      /* DARK_F32 | DARK_BF16 */
  int32_t vocab;      /* V */
  int32_t vocab_pad;  /* rows of lm_head, multiple of 128 */
  int32_t d;          /* hidden_size, multiple of 64 */
  int32_t ff;         /* intermediate_size, multiple of 64 */
  int32_t n_layers;
  int32_t n_heads;
  int32_t head_dim;   /* 64 or 96 */
  int32_t max_pos;    /* rows of the RoPE tables */
  float rms_eps;
  int32_t resid_dtype; /* element type of the residual stream (`hidden`, `hidden_rows` below): DARK_F32, or DARK_BF16 for a
                        * bf16 model -- what the reference's bf16 model carries (`residual + hidden_states` on bf16
                        * tensors, modeling_dream_length.py:508, :514); each residual add is then float(stream) + the
                        * fp32 accumulator, rounded once */
  const void* embed;        /* [V, d] model dtype (embed_tokens, :615) */
  const float* final_norm;  /* [d] fp32 (model.norm, :620) */
  const void* lm_head;      /* [vocab_pad, d] model dtype, untied (:743) */
  const float* rope_cos;    /* [max_pos, head_dim/2] fp32 (DreamRotaryEmbedding.forward, :151-171) */
  const float* rope_sin;    /* [max_pos, head_dim/2] fp32 */
  const dark_layer_weights* layers; /* HOST array of n_layers entries */
} dark_model_desc;

int dark_hip_abi_version(void);
const char* dark_hip_last_error(void);

/* Bytes of scratch `dark_forward_hidden` / `dark_lm_head_rows` need for the given sizes. */
size_t dark_forward_workspace_bytes(const dark_model_desc* m, int B, int N);
size_t dark_head_workspace_bytes(const dark_model_desc* m, int M);

/* Transformer body.  Replaces DreamBaseModel.forward up to, but excluding, the final
 * RMSNorm (modeling_dream_length.py:633-716): embedding gather, RoPE, L x [RMSNorm, QKV+bias,
 * RoPE, bidirectional SDPA, o_proj, residual, RMSNorm, SwiGLU MLP, residual].
 *   ids      [B,N] int64
 *   pos      [B,N] int32 position ids, or NULL for arange(N)            (generation_utils.py:465-466)
 *   key_mask [B,N] uint8 (1 = attend), or NULL for no masking; the pairwise mask is the outer
 *            AND of this vector with itself, fully-masked query rows give 0 (:469-472)
 *   hidden   [B*N, d] m->resid_dtype OUT: residual stream after the last layer (pre final norm) */
int dark_forward_hidden(const dark_model_desc* m, const int64_t* ids, const int32_t* pos,
                        const uint8_t* key_mask, int B, int N, void* workspace, size_t workspace_bytes,
                        void* hidden, void* stream);

/* The same body in two calls, for loops that only need the rows the sampler reads.  Bidirectional attention needs
 * every position's keys and values in every layer, but the OUTPUT of the last layer is read by nothing except the LM
 * head, and the head reads only `src_rows` (the reference evaluates all B*N rows, modeling_dream_length.py:688-721,
 * and drops the others at generation_utils.py:501).
 *   dark_forward_hidden_begin: everything up to and including the last layer's attention (independent of the rows);
 *     `hidden` then holds the residual stream BEFORE the last layer's o_proj.
 *   dark_forward_hidden_rows: o_proj + residual + RMSNorm + SwiGLU MLP + residual of the last layer for `rows`
 *     [n_rows] int32 only: hidden_rows [n_rows, d] m->resid_dtype OUT = the rows hidden[rows[r]] that dark_forward_hidden
 *     would have produced, bit for bit.  Same workspace as the begin call (its contents are consumed).
 * Feed hidden_rows to dark_lm_head_rows with src_rows = NULL. */
int dark_forward_hidden_begin(const dark_model_desc* m, const int64_t* ids, const int32_t* pos,
                              const uint8_t* key_mask, int B, int N, void* workspace, size_t workspace_bytes,
                              void* hidden, void* stream);
int dark_forward_hidden_rows(const dark_model_desc* m, int B, int N, const int32_t* rows, int n_rows, void* workspace,
                             size_t workspace_bytes, const void* hidden, void* hidden_rows, void* stream);

/* Final RMSNorm + LM head on selected rows.  Replaces `self.norm` (:721) and `lm_head` (:809)
 * restricted to the rows the sampler consumes, including the logit shift of
 * generation_utils.py:496 when `src_rows` comes from dark_mask_rows.
 *   src_rows [M] int32 row ids into `hidden`, or NULL for rows 0..M-1
 *   logits   [M, ldl] fp32 OUT.  When V is not a multiple of four and ldl >= ((V + 3) & ~3), the columns up to that
 *            multiple are written too (with the zero logits of lm_head's zero padding rows), so that 16-byte
 *            stores apply; every other column >= V is untouched */
int dark_lm_head_rows(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int M,
                      void* workspace, size_t workspace_bytes, float* logits, int64_t ldl, void* stream);

/* Allocation-free variant of dark_lm_head_rows.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate or pass a workspace -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * The gathered + normalised rows that dark_lm_head_rows keeps in `workspace` live in a static device array of
 * DARK_NOALLOC_SCRATCH_BYTES inside the library (one copy per GPU, held until the process ends), so
 * M * d * sizeof(model dtype) must not exceed it (dark_noalloc_max_rows(m) rows: 5461 bf16 / 2730 fp32 rows at
 * d = 768); larger M is refused with DARK_ERR_ARG.  The call makes no hipMalloc / hipFree and no allocating library
 * call.  Every call on a GPU shares that GPU's array: the array is fully rewritten for the M rows of each call, so
 * nothing has to be cleared, but two calls that OVERLAP on one GPU (different streams or host threads) race on it.
 * dark_noalloc_init() resolves the array's device address for the CURRENT device (hipGetSymbolAddress); the entry
 * point does the same on its first call on a device.  The first call of each kernel flavour on a device also loads
 * its code object and sets its dynamic-LDS attribute (hipFuncSetAttribute): before HIP graph capture, call
 * dark_noalloc_init() and make ONE warm-up call of this entry point with the shapes to be captured.  `stream` must
 * belong to the current device (checked before anything is enqueued; DARK_ERR_ARG otherwise).
 * Results are bit-identical to dark_lm_head_rows. */
#define DARK_NOALLOC_SCRATCH_BYTES (8u << 20)
int dark_noalloc_init(void);
size_t dark_noalloc_max_rows(const dark_model_desc* m);
int dark_lm_head_rows_noalloc(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int M,
                              float* logits, int64_t ldl, void* stream);

/* `mask_index = (x == mask_token_id)` + the row-major compaction `logits[mask_index]` uses
 * (generation_utils.py:489, :501).  row_index[m] = b*N+j of the m-th masked position,
 * src_rows[m] = b*N + max(j-1, 0) (the logit shift, :496); *count = M.  Buffers hold B*N ints. */
int dark_mask_rows(const int64_t* x, int B, int N, int64_t mask_id, int32_t* row_index,
                   int32_t* src_rows, int32_t* count, void* stream);

/* sample_tokens (generation_utils.py:58-90) incl. top_p_logits (:37-48) and top_k_logits (:50-55).
 *   logits [M, ldl] fp32 (read-only), V columns used
 *   temperature: 0 => greedy (probs.max); > 0 => x0 = argmax(p_norm / noise), the ATen
 *                multinomial(1) that Categorical.sample() runs (:70)
 *   top_p: nucleus threshold, applied iff 0 <= top_p < 1 (0 keeps the top-1 token; pass a negative value for
 *          `top_p=None`);  top_k: applied iff > 0
 *   noise [M, ldn] fp32 Exp(1) draws (`empty_like(probs).exponential_()`), required iff temperature > 0
 *   conf [M] fp32 OUT, x0 [M] int64 OUT */
int dark_sample_rows(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p,
                     int top_k, int conf_mode, const float* noise, int64_t ldn, float* conf,
                     int64_t* x0, void* stream);

/* Vocabularies beyond DARK_SAMPLE_LDS_MAX_V (the reference's WN18RR vocabulary is 100 + 40 943 entities + 22
 * relations + 1 = 41 066, akgr/tokenizer.py:19-27; its fp32 row no longer fits in a CU's LDS): every sampler entry
 * point (dark_sample_rows / _philox / _indexed, and the calls that sample inside dark_head_sample_rows and
 * dark_denoise_step) then keeps the rows it works on in a device workspace of M * ((V + 3) & ~3) * 4 bytes that the
 * caller registers ONCE per device (the current device of the call), since no entry point allocates.  Same program,
 * same results as a function of the row; logits stay read-only.  The registration holds until it is replaced ((NULL, 0)
 * clears it); sampler calls that overlap on one GPU share it and race, as with any workspace.  A sampler call with
 * V > DARK_SAMPLE_LDS_MAX_V and no sufficient registration returns DARK_ERR_ARG with the size it needs. */
#define DARK_SAMPLE_LDS_MAX_V 32768
#define DARK_SAMPLE_MAX_V (1 << 22)
int dark_sampler_set_row_workspace(void* ptr, size_t bytes);

/* dark_sample_rows with the Exp(1) draws computed where they are consumed instead of read from a [M, V] buffer:
 * element (m, v) is element m*V + v of what `torch.empty(M, V).exponential_()` writes on this device for a generator
 * at (seed, offset) -- ATen's grid-stride kernel over `nthreads` threads, Philox4x32-10 through rocRAND's
 * rocrand_uniform4, transformation::exponential -- so `Categorical.sample()` (generation_utils.py:70) sees the same
 * numbers while the 4*M*V-byte fill and its read disappear.  The caller advances the generator by
 * ((M*V - 1) / (4*nthreads) + 1) * 4, as ATen does (dark_amd/generation_utils.py:philox_fill_plan).
 * dark_philox_exponential materialises the same stream (tests: bit-equality with torch). */
int dark_sample_rows_philox(const float* logits, int64_t ldl, int M, int V, float temperature, float top_p, int top_k,
                            int conf_mode, uint64_t seed, uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0,
                            void* stream);
int dark_philox_exponential(uint64_t seed, uint64_t offset, uint32_t nthreads, uint64_t numel, float* out,
                            void* stream);

/* ---- vocabulary-sharded LM head + sampler (BASELINE config 3: "logit head vocabulary-sharded with an RCCL
 * all-gather of per-shard top-k").  The head's columns (modeling_dream_length.py:809) are split over S shards, one
 * per GPU; a shard's logits [M, Vs] never leave it.  Per step the shards exchange (dark_amd/dispatch.py):
 *   the normalised hidden rows of every rank's masked positions (all-gather of [M, d]);
 *   the row maxima of logits / T (all-reduce MAX of [M] floats);
 *   one DARK_SHARD_PART_BYTES record per row and shard: fixed-point mass of the shard's columns relative to the
 *   GLOBAL row maximum (integers: sums are exact in any order) and the <= DARK_SHARD_CAND_CAP elements within 6.0 of
 *   it as (order key, mass, vocabulary id).
 * dark_sample_from_shards merges the S records of a row and finishes it exactly as dark_sample_rows does on peaked
 * rows: conf / x0 are BIT-IDENTICAL to the unsharded sampler (same generator stream when Philox noise is used).  Rows
 * whose candidates do not hold top_p of the mass, hold more than 64 elements, or overflow a shard's record (flat
 * rows; also every row when the nucleus filter is off or top_k is on) get fallback[m] = 1 and no result: gather
 * their logits and run dark_sample_rows_indexed.
 *   dark_final_norm_rows: out [M, d] model dtype = final RMSNorm (:721) of hidden[src_rows[m]] (NULL = rows 0..M-1)
 *   dark_lm_head_shard: logits [M, ldl] fp32 = xn [M, d] x lm_head[v0 : v0 + Vs]^T; v0 must be a multiple of 128
 *   dark_shard_row_max: row_max[m] = max_v(logits[m, v] / T) over the shard's Vs columns (T = 0: no division)
 *   dark_shard_partials: parts[m] for the shard whose first column is v0, relative to row_max (the GLOBAL maxima)
 *   dark_sample_from_shards: parts [n_shards, ld_parts] records (shard-major, ld_parts >= M rows per shard); noise/seed/offset/nthreads as in
 *     dark_sample_rows / dark_sample_rows_philox, for the owner's own [M, V] fill
 *   dark_sample_rows_indexed: dark_sample_rows(_philox) on a compact list of n rows: logits [n, ldl] is row
 *     row_ids[i] of the [M, V] problem; noise row, Philox element index and conf / x0 slot are those of row_ids[i]. */
#define DARK_SHARD_CAND_CAP 16
#define DARK_SHARD_PART_BYTES 280
int dark_final_norm_rows(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int M, void* out,
                         void* stream);
int dark_lm_head_shard(const dark_model_desc* m, const void* xn, int M, int v0, int Vs, float* logits, int64_t ldl,
                       void* stream);
int dark_shard_row_max(const float* logits, int64_t ldl, int M, int Vs, float temperature, float* row_max,
                       void* stream);
int dark_shard_partials(const float* logits, int64_t ldl, int M, int Vs, int v0, float temperature,
                        const float* row_max, void* parts, void* stream);
int dark_sample_from_shards(const void* parts, int64_t ld_parts, int n_shards, int M, int V, float temperature, float top_p, int top_k,
                            int conf_mode, const float* row_max, const float* noise, int64_t ldn, uint64_t seed,
                            uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0, uint8_t* fallback,
                            void* stream);
int dark_sample_rows_indexed(const float* logits, int64_t ldl, int n, int V, float temperature, float top_p, int top_k,
                             int conf_mode, const int32_t* row_ids, const float* noise, int64_t ldn, uint64_t seed,
                             uint64_t offset, uint32_t nthreads, float* conf, int64_t* x0, void* stream);

/* Confidence-ranked commit (generation_utils.py:522-534 with alg_temp in {None, 0}):
 * per row, the k positions with the largest confidence (non-masked positions count as -inf, ties
 * go to the lower index) receive x0 if masked, mask_id otherwise (the re-mask quirk).
 *   x [B,N] int64 IN/OUT; row_index [M] from dark_mask_rows; conf/x0 [M] from dark_sample_rows */
int dark_commit_topk(int64_t* x, int B, int N, int64_t mask_id, const int32_t* row_index, int M,
                     const float* conf, const int64_t* x0, int k, void* stream);

/* The same commit with alg_temp > 0 (generation_utils.py:527-530): p = softmax(full_confidence / alg_temp) per row and
 * `torch.multinomial(p, k)` without replacement, which ATen evaluates as topk(p / q, k) with q one `exponential_` fill
 * of [B, N]: exp_noise is that fill (dark_philox_exponential reproduces torch's own stream).  A row without any masked
 * position makes the reference raise; here its first k positions are re-masked. */
int dark_commit_sampled(int64_t* x, int B, int N, int64_t mask_id, const int32_t* row_index, int M, const float* conf,
                        const int64_t* x0, int k, float alg_temp, const float* exp_noise, void* stream);

/* alg = 'origin' (generation_utils.py:505-510): x[mask_index] = x0 with x0 = MASK except at the masked rows that drew
 * `rand < p_transfer`; only those n rows are sampled.  x.flat[row_index[sel[i]]] = tokens[i] for i < n
 * (sel: int64 indices into the masked-row list, as torch.nonzero returns them). */
int dark_scatter_tokens(int64_t* x, int B, int N, const int32_t* row_index, int M, const int64_t* sel,
                        const int64_t* tokens, int n, void* stream);

/* One whole denoise step for a batch whose masked-row count M is known on the host
 * (uniform layouts): body forward -> mask rows -> head on masked rows -> sample -> commit.
 * Replaces one iteration of the loop at generation_utils.py:488-540 for the confidence algs.
 *   noise [M, V] fp32 or NULL when temperature == 0;  k = number_transfer_tokens of this step
 *   workspace: dark_step_workspace_bytes(m, B, N, M) bytes */
size_t dark_step_workspace_bytes(const dark_model_desc* m, int B, int N, int M);
/* Byte offsets inside the step workspace of {hidden (m->resid_dtype) [B*N,d] (the residual stream before the last layer's o_proj
 * when M > 0: the step evaluates the rest of that layer for the M masked rows only), row_index i32, src_rows i32, count i32,
 * logits fp32 [M,V], conf fp32 [M], x0 i64 [M]} so callers (hooks, parity tests) can read the
 * intermediates that generation_utils.py:495-517 exposes as Python tensors. */
int dark_step_workspace_offsets(const dark_model_desc* m, int B, int N, int M, size_t* out7);
int dark_denoise_step(const dark_model_desc* m, int64_t* x, const int32_t* pos, const uint8_t* key_mask,
                      int B, int N, int M, int64_t mask_id, float temperature, float top_p, int top_k,
                      int conf_mode, const float* noise, int k, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ---- variable-length (EXPAND / DELETE) sampler, batched.
 * Replaces the per-step tensor ops of DreamGenerationMixin._sample in
 * akgr/abduction_model/model/custom_mydream/generation_expand_utils.py:358-503.  The reference runs ONE sequence
 * per call (it edits `x[0]` only); here every sequence b of a [B,N] batch (N = max_length, rows padded with eos)
 * carries its own state, so B independent reference runs advance per launch:
 *   win[b]    int32 current_window_length (:397), IN/OUT of the edit step
 *   budget[b] int32 remaining expand_budget (:381-383, :488)
 *   limit[b]  int32 number of loop iterations the reference would allow (2*max_gen_len + 2*expand_budget, :395)
 *
 * dark_expand_prepare: window mask and tok_idx (:398-412) as key_mask/pos for dark_forward_hidden, the masked-row
 * lists of `logits[mask_index]` with the logit shift (:403, :417-418) for dark_lm_head_rows, and the exit test
 * (:404-405): active[b] = 1 iff step < limit[b] and the window still holds a mask; count[0] = masked rows M over
 * the active sequences, count[1] = active sequences.  key_mask/pos/row_index/src_rows hold B*N entries. */
int dark_expand_prepare(const int64_t* x, const int32_t* win, const int32_t* limit, int step, int B, int N,
                        int64_t mask_id, uint8_t* key_mask, int32_t* pos, int32_t* row_index, int32_t* src_rows,
                        uint8_t* active, int32_t* count, void* stream);

/* `logits[:, expand_token_id] -= 1e9` for the rows of sequences whose window is full or whose budget is spent
 * (:420-422).  logits [M, ldl] fp32 IN/OUT, row_index [M] from dark_expand_prepare. */
int dark_expand_block_logit(float* logits, int64_t ldl, int V, const int32_t* row_index, int M, int B, int N,
                            const int32_t* win, const int32_t* budget, int64_t expand_id, void* stream);

/* Per active sequence: `torch.topk(confidence, k)` over its masked positions (ties -> lower index; k is clamped
 * to their number, where the reference raises) and commit of x0 there (:434-442); pad_delete_to_right (:449-462);
 * removal of this step's DELETE tokens with MASK appended at the row end (:465-473); replacement of every
 * EXPAND token in the row by two MASKs, truncated to N (:478-491); win/budget updated (:473, :487-488).
 *   conf/x0 [M] from dark_sample_rows;  delete_id / expand_id: -1 = none
 *   snaps [3,B,N] int64 + snap_len [3,B] int32, or both NULL: the rows the reference appends to `histories`
 *   after the commit (:444-445), after deletions (:474-476), after expansions (:493-495); snap_len = visible
 *   length, -1 where the reference appends nothing. */
int dark_expand_commit_edit(int64_t* x, int B, int N, const int32_t* row_index, int M, const float* conf,
                            const int64_t* x0, int k, const uint8_t* active, int32_t* win, int32_t* budget,
                            int64_t mask_id, int64_t delete_id, int64_t expand_id, int pad_delete_to_right,
                            int64_t* snaps, int32_t* snap_len, void* stream);

/* The same step with alg_temp > 0 (:436-439): the k positions of every sequence are drawn,
 * `torch.multinomial(softmax(confidence / alg_temp), k)` == topk(p / q, k) per sequence, exp_noise [M] = the Exp(1)
 * draws q in masked-row order (for one sequence: ATen's own fill, dark_philox_exponential reproduces it). */
int dark_expand_commit_edit_sampled(int64_t* x, int B, int N, const int32_t* row_index, int M, const float* conf,
                                    const int64_t* x0, int k, const uint8_t* active, int32_t* win, int32_t* budget,
                                    int64_t mask_id, int64_t delete_id, int64_t expand_id, int pad_delete_to_right,
                                    int64_t* snaps, int32_t* snap_len, float alg_temp, const float* exp_noise,
                                    void* stream);

/* ---- forward-only log-probabilities of the diffusion GRPO trainer, as computed under no_grad for
 * the old / reference policy in akgr/abduction_model/custom_grpo2.py:681-709.
 *
 * dark_lm_head_logprob: `F.log_softmax(logits, -1).gather(-1, index)` (:133-134) fused with the final RMSNorm and
 * the LM head on selected rows: logp[i] = log_softmax(lm_head(norm(hidden[src_rows[i]])))[targets[i]].
 * bf16 models never materialise the logits (log-sum-exp partials in the GEMM epilogue); fp32 models go through
 * fp32 logits in chunks of 4096 rows.  src_rows NULL = rows 0..M-1; n_src = rows of `hidden`; targets clamp to [0,V). */
size_t dark_logprob_workspace_bytes(const dark_model_desc* m, int M);
int dark_lm_head_logprob(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int n_src,
                         const int64_t* targets, int M, void* workspace, size_t workspace_bytes, float* logp,
                         void* stream);
/* The same second stage on materialised fp32 logits [M, ldl] (selective_log_softmax's inner step, :133-134). */
int dark_logprob_from_logits(const float* logits, int64_t ldl, const int64_t* targets, int M, int V, float* logp,
                             void* stream);
/* forward_process (:373-385): noisy [3,b,l] = {all target tokens masked, masked where rand < t_p, masked where
 * rand > t_p}; completion_mask [b,l] = (rand < t_p) & target.  mask_target [b,l] uint8, rnd [b,l] fp32 uniforms. */
int dark_grpo_perturb(const int64_t* batch, const uint8_t* mask_target, const float* rnd, float t_p, int64_t mask_id,
                      int b, int l, int64_t* noisy, uint8_t* completion_mask, void* stream);
/* src_rows[s*N+j] = s*N + max(j-1, 0): get_logits' shift `cat([logits[:, :1], logits[:, :-1]])` (:412-414) as a
 * row gather for dark_lm_head_logprob. */
int dark_shift_rows(int32_t* src_rows, int S, int N, void* stream);
/* selective_log_softmax's weighting (:136-152): logp [I*3*B, N] in the order the trainer concatenates the three
 * versions per iteration; weights [3I] device fp32; mask [I*B, N] uint8 -> out [I*B, N]
 * = (p0 + (mask ? p1*w1 : p2*w2)) / 2. */
int dark_grpo_combine(const float* logp, const float* weights, const uint8_t* mask, int I, int B, int N, float* out,
                      void* stream);

/* ---- host pre / post steps of the batched evaluation as device kernels.
 * dark_build_layout: `test_unify_extract_sample_to_device_length` after tokenisation (akgr/tokenizer.py:318-335):
 * input_ids [B, Q+A+3] = [bos | query | sep | observation | eos] with the query (mode 1, 'sft_abd') or the
 * observation (mode 2, 'sft_ded') replaced by MASK (mode 0 = 'unify': nothing masked); need_to_mask (nullable)
 * [B, Q+A+3] fp32 = the reference's `attention_mask` return (1 on the side to generate; both sides for 'unify'). */
int dark_build_layout(const int64_t* query_ids, const int64_t* observation_ids, int B, int Q, int A, int mode,
                      int64_t bos_id, int64_t sep_id, int64_t eos_id, int64_t mask_id, int64_t* input_ids,
                      float* need_to_mask, void* stream);
/* The SEP split and special-token strip of akgr/abduction_model/main.py:409-421: sep_pos[0] = first SEP of ROW 0
 * (-1 if none: the reference raises IndexError); left[b] = ids[b, :sep_pos] and right[b] = ids[b, sep_pos+1:] with
 * the ids `tokenizer.decode(..., skip_special_tokens=True)` drops removed (special_lo: bit i set = id i is
 * special, ids 0..63), order kept, padded with -1; *_len [B] = kept count. */
int dark_split_strip(const int64_t* ids, int B, int N, int64_t sep_id, uint64_t special_lo, int64_t* left,
                     int32_t* left_len, int64_t* right, int32_t* right_len, int32_t* sep_pos, void* stream);

/* ---- measurement hook (no counterpart in the reference): bracket every launch of ONE GEMM flavour
 * (epi as in dark_dbg_gemm; -1 = off) with HIP events on the launch stream.  dark_prof_read waits for
 * the recorded events, returns their summed duration, the algorithmic FLOPs (2*M*N*K) of those launches
 * and their number, and resets the counters.  Used by bench.py for the live roofline figure; event
 * objects are created on first use while the hook is on (the only allocation in the library). */
int dark_prof_select_gemm(int epi);
/* the same, restricted to launches with exactly `rows` rows of A (0 = any): the last layer's tail runs the same
 * flavour on fewer rows (dark_forward_hidden_rows), the roofline figure is quoted on the full-size launches */
int dark_prof_select_gemm_rows(int epi, int rows);
int dark_prof_read(double* total_ms, double* total_flops, int* launches);

/* ---- GPT-2 backbone of the legacy denoiser: akgr/abduction_model/model/custom_gpt.py:25-114,
 * `DiffusionModel(GPT2LMHeadModel)` built by create_gpt2_transformer (:446-472).  Weights are packed by
 * dark_amd/custom_gpt.py from the HF GPT-2 state_dict: Conv1D weights ([in, out]) are stored TRANSPOSED here, i.e.
 * row-major [out_features_padded, in_features] in the model dtype like every other weight of this library. */
typedef struct dark_gpt2_layer_weights {
  const void* w_qkv;    /* [3d, d]  = attn.c_attn.weight^T (q rows, k rows, v rows)  */
  const float* b_qkv;   /* [3d]                                                      */
  const void* w_o;      /* [d, d]   = attn.c_proj.weight^T                           */
  const float* b_o;     /* [d]                                                       */
  const void* w_fc;     /* [ff, d]  = mlp.c_fc.weight^T                              */
  const float* b_fc;    /* [ff]                                                      */
  const void* w_proj;   /* [d, ff]  = mlp.c_proj.weight^T                            */
  const float* b_proj;  /* [d]                                                       */
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b; /* ln_1 / ln_2 weight and bias [d] fp32 */
} dark_gpt2_layer_weights;

typedef struct dark_gpt2_desc {
  int32_t dtype;        /* DARK_F32 | DARK_BF16 */
  int32_t vocab;
  int32_t vocab_pad;    /* rows of wte, multiple of 128 */
  int32_t d;            /* n_embd, multiple of 64 */
  int32_t ff;           /* n_inner (4 * n_embd), multiple of 64 */
  int32_t n_layers;
  int32_t n_heads;
  int32_t head_dim;     /* 64 or 96 */
  int32_t n_positions;  /* rows of wpe */
  float ln_eps;         /* layer_norm_epsilon */
  int32_t resid_dtype;  /* residual stream: DARK_F32, or DARK_BF16 for a bf16 model (as dark_model_desc.resid_dtype) */
  const void* wte;      /* [vocab_pad, d] model dtype: token embedding AND the tied LM head (transformer.wte) */
  const void* wpe;      /* [n_positions, d] model dtype (transformer.wpe) */
  const float* lnf_w;   /* transformer.ln_f weight / bias [d] fp32 */
  const float* lnf_b;
  const float* ident_cos; /* [head_dim/2] fp32, all 1: one-row rotation table for the shared attention kernel */
  const float* ident_sin; /* [head_dim/2] fp32, all 0 */
  const dark_gpt2_layer_weights* layers; /* HOST array of n_layers entries */
} dark_gpt2_desc;

/* Body of DiffusionModel.forward (custom_gpt.py:58-100 -> transformers' GPT2Model): wte[ids] + wpe[pos], L x [LayerNorm,
 * c_attn + bias, attention (1/sqrt(head_dim) scaling, key padding mask), c_proj + bias + residual, LayerNorm, c_fc +
 * bias + GELU-new, c_proj + bias + residual]; ln_f is applied by dark_gpt2_lm_head_rows.
 *   pos      [B,N] int32 position ids (custom_gpt.py:352-353: cumsum(attention_mask) - 1, padding -> 1) or NULL = arange
 *   key_mask [B,N] uint8 (1 = attend) or NULL: the [B, N] attention_mask of :344-346 (keys only; rows whose own entry
 *            is 0 come out as zeros here -- the reference never reads them: mask_index requires attention_mask == 1)
 *   causal   0 = bidirectional (what the class intends, :34-36), 1 = causal (what transformers' sdpa path computes
 *            whatever the `attn.bias` buffer holds); see gpt2.hip
 *   hidden   [B*N, d] m->resid_dtype OUT (pre ln_f).  Workspace: dark_gpt2_forward_workspace_bytes(m, B, N). */
size_t dark_gpt2_forward_workspace_bytes(const dark_gpt2_desc* m, int B, int N);
int dark_gpt2_forward_hidden(const dark_gpt2_desc* m, const int64_t* ids, const int32_t* pos, const uint8_t* key_mask,
                             int causal, int B, int N, void* workspace, size_t workspace_bytes, void* hidden,
                             void* stream);
/* ln_f + the tied LM head (custom_gpt.py:100-103) on selected rows; arguments as dark_lm_head_rows, workspace
 * >= M * d * sizeof(model dtype) rounded up to 256. */
int dark_gpt2_lm_head_rows(const dark_gpt2_desc* m, const void* hidden, const int32_t* src_rows, int M, void* workspace,
                           size_t workspace_bytes, float* logits, int64_t ldl, void* stream);
/* test hook: torch.nn.LayerNorm over (gathered) rows of the residual stream (x_dtype) -> model dtype */
int dark_dbg_layernorm(int dtype, int x_dtype, const void* x, const float* w, const float* b, void* out, const int32_t* src_rows,
                       int n_src, int M, int d, float eps, void* stream);

/* ---- unit-test hooks (no counterpart in the reference): launch ONE kernel of the path so that
 * tests/test_hip_kernels.py can pin it against the oracle in isolation.
 *   epi: 0 = +bias (out in model dtype), 1 = residual add into fp32 out, 2 = SwiGLU over interleaved
 *        gate/up rows, groups of 16 (bf16) / 32 (fp32) (out [M, N/2] model dtype), 3 = plain fp32 store,
 *        4 = bf16 only, log-sum-exp partials: bias = int64 targets [M], out = float2 [M, ldo = 2*ceil(N/128)] then
 *        float [M] target logits; 6 = gelu_new(acc + bias) (out in model dtype), 7 = residual add of acc + bias into
 *        fp32 out (the GPT-2 backbone's c_fc / c_proj); 8 / 9 = 1 / 7 on a bf16 residual stream (bf16 only):
 *        out (bf16) = bf16(float(out) + acc (+ bias)).  W has rows padded to 128. */
int dark_dbg_gemm(int dtype, int epi, const void* A, const void* W, const float* bias, void* out, int M,
                  int N, int K, int64_t ldo, void* stream);
/* LM head + sampler for the M masked rows in one call, without materialising the logits where that is possible.
 * Same inputs and the same conf / x0, bit for bit, as dark_lm_head_rows followed by dark_sample_rows(_philox)
 * (modeling_dream_length.py:809 + generation_utils.py:58-90).  For a bf16 model with temperature > 0, an active
 * nucleus filter (0 <= top_p < 1), no top-k, a vocabulary that is a multiple of 4 and M >= 1024 the head GEMM's epilogue reduces every 64-column slice of
 * a row to a 16-byte record (maximum, its column, columns in the head range, sum of the other columns' exp), one
 * wave per row rebuilds the sampler's head set from the records (recomputing single slices with the GEMM's own MFMA
 * chain where a slice holds more than one candidate), bounds the total mass Z from both sides and finishes the row
 * when the nucleus crossing is the same for every Z in the interval; rows it cannot certify (flat distributions) get
 * their logits from a second, device-sized GEMM into `logits` and go through the full sampler.  Every other case runs
 * the two-call form (so does an ldl that is not a multiple of 4).  `logits` [M, ldl] fp32 is scratch either way.  noise [M, ldn] or a Philox state (nthreads > 0) as
 * in dark_sample_rows(_philox).  stats (device int32[2], nullable): [0] += M, [1] += rows that took the fallback.
 * workspace: dark_head_sample_workspace_bytes(m, M) (>= what dark_lm_head_rows needs). */
size_t dark_head_sample_workspace_bytes(const dark_model_desc* m, int M);
int dark_head_sample_rows(const dark_model_desc* m, const void* hidden, const int32_t* src_rows, int M, void* workspace,
                          size_t workspace_bytes, float* logits, int64_t ldl, float temperature, float top_p, int top_k,
                          int conf_mode, const float* noise, int64_t ldn, uint64_t seed, uint64_t offset,
                          uint32_t nthreads, float* conf, int64_t* x0, int32_t* stats, void* stream);
/* The LM-head GEMM flavour of the fused head + sampler (dark_head_sample_rows): logits[M,N] = A[M,K] * W[N,K]^T are
 * reduced in the epilogue to one 16-byte record per (row, 64-column slice) -- {max logit, sum over the slice's other
 * columns of exp((l - max) / T), (count of columns within the sampler's head range of the max) | (column of the max
 * within the slice) << 16, 0} -- records[M][4 * ceil(N / 256)]; no logits are written.  m_dev (nullable): row count
 * read from device memory.  W rows padded to a multiple of 128.  Test hook. */
int dark_dbg_gemm_head(const void* A, const void* W, void* records, int M, int N, int K, float temperature,
                       const int32_t* m_dev, void* stream);
/* The row-panel kernel that variant 5 of dark_dbg_set_gemm_variant puts behind the forward's o_proj / down_proj for
 * bf16 models of width 768 (gemm_rowpanel.hip; not the default):
 * hidden[M,768] (fp32, in place) += A[M,K] * W[768,K]^T and, when ln_w and xn are given (both or neither),
 * xn[M,768] (bf16) = ln_w * hidden / sqrt(mean(hidden^2) + eps) from the same registers.  K a multiple of 96,
 * 16-byte aligned pointers; anything else is refused (DARK_ERR_ARG).  Replaces nn.Linear + residual add +
 * DreamRMSNorm of modeling_dream_length.py:384/:451/:69-74 (attention) and :222/:458 (MLP). */
int dark_dbg_gemm_resid_norm(const void* A, const void* W, float* hidden, const float* ln_w, void* xn, int M, int K,
                             float eps, void* stream);
/* The same kernel on a bf16 residual stream (hidden [M,768] bf16, in place): out = bf16(float(hidden) + acc), one
 * rounding, and xn = RMSNorm of the ROUNDED row -- hidden and xn are bit-identical to dark_dbg_gemm(epi 8) followed by
 * dark_dbg_rmsnorm on the bf16 stream (the sum of squares is rebuilt in that kernel's order). */
int dark_dbg_gemm_resid_norm_h(const void* A, const void* W, void* hidden, const float* ln_w, void* xn, int M, int K,
                               float eps, void* stream);
/* Which bf16 GEMM kernel dark_dbg_gemm and the forward use (A/B measurements and tests):
 *   0 = the library's choice per shape;
 *   1 = always the 128-row kernels of gemm.hip (and the unfused head);
 *   2 = the persistent 256 x 256 kernel wherever it supports the shape;
 *   4 = its staggered two-group form (gemm_bf16_s256) wherever it supports the shape, automatic tile height;
 *   8 / 9 / 10 = the same with the tile height forced to 256 / 288 / 320 rows;
 *   5 = the library's choice + the row-panel kernel for both residual projections (either stream dtype);
 *   13 = the library's choice + the row-panel kernel for o_proj only (bf16 residual stream);
 *   6 / 7 / 11 = the library's choice without its one-round down_proj routing / without q/k/v on the persistent
 *   kernel / without o_proj there (the in-loop A/Bs recorded in DESIGN.md);
 *   12 = the library's GEMM choice with the N <= 128 attention instance's staging loads issued per loop iteration
 *   instead of all up front (the round-2 form; results equal up to the contraction of one multiply-add in the rotation);
 *   14 = the library's choice with the bias / residual flavours of gemm_bf16_s256 on the sequential LOAD interval
 *   (fragment reads, then DMA issue) instead of the interleaved one; results unchanged;
 * PROBE BUILD ONLY (libdark_hip_probes.so: `make -C dark_amd/csrc probes`, -DDARK_PROBES; the shipped libdark_hip.so
 * holds no instantiation that gives wrong results and answers DARK_ERR_UNSUPPORTED at the launch):
 *   3 = timing probe of variant 2 (WRONG results); 16 + p = timing probe p of gemm_bf16_s256 (WRONG results unless
 *   p = 8, 9 or 88..90: per-phase cycle sums); 1000 + t = the library's choice with gemm_bf16_s256's TUNE = t (A/B forms
 *   of the interval structure, results unchanged; the values instantiated in gemm256.hip).
 * THREADING: this switch, dark_dbg_set_sampler_stages, dark_prof_select_gemm* and dark_sampler_set_row_workspace are
 * process-global settings read by every later launch on every stream: set them from one host thread while no other
 * thread is inside the library.  (dark_prof_* keep their event list behind a mutex; the others are single atomics.) */
int dark_dbg_set_gemm_variant(int variant);
/* A/B and test hook of the sampler: 0 = automatic (plain calls with the nucleus filter on, top-k off and V <= 26 624
 * run the two-stage form: sample_head_kernel with the row in registers, then sample_kernel for the rows it marked),
 * 1 = the single-stage kernel only.  Results are bit-identical either way (tests/test_hip_kernels.py). */
int dark_dbg_set_sampler_stages(int stages);
int dark_dbg_attention(int dtype, const void* qkv, const float* cosT, const float* sinT, const int32_t* pos,
                       const uint8_t* key_mask, void* out, int B, int N, int H, int HD, int max_pos,
                       void* stream);
/* (x_dtype = element type of the residual-stream operand: DARK_F32, or DARK_BF16 with dtype = DARK_BF16) */
int dark_dbg_rmsnorm(int dtype, int x_dtype, const void* x, const float* w, void* out, const int32_t* src_rows,
                     int n_src, int M, int d, float eps, void* stream);
int dark_dbg_embed(int dtype, int x_dtype, const int64_t* ids, const void* table, void* out, int M, int d, int vocab,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DARK_HIP_H_ */
