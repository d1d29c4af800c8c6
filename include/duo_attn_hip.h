/*
 * duo_attn_hip.h — C ABI of the MI355X (gfx950) DuoAttention hot path.
 *
 * This is the drop-in boundary below the Python patch API.  Every entry point
 * takes plain device pointers, element strides and a hipStream_t (passed as
 * void*); there are no torch types.  All tensors are bf16 unless stated.
 * Return value: 0 on success, a hipError_t (>0) for a HIP failure, or a
 * negative DUO_E* code for an argument error.  Nothing here falls back to the
 * CPU: if the shared library is missing the Python host raises.
 *
 * What each entry point replaces in the reference (mit-han-lab/duo-attention):
 *
 *   duo_rope_inplace_bf16     flashinfer.rope.apply_rope_inplace as called from
 *                             duo_attn/patch/flashinfer_utils.py:29-59
 *                             (<- duo_attn/patch/llama.py:347-352)
 *   duo_kv_append_bf16        DuoAttentionStaticKVCache.put_full_kv copy_ pair,
 *                             duo_attn/patch/static_kv_cache.py:109-125
 *   duo_stream_compress_bf16  DuoAttentionStaticKVCache
 *                             .compress_and_replace_streaming_kv,
 *                             duo_attn/patch/static_kv_cache.py:127-167
 *                             (fed by the torch.cat at llama.py:385-390)
 *   duo_attn_decode_bf16      the two flash_attn_func calls of the decode
 *                             branch, duo_attn/patch/llama.py:392-421 (q_len==1)
 *   duo_attn_prefill_bf16     flash_attn_func at llama.py:366-372 (first chunk,
 *                             all heads causal) and llama.py:392-421 (later
 *                             chunks, one call per head class)
 *   duo_decode_layer_bf16     llama.py:332-425 for q_len == 1 (whole decode step of a layer)
 *   duo_attn_prefill_f16      flash_attn_func of the INT4 demo's prefill, demo/w8a8kv4_llama.py:226-274
 *   duo_rmsnorm_bf16          flashinfer.norm.rmsnorm, flashinfer_utils.py:9-16
 *   duo_int4_quantize / duo_int4_dequantize_f16 / duo_int4_stream_compress /
 *   duo_attn_decode_int4_f16  demo/quantize_int4.cu:9-178, demo/int4_kv.py:261-492
 *   *_batched_*               the same calls with the reference's batch dimension
 *                             (static_kv_cache.py:60-125; flash_attn_func batches natively):
 *                             the batch row is a grid dimension, one launch for all rows
 *
 *   duo_silu_mul_bf16         act_fn(gate) * up of HF's LlamaMLP on prefill chunks (<- static_kv_cache.py:528-537)
 *   duo_token_linear_bf16     the torch.nn.Linear / RMSNorm / SiLU*mul / residual-add modules either side of the
 *                             attention op at q_len == 1: llama.py:332-340, :430-432; static_kv_cache.py:482-537
 *   duo_tuple_decode_prep_bf16  the data movement of the TUPLE-cache forward at q_len == 1 (llama.py:146-306): HF rotary
 *                             on q / k (:177-184), the torch.cat of the new row onto the retrieval cache (:202-223) as
 *                             an in-place append, the streaming cache's cat + sink/recent truncation (:273-301)
 *
 * ABI version 2 (round 3): duo_kv_seg gained `batch_stride`, the `_batched` entry points were
 * added, duo_int4_dequantize_f16 / duo_attn_decode_int4_f16 take a `fused` flag.
 * ABI version 3: + duo_token_linear_bf16, duo_silu_mul_bf16 (additions only).
 * ABI version 4 (round 4, additions only; a v3 caller that zeroed `reserved` is unaffected):
 *   duo_token_linear_args.reserved became `flags` (DUO_LINEAR_NORM_HF), + duo_tuple_decode_prep_bf16 (the tuple-cache
 *   decode step), + duo_rope_hf_inplace_bf16 / duo_rmsnorm_hf_bf16 (its prefill chunks), + duo_decode_layer_batched_dev_bf16
 *   (batched decode step with device-side lengths).
 * ABI version 5 (round 5, one addition, no signature changed): + duo_decode_plan_bucket; every bf16 decode entry point sizes its
 *   split-KV grid from that bucket of the visible rows (static_kv_cache.py:44-45 keeps the lengths as Python ints, so the
 *   reference has no captured step to keep consistent — here a captured launch equals the eager launch of its bucket).
 * ABI version 6 (round 6, two additions, no signature changed): + duo_debug_prefill_last_plan, duo_debug_prefill_plan; the prefill entry points choose
 *   key-range splits for BOTH head classes (up to 16 pieces) by replaying the launch on the chip's 256 CUs, and
 *   duo_attn_prefill_workspace_bytes() doubled (2048 partials).  Results of a launch may be split — summed — differently than by a v5 build; outputs stay inside the same bar.
 *
 * Attention semantics (flash-attn 2.6.3 flash_attn_func, causal=True,
 * bottom-right aligned): a query at row i of the S new rows sees every key of
 * segment A (the cached pool, lenA rows) and keys j <= i of segment B (the S
 * new rows).  softmax scale = `scale`, fp32 softmax/accumulate, P rounded to
 * bf16 before P.V, bf16 output.  GQA: q head h uses kv head h / group.
 */
#ifndef DUO_ATTN_HIP_H
#define DUO_ATTN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DUO_ABI_VERSION 6

/* argument errors (negative so they never collide with hipError_t) */
#define DUO_EINVAL   (-1)  /* bad pointer / size / stride                     */
#define DUO_EHEADDIM (-2)  /* head_dim other than 128                         */
#define DUO_EGROUP   (-3)  /* unsupported q-heads-per-kv-head group size      */
#define DUO_EWORKSPC (-4)  /* workspace too small                             */

/*
 * One key/value source for one head class.  `k`/`v` point at (token 0, first
 * kv head of the class, dim 0); element (t, h, d) lives at
 * base + t*token_stride + h*head_stride + d   (strides in ELEMENTS, d
 * contiguous).  This covers the reference's token-major [B, T, h, D] pools
 * (static_kv_cache.py:60-99), head-major pools, and the freshly projected
 * k/v activations [B, S, Hkv, D] alike.
 */
typedef struct duo_kv_seg {
    const void *k;
    const void *v;
    int64_t token_stride;
    int64_t head_stride;
    int32_t len;   /* rows in this segment                                    */
    int32_t _pad;
    int64_t batch_stride;  /* elements between the batch rows of this segment (batched entry points; 0 otherwise) */
} duo_kv_seg;

/*
 * One head class (retrieval = "full", or streaming) of one layer.
 * Keys visible to the class are segA (entirely) then segB (causally).
 * For the retrieval class after put_full_kv the pool already holds the new
 * rows: pass segA = pool[:past], segB = pool[past:past+S].
 * For the streaming class: segA = streaming pool[:len], segB = new k/v rows.
 */
typedef struct duo_head_class {
    int32_t n_kv_heads;      /* 0 => class absent in this layer               */
    int32_t q_head_offset;   /* first q head of the class (full heads first)  */
    duo_kv_seg segA;
    duo_kv_seg segB;
} duo_head_class;

int duo_abi_version(void);
/* "gfx950" — the only architecture this library is built for */
const char *duo_target_arch(void);
const char *duo_error_string(int code);

/* bit 0: prefill uses scalar LDS gathers instead of ds_read_b64_tr_b16 for V
 *        (slow, debugging aid for the transpose-read layout);
 * bit 1: decode skips the merge launch (output invalid; lets a profiler or a
 *        HIP-event pair bracket the split-KV kernel alone);
 * bit 2: decode K/V loads temporal instead of non-temporal;  bit 3: no register prefetch;
 * bit 4: INT4 decode on the scalar-FMA kernel instead of the matrix-core one;
 * bit 5 / bit 6: INT4 / bf16 decode consume their loads without the arithmetic (output
 *        invalid: the memory-side ceiling of the launch);
 * bit 7: bf16 prefill stays on the 8-wave x 32-row kernel (the 4-wave x 64-row kernel is the default where it
 *        applies);  bit 8: no key-range split of the prefill launch;  bits 12-15 / bits 16-19: force that many key-range
 *        splits of the retrieval / the streaming class (capped by the workspace and the tile count);  bit 21: the prefill
 *        kernel's first bulk run stops at the end of segment A instead of chaining into segment B (the round-1..5 form);  bit 9: decode scan on the long-prologue kernel
 *        (duo_decode_split_kernel) instead of the short-prologue one;  bit 10: prefill in the plain q-tile-major block
 *        order instead of the XCD-aware one;  bit 11: INT4 decode on the dequantising kernel whatever `fused` asks for.
 *        Measurement / test aids only. */
void duo_set_debug_flags(uint32_t flags);
uint32_t duo_get_debug_flags(void);
/* The plan of the calling thread's last prefill launch (tests, probes): out4 = {key-range pieces of the retrieval class, of
 * the streaming class, the planner's estimate of the launch in microseconds, its estimate without any split} — the
 * flash_attn_func call sites this replaces (llama.py:366-372, :392-421) have no such choice to report. */
void duo_debug_prefill_last_plan(double *out4);
/* Host-only (no GPU, no launch): the plan for a launch shape and the work of every block id of that launch.
 * shape = {n_kv_heads of the retrieval class, of the streaming class, group, n_tokens, lenA / lenB of the retrieval class,
 * lenA / lenB of the streaming class, partials the workspace holds, order (bit 0 / 1: XCD-aware order of class 0 / 1)};
 * force: bits 0-7 / 8-15 forced piece counts, bit 16 no split; out5 = {pieces of class 0, of class 1, blocks, partials,
 * blocks of class 0}; est2 (may be NULL) = estimated microseconds as planned / unsplit; blocks (may be NULL): six int32
 * per block id {class, q tile or -1 for padding, kv head, q head of the group, piece, partial slot}.  Returns the number
 * of blocks, or a negative DUO_E* code. */
int32_t duo_debug_prefill_plan(const int32_t *shape10, uint32_t force, int32_t *out5, double *est2, int32_t *blocks,
                               int32_t blocks_cap);

/* ---- RoPE (NeoX / rotate-half, interleave=False) in place on q and k -------
 * q: [n_tokens, n_q_heads, 128], k: [n_tokens, n_kv_heads, 128];
 * position of token t = pos0 + t; angle = pos / rope_scale * theta^(-2i/128);
 * fp32 trig, result rounded to bf16.                                          */
int duo_rope_inplace_bf16(void *q, int64_t q_token_stride, int64_t q_head_stride,
                          int32_t n_q_heads, void *k, int64_t k_token_stride,
                          int64_t k_head_stride, int32_t n_kv_heads,
                          int32_t n_tokens, int64_t pos0, float rope_scale,
                          float rope_theta, int32_t head_dim, void *stream);
/* fp16 twin: apply_rope_inplace of the INT4-KV path's fp16 model (demo/w8a8kv4_llama.py:207-215) */
int duo_rope_inplace_f16(void *q, int64_t q_token_stride, int64_t q_head_stride,
                         int32_t n_q_heads, void *k, int64_t k_token_stride,
                         int64_t k_head_stride, int32_t n_kv_heads,
                         int32_t n_tokens, int64_t pos0, float rope_scale,
                         float rope_theta, int32_t head_dim, void *stream);

/* ---- copy S new rows of n_heads kv heads into a pool at row `dst_row0` ----- */
int duo_kv_append_bf16(const void *k_src, const void *v_src,
                       int64_t src_token_stride, int64_t src_head_stride,
                       void *k_pool, void *v_pool, int64_t pool_token_stride,
                       int64_t pool_head_stride, int32_t n_heads,
                       int32_t n_tokens, int32_t dst_row0, int32_t head_dim,
                       void *stream);

/* ---- streaming pool update ------------------------------------------------
 * Logical input = pool[:cur_len] ++ new[:n_new] (the torch.cat of
 * llama.py:385-390).  If cur_len + n_new <= sink + recent the new rows are
 * appended; otherwise the pool becomes  first `sink` rows ++ last `recent`
 * rows of the logical input.  In place (overlap-safe).  Returns the new
 * length through *new_len (host int).                                         */
int duo_stream_compress_bf16(void *k_pool, void *v_pool,
                             int64_t pool_token_stride, int64_t pool_head_stride,
                             const void *k_new, const void *v_new,
                             int64_t new_token_stride, int64_t new_head_stride,
                             int32_t n_heads, int32_t cur_len, int32_t n_new,
                             int32_t sink, int32_t recent, int32_t head_dim,
                             int32_t *new_len, void *stream);

/* ---- decode (S == 1): split-KV scan of both head classes in ONE launch,
 *      followed by the merge launch.  q/out: [n_q_heads, 128] (row stride
 *      q_head_stride / out_head_stride).  `workspace` holds fp32 partials;
 *      duo_attn_decode_workspace_bytes() bounds the size needed.              */
int64_t duo_attn_decode_workspace_bytes(int32_t n_q_heads, int32_t max_splits);
int duo_attn_decode_bf16(const void *q, int64_t q_head_stride, void *out,
                         int64_t out_head_stride, int32_t group,
                         const duo_head_class *full, const duo_head_class *stream_cls,
                         float scale, int32_t head_dim, void *workspace,
                         int64_t workspace_bytes, void *stream);

/* ---- one decode step of one layer of the static dual-cache path, fused ------
 * Everything reference llama.py:332-425 does for q_len == 1 after a prefill
 * (RoPE, put_full_kv, the two flash_attn_func calls, the torch.cat and
 * compress_and_replace_streaming_kv) in TWO launches:
 *   1. the split-KV scan of both head classes (as duo_attn_decode_bf16) over the
 *      rows cached so far; q is rotated as it is loaded, and the workgroup that
 *      owns a kv head's last split rotates the new key row, scores it and, for a
 *      retrieval head, appends the rotated K and the V row to the full pool at
 *      row `full_len`;
 *   2. merge of the partials + the streaming pool's sink+recent update (the new
 *      streaming key row is rotated on its way into the pool).
 * q, k, v are inputs only: un-rotated projections of the new token.
 * kv heads [0, n_full) are retrieval heads, the rest streaming heads; q heads
 * follow in the same order (n_q_heads / n_kv_heads per kv head).
 * *new_stream_len receives the streaming pool length after the step.          */
typedef struct duo_decode_layer_args {
    void *q;                 /* [n_q_heads, 128] un-rotated; read only            */
    int64_t q_head_stride;
    int32_t n_q_heads;
    int32_t n_kv_heads;
    void *k;                 /* [n_kv_heads, 128] new key rows, un-rotated; read only */
    const void *v;           /* [n_kv_heads, 128] new value rows                  */
    int64_t kv_head_stride;
    void *out;               /* [n_q_heads, 128]                                  */
    int64_t out_head_stride;
    int32_t n_full;
    int32_t head_dim;
    void *full_k, *full_v;   /* full pool, row 0 / head 0                         */
    int64_t full_token_stride, full_head_stride;
    int32_t full_len;        /* rows cached before this token                     */
    int32_t full_capacity;   /* rows allocated                                    */
    void *str_k, *str_v;     /* streaming pool                                    */
    int64_t str_token_stride, str_head_stride;
    int32_t str_len;
    int32_t sink, recent;
    int32_t _pad;
    int64_t pos;             /* position id of the new token                      */
    float rope_scale, rope_theta;
    float scale;             /* softmax scale                                     */
    float _pad2;
} duo_decode_layer_args;
int duo_decode_layer_bf16(const duo_decode_layer_args *args, int32_t *new_stream_len,
                          void *workspace, int64_t workspace_bytes, void *stream);

/* ---- batched forms ---------------------------------------------------------------------------------------------
 * The reference's pools, counters and forward carry a batch dimension (static_kv_cache.py:60-125, llama.py:309-434;
 * flash_attn_func batches natively).  Every batch row has the same lengths (the reference keeps ONE counter per layer);
 * rows may start at different RoPE positions (position_ids[:, 0], llama.py:347-352).  The batch row is a grid
 * dimension of the same kernels: one launch (pair) for all rows instead of one per row.  Strides in elements; the
 * duo_kv_seg::batch_stride fields of the head classes say how far apart the rows of each segment are.           */
typedef struct duo_decode_batch {
    int32_t n_batch;
    int32_t _pad;
    int64_t q_batch_stride;      /* q   [B, n_q_heads, 128]                                   */
    int64_t kv_batch_stride;     /* k/v [B, n_kv_heads, 128]: the new token's rows            */
    int64_t out_batch_stride;
    int64_t full_batch_stride;   /* retrieval pool                                            */
    int64_t str_batch_stride;    /* streaming pool                                            */
    const int64_t *pos;          /* HOST array of n_batch position ids, or NULL: args->pos for every row.  Rows at
                                    different positions are issued as one launch pair per row.                    */
} duo_decode_batch;
/* `args` describes batch row 0; *new_stream_len as duo_decode_layer_bf16 (the same for every row) */
int duo_decode_layer_batched_bf16(const duo_decode_layer_args *args, const duo_decode_batch *batch,
                                  int32_t *new_stream_len, void *workspace, int64_t workspace_bytes, void *stream);
int duo_attn_decode_batched_bf16(const void *q, int64_t q_batch_stride, int64_t q_head_stride, void *out,
                                 int64_t out_batch_stride, int64_t out_head_stride, int32_t n_batch, int32_t group,
                                 const duo_head_class *full, const duo_head_class *stream_cls, float scale,
                                 int32_t head_dim, void *workspace, int64_t workspace_bytes, void *stream);
int duo_attn_prefill_batched_bf16(const void *q, int64_t q_batch_stride, int64_t q_token_stride, int64_t q_head_stride,
                                  void *out, int64_t out_batch_stride, int64_t out_token_stride, int64_t out_head_stride,
                                  int32_t n_batch, int32_t n_tokens, int32_t group, const duo_head_class *full,
                                  const duo_head_class *stream_cls, float scale, int32_t head_dim, void *workspace,
                                  int64_t workspace_bytes, void *stream);
int duo_attn_prefill_batched_f16(const void *q, int64_t q_batch_stride, int64_t q_token_stride, int64_t q_head_stride,
                                 void *out, int64_t out_batch_stride, int64_t out_token_stride, int64_t out_head_stride,
                                 int32_t n_batch, int32_t n_tokens, int32_t group, const duo_head_class *full,
                                 const duo_head_class *stream_cls, float scale, int32_t head_dim, void *workspace,
                                 int64_t workspace_bytes, void *stream);
/* pos0: HOST array of n_batch first positions; equal positions -> one launch, else one per row */
int duo_rope_inplace_batched_bf16(void *q, int64_t q_batch_stride, int64_t q_token_stride, int64_t q_head_stride,
                                  int32_t n_q_heads, void *k, int64_t k_batch_stride, int64_t k_token_stride,
                                  int64_t k_head_stride, int32_t n_kv_heads, int32_t n_batch, int32_t n_tokens,
                                  const int64_t *pos0, float rope_scale, float rope_theta, int32_t head_dim,
                                  void *stream);
int duo_rope_inplace_batched_f16(void *q, int64_t q_batch_stride, int64_t q_token_stride, int64_t q_head_stride,
                                 int32_t n_q_heads, void *k, int64_t k_batch_stride, int64_t k_token_stride,
                                 int64_t k_head_stride, int32_t n_kv_heads, int32_t n_batch, int32_t n_tokens,
                                 const int64_t *pos0, float rope_scale, float rope_theta, int32_t head_dim,
                                 void *stream);
int duo_kv_append_batched_bf16(const void *k_src, const void *v_src, int64_t src_batch_stride,
                               int64_t src_token_stride, int64_t src_head_stride, void *k_pool, void *v_pool,
                               int64_t pool_batch_stride, int64_t pool_token_stride, int64_t pool_head_stride,
                               int32_t n_batch, int32_t n_heads, int32_t n_tokens, int32_t dst_row0,
                               int32_t head_dim, void *stream);
int duo_stream_compress_batched_bf16(void *k_pool, void *v_pool, int64_t pool_batch_stride,
                                     int64_t pool_token_stride, int64_t pool_head_stride, const void *k_new,
                                     const void *v_new, int64_t new_batch_stride, int64_t new_token_stride,
                                     int64_t new_head_stride, int32_t n_batch, int32_t n_heads, int32_t cur_len,
                                     int32_t n_new, int32_t sink, int32_t recent, int32_t head_dim,
                                     int32_t *new_len, void *stream);

/* ---- the same step with device-side lengths (graph-captured decode) ----------------
 * The reference keeps the cache lengths as Python ints (static_kv_cache.py:44-45), which bakes them
 * into every launch and rules out capturing the decode step in a graph.  Here each layer has a
 * duo_decode_state in DEVICE memory; duo_decode_layer_dev_bf16 issues the same two launches as
 * duo_decode_layer_bf16 but the kernels read full_len / str_len / pos from *dev_state.  The
 * full_len / str_len / pos fields of `args` are only planning hints (they size the split-KV grid; the
 * balanced partition in the kernel adapts to the real length), so a captured launch stays valid as the
 * cache grows.  The caller keeps full_len + 1 <= full_capacity.
 * The split-KV grid of every bf16 decode entry point (not the INT4 ones: no captured INT4 step exists) is a function of duo_decode_plan_bucket(visible rows) — the rows' 64-token
 * units rounded up to a power of two — not of the length itself: a captured launch therefore equals the eager launch of
 * every length in its bucket bit for bit, and the owner of a graph re-captures when the bucket of full_len + 1 (or of
 * str_len + 1) leaves the captured one (duo_attn/graph.py does); replaying beyond it stays correct, only the grid is the
 * shorter context's (ABI v5).
 * duo_decode_state_add advances (or rewinds: evict_last) the states of all layers in one launch:
 *   full_len = max(0, full_len + d_full); str_len = clamp(str_len + d_str, 0, str_cap); pos += d_pos. */
typedef struct duo_decode_state {
    int32_t full_len;        /* rows in the retrieval pool (kv_seq_len_list[l])              */
    int32_t str_len;         /* rows in the streaming pool (streaming_kv_seq_len_list[l])    */
    int32_t pos;             /* position id of the next token                                */
    int32_t _pad;
} duo_decode_state;
int duo_decode_layer_dev_bf16(const duo_decode_layer_args *args, const duo_decode_state *dev_state,
                              void *workspace, int64_t workspace_bytes, void *stream);
int duo_decode_state_add(duo_decode_state *dev_states, int32_t n_layers, int32_t d_full, int32_t d_str,
                         int32_t d_pos, int32_t str_cap, void *stream);
int32_t duo_decode_plan_bucket(int32_t n_tokens);
/* the batched step (duo_decode_layer_batched_bf16) with device-side lengths: every batch row shares the layer's
 * duo_decode_state (the reference keeps ONE counter per layer, static_kv_cache.py:44-45).  args->pos is the HOST's view
 * of dev_state->pos at the time of the call (the cache length); batch row b runs at dev_state->pos + (batch->pos[b] -
 * args->pos) — left-padded batches: a row's offset from the counter is fixed for the life of the sequence, so a captured
 * launch stays valid.  One launch pair when every row has the same offset, one pair per row otherwise.               */
int duo_decode_layer_batched_dev_bf16(const duo_decode_layer_args *args, const duo_decode_batch *batch,
                                      const duo_decode_state *dev_state, void *workspace, int64_t workspace_bytes,
                                      void *stream);

/* ---- the same step in ONE launch ------------------------------------------------------------------
 * duo_decode_layer_bf16 / _dev_bf16 issue two launches (scan — which also updates the pool of every streaming head it scans with one workgroup — then the merge).  Here both
 * are folded into the scan kernel: every workgroup publishes its split-KV partial (agent-scope release) and
 * takes a ticket of its kv head; the last min(splits, 4 * group) arrivals of a head wait until all of its
 * partials are published (bounded spin, agent-scope acquire) and merge one (q head, 32-dim quarter) each;
 * a streaming head's workgroup runs that head's sink+recent pool update after its own scan.  Replaces the
 * reference's ~12 launches per layer and step (llama.py:332-425) with one; results are bit-identical to
 * the two-launch form (same partials, same merge arithmetic).
 * `tickets`: DUO_DECODE_TICKET_BYTES of device memory, ZERO-FILLED ONCE by the caller and owned by one
 * stream at a time; every launch leaves it zeroed again.  dev_state NULL: lengths / position from `args`
 * (*new_stream_len is written); non-NULL: read on the device as duo_decode_layer_dev_bf16 does.
 * Falls back to the two launches when group is not 1, 2 or 4 (a kv head's q heads then span several
 * workgroup rows).                                                                                   */
#define DUO_DECODE_TICKET_BYTES 4096
int duo_decode_step_bf16(const duo_decode_layer_args *args, int32_t *new_stream_len,
                         const duo_decode_state *dev_state, void *workspace, int64_t workspace_bytes,
                         void *tickets, void *stream);

/* ---- prefill / chunked prefill (S >= 1): MFMA flash attention --------------
 * q/out: [S, n_q_heads, 128] with the given token/head strides.  Per head class: segA rows are all
 * visible; segB (len >= S) ends with the S query rows — query i sees segB rows 0 .. i + (len - S)
 * (bottom-right causal alignment, as flash_attn_func with seqlen_q < seqlen_k).  len == S is the
 * chunked-prefill call of llama.py:366-421; len > S processes a chunk in row blocks.          */
int duo_attn_prefill_bf16(const void *q, int64_t q_token_stride,
                          int64_t q_head_stride, void *out,
                          int64_t out_token_stride, int64_t out_head_stride,
                          int32_t n_tokens, int32_t group,
                          const duo_head_class *full, const duo_head_class *stream_cls,
                          float scale, int32_t head_dim, void *stream);
/* fp16 twin (q, K, V, out fp16): the chunked-prefill attention of the INT4 path over dequantised pools
 * (flash_attn_func at demo/w8a8kv4_llama.py:226-274) and of fp16 models.  Same layouts and semantics. */
int duo_attn_prefill_f16(const void *q, int64_t q_token_stride,
                          int64_t q_head_stride, void *out,
                          int64_t out_token_stride, int64_t out_head_stride,
                          int32_t n_tokens, int32_t group,
                          const duo_head_class *full, const duo_head_class *stream_cls,
                          float scale, int32_t head_dim, void *stream);

/* Prefill with a caller-owned workspace: lets the launcher split the retrieval class over KEY RANGES when its
 * long workgroups (retrieval q heads x 256-row q tiles) would not fill the 256 CUs — small chunks, layers with
 * one or two retrieval kv heads.  Each split leaves un-normalised partials in the workspace and a second
 * launch merges them.  Results are the same up to fp32 summation order.  duo_attn_prefill_workspace_bytes()
 * is the size that never limits the split choice; a smaller (or NULL) workspace only limits / disables it. */
int64_t duo_attn_prefill_workspace_bytes(void);
int duo_attn_prefill_ws_bf16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                             void *out, int64_t out_token_stride, int64_t out_head_stride,
                             int32_t n_tokens, int32_t group, const duo_head_class *full,
                             const duo_head_class *stream_cls, float scale, int32_t head_dim,
                             void *workspace, int64_t workspace_bytes, void *stream);
int duo_attn_prefill_ws_f16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                            void *out, int64_t out_token_stride, int64_t out_head_stride,
                            int32_t n_tokens, int32_t group, const duo_head_class *full,
                            const duo_head_class *stream_cls, float scale, int32_t head_dim,
                            void *workspace, int64_t workspace_bytes, void *stream);

/* ---- INT4 KV pools (BASELINE config 5): the reference's only native code,
 * demo/quantize_int4.cu, and the attention over its pools ---------------------
 * Row = one (token, kv head): 128 values -> 64 packed bytes (even element in the
 * HIGH nibble) + fp16 scale + fp16 zero (group_size = head_dim = 128, demo/int4_kv.py:140).
 * scale = (max-min)/15 + 1e-8 (fp32), zero = min, q = clamp(roundf((x-zero)/scale),0,15)
 * (quantize_int4.cu:104-143); dequantised value = hadd(hmul(half(q), scale), zero) in fp16
 * (:27-40).  Pools: packed bytes at q + row*64, (scale, zero) fp16 pair at sz + row*2 with
 * row = t*token_stride_rows + h*head_stride_rows.                                        */
typedef struct duo_int4_pool {
    const void *k_q, *v_q;        /* packed nibbles                                        */
    const void *k_sz, *v_sz;      /* fp16 (scale, zero) pairs                              */
    int64_t token_stride_rows;
    int64_t head_stride_rows;
    int32_t len;                  /* rows per head to attend to                            */
    int32_t n_kv_heads;           /* 0 => class absent                                     */
    int32_t q_head_offset;
    int32_t _pad;
    int64_t batch_stride_rows;    /* pool rows between batch rows (batched entry points; 0 otherwise) */
} duo_int4_pool;

/* quantise n_tokens rows of n_heads heads of src ([T, h, 128] fp16 or bf16, element strides)
 * straight into the pool at row dst_row0 (replaces quantize_int4.cu:73-178 + the copy_ chains of
 * demo/int4_kv.py:296-365)                                                                 */
int duo_int4_quantize(const void *src, int32_t src_is_bf16, int64_t src_token_stride,
                      int64_t src_head_stride, void *q_pool, void *sz_pool,
                      int64_t pool_token_stride_rows, int64_t pool_head_stride_rows,
                      int32_t n_heads, int32_t n_tokens, int32_t dst_row0, int32_t head_dim,
                      void *stream);
/* pool rows [0, n_tokens) x n_heads -> out [n_tokens, n_heads, 128] fp16 contiguous
 * (quantize_int4.cu:9-71; kept for DuoAttentionStaticINT4KVCache.get()).
 * `fused` picks the rounding of q*s + z: 0 = hadd(hmul(q, s), z), two roundings — the source as written
 * (quantize_int4.cu:38-39) and what its -ffp-contract=off build computes; 1 = one fma, what a compiler that
 * contracts the pair emits (hipcc by default; plausibly nvcc under the reference's --use_fast_math,
 * demo/int4_kv.py:46-56).  Both are pinned bit for bit against builds of the reference's own file
 * (tests/golden/int4_ref.npz: `nocontract` / `default`); they differ by one fp16 ulp on ~48 % of the values.   */
int duo_int4_dequantize_f16(const void *q_pool, const void *sz_pool, int64_t pool_token_stride_rows,
                            int64_t pool_head_stride_rows, void *out, int32_t n_heads,
                            int32_t n_tokens, int32_t head_dim, int32_t fused, void *stream);
/* streaming pool: keep the first `sink` and the last `recent` of `len` rows, in place
 * (DuoAttentionStaticINT4KVCache.compress, demo/int4_kv.py:438-492)                        */
int duo_int4_stream_compress(void *k_q, void *k_sz, void *v_q, void *v_sz,
                             int64_t pool_token_stride_rows, int64_t pool_head_stride_rows,
                             int32_t n_heads, int32_t len, int32_t sink, int32_t recent,
                             int32_t *new_len, void *stream);
/* decode attention (one fp16 query token) straight over the packed pools: dequantisation in
 * registers instead of the reference's dequantise-everything-to-scratch + flash_attn_func
 * (demo/int4_kv.py:373-436, demo/w8a8kv4_llama.py:240-274).  q/out [n_q_heads, 128] fp16.
 * `fused`: 0 / 1 = the dequantisation form, as duo_int4_dequantize_f16 (the attention then sees exactly the values
 * that function would have written to scratch);  2 / 3 = FOLDED (+ form 0 / 1 in the tiles that are not): in every 32-key
 * tile whose rows are tame (K and V scales < 1, zero points in (-8, 8)) there is no per-element dequantisation — the matrix
 * cores multiply the raw nibbles, scale and zero are applied to the 16 x 16 score tile and to P (score = s_k (N.Q^T) + z_k
 * sum_d q_d, out = (P s').N' + sum_k p_k z'_k): the attention over n s + z WITHOUT the dequantiser's fp16 roundings (<= 1
 * fp16 ulp per element away from either form), ~70 % of the instruction issue of the dequantising kernel; a tile with an
 * outlier row (where one fp16 ulp of a value is large enough to show in the reference's own output) dequantises element by
 * element as fused = 0 / 1 does.  Needs head-major pools (token_stride_rows == 1); other layouts run as fused & 1.       */
int duo_attn_decode_int4_f16(const void *q, int64_t q_head_stride, void *out, int64_t out_head_stride,
                             int32_t group, const duo_int4_pool *full, const duo_int4_pool *stream_cls,
                             float scale, int32_t head_dim, int32_t fused, void *workspace,
                             int64_t workspace_bytes, void *stream);

/* batched forms of the INT4 entry points (as the bf16 ones above: the batch row is a grid dimension; rows of a pool are
 * `pool_batch_stride_rows` apart, dequantised output rows `out_batch_stride` elements apart)                              */
int duo_int4_quantize_batched(const void *src, int32_t src_is_bf16, int64_t src_batch_stride, int64_t src_token_stride,
                              int64_t src_head_stride, void *q_pool, void *sz_pool, int64_t pool_batch_stride_rows,
                              int64_t pool_token_stride_rows, int64_t pool_head_stride_rows, int32_t n_batch,
                              int32_t n_heads, int32_t n_tokens, int32_t dst_row0, int32_t head_dim, void *stream);
int duo_int4_dequantize_batched_f16(const void *q_pool, const void *sz_pool, int64_t pool_batch_stride_rows,
                                    int64_t pool_token_stride_rows, int64_t pool_head_stride_rows, void *out,
                                    int64_t out_batch_stride, int32_t n_batch, int32_t n_heads, int32_t n_tokens,
                                    int32_t head_dim, int32_t fused, void *stream);
int duo_int4_stream_compress_batched(void *k_q, void *k_sz, void *v_q, void *v_sz, int64_t pool_batch_stride_rows,
                                     int64_t pool_token_stride_rows, int64_t pool_head_stride_rows, int32_t n_batch,
                                     int32_t n_heads, int32_t len, int32_t sink, int32_t recent, int32_t *new_len,
                                     void *stream);
int duo_attn_decode_int4_batched_f16(const void *q, int64_t q_batch_stride, int64_t q_head_stride, void *out,
                                     int64_t out_batch_stride, int64_t out_head_stride, int32_t n_batch, int32_t group,
                                     const duo_int4_pool *full, const duo_int4_pool *stream_cls, float scale,
                                     int32_t head_dim, int32_t fused, void *workspace, int64_t workspace_bytes,
                                     void *stream);

/* ---- RMSNorm: y = x * rsqrt(mean(x^2) + eps) * w, rows of `hidden` bf16 ---- */
int duo_rmsnorm_bf16(const void *x, const void *w, void *y, int64_t n_rows,
                     int32_t hidden, float eps, void *stream);

/* ---- SiLU(gate) * up of the SwiGLU MLP on whole chunks (HF LlamaMLP.forward: act_fn(gate_proj(x)) * up_proj(x), called from
 * the decoder layer at duo_attn/patch/static_kv_cache.py:528-537): one pass, silu(gate) rounded to bf16 before the product
 * as the module sequence does.  [n_rows, n_cols] bf16, row strides in elements (multiples of 8), 16-byte aligned bases. ---- */
int duo_silu_mul_bf16(const void *gate, int64_t gate_row_stride, const void *up, int64_t up_row_stride, void *y,
                      int64_t y_row_stride, int64_t n_rows, int32_t n_cols, void *stream);

/* ---- token-row linear layers of the decode step -------------------------------------------------------------------
 * At q_len == 1 the projections either side of the attention op are matrix-vector products over weights that are read
 * once per token (HBM-bound streaming): q/k/v_proj and o_proj of the static forward (duo_attn/patch/llama.py:332-340,
 * :430-432) and the MLP + norms + residual adds of its decoder layer (duo_attn/patch/static_kv_cache.py:482-537).
 *
 *   y[b, n] = sum_k W[n, k] * xn[b, k] + bias[n]  (+ residual[b, n])        b < n_rows <= DUO_TOKEN_LINEAR_MAX_ROWS
 *   xn = x                                   when norm_weight == NULL and x2 == NULL
 *      = rmsnorm(x; norm_weight, norm_eps)   (flashinfer.norm.rmsnorm: fp32, one rounding to bf16 — the static path's norm,
 *                                            flashinfer_utils.py:9-26; with DUO_LINEAR_NORM_HF in `flags` the HuggingFace
 *                                            LlamaRMSNorm / MistralRMSNorm form the TUPLE path keeps: the normalised x is
 *                                            rounded to bf16 BEFORE the multiplication by the weight, which rounds again)
 *      = silu(x) * x2                        (LlamaMLP's act_fn(gate) * up; silu(x) rounded to bf16 first, as the module does)
 * W: up to three row-major [n, n_in] bf16 blocks (torch.nn.Linear.weight layout) whose outputs are concatenated along n
 * (q | k | v, gate | up); seg[i].n == 0 ends the list.  fp32 accumulation, one rounding to bf16, then the residual add
 * with its own rounding — the values a module-by-module run materialises in bf16 are rounded at the same points.
 * Alignment: x, x2, norm_weight, every W block 16 bytes; strides multiples of 8 elements; n_in a multiple of 8;
 * n_rows * round_up(n_in, DUO_TOKEN_LINEAR_PAD) * 2 bytes must fit one CU's LDS (156 KiB).                              */
#define DUO_TOKEN_LINEAR_MAX_ROWS 4
#define DUO_TOKEN_LINEAR_PAD 2048     /* token rows are staged in LDS padded to a multiple of this many elements */
#define DUO_LINEAR_NORM_HF 1          /* duo_token_linear_args.flags: RMSNorm prologue in the HuggingFace two-rounding form */
typedef struct duo_linear_seg {
    const void *w;          /* [n, n_in] bf16, rows `row_stride` elements apart */
    const void *bias;       /* [n] bf16 or NULL */
    int64_t row_stride;
    int32_t n;
    int32_t reserved;
} duo_linear_seg;
typedef struct duo_token_linear_args {
    const void *x;                  /* [n_rows, n_in] bf16, rows x_row_stride elements apart */
    const void *x2;                 /* silu(x) * x2 prologue: same shape and stride as x; else NULL */
    int64_t x_row_stride;
    int32_t n_rows;
    int32_t n_in;
    duo_linear_seg seg[3];
    const void *norm_weight;        /* RMSNorm prologue: [n_in] bf16; else NULL */
    float norm_eps;
    int32_t flags;                  /* DUO_LINEAR_* bits; 0 = ABI v3 behaviour */
    const void *residual;           /* [n_rows, n_total] bf16 added to the rounded product; else NULL */
    int64_t residual_row_stride;
    void *y;                        /* [n_rows, n_total] bf16 */
    int64_t y_row_stride;
} duo_token_linear_args;
int duo_token_linear_bf16(const duo_token_linear_args *args, void *stream);

/* ---- the tuple-cache decode step's data movement (one launch, one batch row) --------------------------------------------
 * Reference llama_duo_attention_forward_one_way_reordered (duo_attn/patch/llama.py:146-306) at q_len == 1, everything
 * except the two flash_attn_func calls (duo_attn_decode_bf16 does those, over the arena and the OLD streaming cache as
 * segment A and the rotated new row as segment B) and the projections:
 *   1. HF rotary (transformers apply_rotary_pos_emb, llama.py:177-184) on q [n_q_heads, 128] and k [n_kv_heads, 128] IN
 *      PLACE, in the bf16 arithmetic torch performs: x' = bf16(bf16(x * cos) + bf16(rotate_half(x) * sin)), cos / sin the
 *      [128] bf16 rows model.rotary_emb produced for this position;
 *   2. retrieval heads (the first n_full kv heads): rotated k row and v row appended at row `full_len` of the arena
 *      (reference: torch.cat of the whole cache, :202-223) — full_len + 1 <= full_capacity;
 *   3. streaming heads: a NEW cache tensor = truncate(old ++ new row) (:273-301): all str_len + 1 rows when that is
 *      <= sink + recent, else the first `sink` old rows, the last recent - 1 old rows, the new row.  Written out of place
 *      (dst != src): tuples handed to the caller earlier stay intact.  *new_stream_len = rows written.
 * Strides in elements, multiples of 8; k / v rows kv_head_stride apart; head_dim == 128.                              */
typedef struct duo_tuple_decode_args {
    void *q; int64_t q_head_stride; int32_t n_q_heads; int32_t n_kv_heads;
    void *k; const void *v; int64_t kv_head_stride;
    const void *cos_row, *sin_row;          /* [128] bf16 each */
    int32_t n_full; int32_t head_dim;
    void *full_k, *full_v;                  /* arena: (row t, head h) at base + t*full_token_stride + h*full_head_stride */
    int64_t full_token_stride, full_head_stride;
    int32_t full_len, full_capacity;
    const void *str_k_src, *str_v_src;      /* old streaming cache, str_len rows */
    int64_t src_token_stride, src_head_stride;
    void *str_k_dst, *str_v_dst;            /* new streaming cache, min(str_len + 1, sink + recent) rows */
    int64_t dst_token_stride, dst_head_stride;
    int32_t str_len, sink, recent, _pad;
} duo_tuple_decode_args;
int duo_tuple_decode_prep_bf16(const duo_tuple_decode_args *args, int32_t *new_stream_len, void *stream);

/* ---- the tuple-cache path on prefill chunks: HF rotary and HF RMSNorm as single passes ---------------------------------
 * duo_rope_hf_inplace_bf16: transformers apply_rotary_pos_emb as the tuple forward calls it (llama.py:177-184, unsqueeze_dim = 2)
 * on q [n_tokens, n_q_heads, 128] and k [n_tokens, n_kv_heads, 128] IN PLACE, cos / sin = the [n_tokens, 128] bf16 rows of
 * model.rotary_emb (rows cos_sin_token_stride elements apart), in torch's bf16 arithmetic — x' = bf16(bf16(x * cos) +
 * bf16(rotate_half(x) * sin)): bit-equal to the six torch kernels it replaces.
 * duo_rmsnorm_hf_bf16: LlamaRMSNorm / MistralRMSNorm.forward (the tuple path leaves HF's norm modules in place, reference
 * tuple_kv_cache.py:431-490): y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))) — two roundings, where duo_rmsnorm_bf16
 * (flashinfer's form, the static path's) has one.  Strides multiples of 8 elements, 16-byte aligned bases, hidden % 8 == 0.  */
int duo_rope_hf_inplace_bf16(void *q, int64_t q_token_stride, int64_t q_head_stride, int32_t n_q_heads, void *k,
                             int64_t k_token_stride, int64_t k_head_stride, int32_t n_kv_heads, int32_t n_tokens,
                             const void *cos_rows, const void *sin_rows, int64_t cos_sin_token_stride, int32_t head_dim,
                             void *stream);
int duo_rmsnorm_hf_bf16(const void *x, const void *w, void *y, int64_t n_rows, int32_t hidden, float eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DUO_ATTN_HIP_H */
