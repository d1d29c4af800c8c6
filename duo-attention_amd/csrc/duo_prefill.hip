// duo_prefill.hip — chunked-prefill flash attention for both DuoAttention head
// classes in one launch (gfx950, bf16 MFMA 32x32x16): the launcher / C entry points and the key-range-split merge kernel.
//
// Which kernel runs: every product launch runs duo_prefill_w64_kernel (4 waves x 64 rows, duo_prefill_w64.h) since
// round 2.  The round-1 kernel (8 waves x 32 rows) lives in its own translation unit, duo_prefill_w32_debug.hip, behind
// duo_prefill_w32_launch(): it serves the gather (non-transposed-LDS) debug layout of debug bit 0 and the same-box A/B /
// cross-check against the w64 kernel (debug bit 7, DUO_PREFILL_W64=0) — the GPU tests run every prefill case on both.
//
// Replaces flash_attn_func at duo_attn/patch/llama.py:366-372 (first chunk: all
// heads causal over the chunk) and llama.py:392-421 (later chunks: retrieval
// heads over the whole full-KV pool, streaming heads over
// [sink+recent pool rows ++ the chunk]).  Semantics: keys = segA (all visible)
// ++ segB (the S new rows, causal: query i sees j <= i) — flash-attn's
// bottom-right aligned causal mask; fp32 scores/softmax, P rounded to bf16
// before P.V (as FA2 does), bf16 output.
//
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "duo_prefill_common.h"
#include "duo_prefill_w64.h"

namespace {

// Combine the `ksplit` partials of one (q tile, q head): out = sum_s 2^((m_s - M) c) O_s / sum_s 2^((m_s - M) c) l_s.
// Block = (q tile, q head of the retrieval class) in the prefill kernel's block order; thread = 32 rows x 8
// column groups of 16 dims per pass, 8 passes.
template <bool F16>
__global__ __launch_bounds__(256) void duo_prefill_merge_kernel(const PrefillParams P) {
    const DuoClassDev C = P.cls[0];
    const int nq_c = C.n_kv_heads * P.group;
    const int b = blockIdx.x;
    const int tile = P.n_qtiles - 1 - b / nq_c;
    const int p = b % nq_c;
    const int kvh = p % C.n_kv_heads;
    const int g = p / C.n_kv_heads;
    const int qh = C.q_head_offset + kvh * P.group + g;
    const int ks = P.ksplit;
    const int j = threadIdx.x & 7;
    for (int pass = 0; pass < 8; ++pass) {
        const int r = pass * 32 + (threadIdx.x >> 3);
        const int q = tile * QBLK + r;
        if (q >= P.S) continue;
        const int64_t row0 = ((int64_t)b * ks + (int64_t)blockIdx.y * P.nblk_full) * QBLK + r;     // split s: + s * QBLK
        float M = -INFINITY;
        for (int s = 0; s < ks; ++s) M = fmaxf(M, P.ws_ml[(row0 + (int64_t)s * QBLK) * 2]);
        float L = 0.f;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < ks; ++s) {
            const int64_t row = row0 + (int64_t)s * QBLK;
            const float m = P.ws_ml[row * 2], l = P.ws_ml[row * 2 + 1];
            if (m == -INFINITY) continue;      // this split saw no key of the row: nothing to add
            const float w = fast_exp2((m - M) * P.scale_log2e);
            L = fmaf(l, w, L);
            const f32x4 *src = reinterpret_cast<const f32x4 *>(P.ws_o + row * DUO_HEAD_DIM + 16 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = acc[i] + src[i] * w;
        }
        const float inv = 1.f / L;
        bf16_t *op = P.out + (int64_t)blockIdx.y * P.o_bs + (int64_t)q * P.o_ts + (int64_t)qh * P.o_hs + 16 * j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 w2;
            w2.x = cvt_pk16<F16>(acc[i].x * inv, acc[i].y * inv);
            w2.y = cvt_pk16<F16>(acc[i].z * inv, acc[i].w * inv);
            *reinterpret_cast<u32x2 *>(op + 4 * i) = w2;
        }
    }
}

}  // namespace

static uint32_t g_debug_flags = 0;
extern "C" void duo_set_debug_flags(uint32_t flags) { g_debug_flags = flags; }
extern "C" uint32_t duo_get_debug_flags(void) { return g_debug_flags; }

// Key-range splits of the retrieval class.  One workgroup per CU is resident (96 KiB of LDS), so a launch
// whose long workgroups (retrieval q heads x q tiles) are fewer than the CUs — small chunks, layers with
// one or two retrieval kv heads — or not a multiple of them leaves CUs idle for the whole launch while
// the streaming-head workgroups are short.  With k splits the long work becomes L0*k workgroups of 1/k the
// length: pick the k that minimises rounds(L0*k) / k, with a small charge per split for the merge pass.
constexpr int64_t kPrefillPartialBytes = (int64_t)QBLK * (DUO_HEAD_DIM + 2) * sizeof(float);
static int prefill_choose_ksplit(int long_wgs, int min_tiles, int64_t workspace_bytes) {
    static const int forced = [] {
        const char *e = getenv("DUO_PREFILL_KSPLIT");   // tuning / test knob: force a split count
        return e ? atoi(e) : 0;
    }();
    if (long_wgs <= 0 || workspace_bytes <= 0) return 1;
    const int kmax = (int)std::min<int64_t>(8, std::min<int64_t>(min_tiles, workspace_bytes / (kPrefillPartialBytes * long_wgs)));
    if (g_debug_flags & 256u) return 1;      // debug bit 8: no key-range split
    if (forced > 0) return std::max(1, std::min(forced, kmax));
    if ((g_debug_flags >> 12) & 15u) return std::max(1, std::min((int)((g_debug_flags >> 12) & 15u), kmax));   // bits 12-15: tests force a count
    int best = 1;
    double best_cost = 1e30;
    for (int k = 1; k <= kmax; ++k) {
        const double rounds = (double)((long_wgs * k + 255) / 256);
        const double cost = rounds / k + 0.03 * (k - 1);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = k; }
    }
    return best;
}

extern "C" int64_t duo_attn_prefill_workspace_bytes(void) { return 1024 * kPrefillPartialBytes; }

template <bool F16>
static int prefill_impl(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                        void *out, int64_t out_token_stride, int64_t out_head_stride,
                        int32_t n_tokens, int32_t group, const duo_head_class *full,
                        const duo_head_class *stream_cls, float scale, int32_t head_dim,
                        void *workspace, int64_t workspace_bytes, void *stream,
                        int32_t n_batch = 1, int64_t q_batch_stride = 0, int64_t out_batch_stride = 0) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (q == nullptr || out == nullptr || group <= 0 || n_tokens < 0 || n_batch < 0 || n_batch > 65535) return DUO_EINVAL;
    if (n_tokens == 0 || n_batch == 0) return 0;
    if (n_batch > 1 && ((q_batch_stride | out_batch_stride) & 7)) return DUO_EINVAL;
    PrefillParams P;
    P.q_bs = q_batch_stride;
    P.o_bs = out_batch_stride;
    P.q = (const bf16_t *)q;
    P.q_ts = q_token_stride;
    P.q_hs = q_head_stride;
    P.out = (bf16_t *)out;
    P.o_ts = out_token_stride;
    P.o_hs = out_head_stride;
    P.S = n_tokens;
    P.group = group;
    P.n_qtiles = (n_tokens + QBLK - 1) / QBLK;
    P.cls[0] = duo_class_dev(full);
    P.cls[1] = duo_class_dev(stream_cls);
    P.scale_log2e = scale * 1.4426950408889634f;
    P.flags = g_debug_flags;
    int nblk = 0;
    for (int c = 0; c < 2; ++c) {
        DuoClassDev &C = P.cls[c];
        if (C.n_kv_heads <= 0) { C.n_kv_heads = 0; continue; }
        if (C.b.len < n_tokens || !C.b.k || !C.b.v) return DUO_EINVAL;   // segB ends with the query rows
        if (C.a.len < 0 || (C.a.len > 0 && (!C.a.k || !C.a.v))) return DUO_EINVAL;
        if ((C.a.token_stride | C.a.head_stride | C.b.token_stride | C.b.head_stride) & 7) {
            // rows must be 16-byte aligned for the dwordx4 tile loads
            if (C.a.len > 0 || ((C.b.token_stride | C.b.head_stride) & 7)) return DUO_EINVAL;
        }
        nblk += C.n_kv_heads * group * P.n_qtiles;
    }
    // split the retrieval class when that fills the chip better (needs the caller's workspace)
    const int long_wgs = P.cls[0].n_kv_heads * group * P.n_qtiles;
    P.ksplit = 1;
    P.ws_o = nullptr;
    P.ws_ml = nullptr;
    if (workspace && long_wgs > 0) {
        const int min_tiles = (P.cls[0].a.len + KVBLK - 1) / KVBLK + 1;   // tiles of the first q tile
        P.ksplit = prefill_choose_ksplit(long_wgs, min_tiles, workspace_bytes / n_batch);   // every batch row has its own partials
        if (P.ksplit > 1) {
            P.ws_o = (float *)workspace;
            P.ws_ml = P.ws_o + (int64_t)n_batch * long_wgs * P.ksplit * QBLK * DUO_HEAD_DIM;
            nblk += long_wgs * (P.ksplit - 1);
        }
    }
    if ((q_token_stride | q_head_stride) & 7) return DUO_EINVAL;
    if ((out_token_stride | out_head_stride) & 3) return DUO_EINVAL;
    if (nblk == 0) return 0;
    P.nblk_full = long_wgs * P.ksplit;
    P.xmap_rows = P.xmap_q = 0;

    hipStream_t st = (hipStream_t)stream;
    const bool tr = !(g_debug_flags & 1u);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return DUO_EINVAL;
    // 4-wave x 64-row kernel (duo_prefill_w64.h): the default for bf16 and fp16 with the
    // transposed-V LDS layout; the gather debug path runs on the 8-wave kernel above.  DUO_PREFILL_W64=0 (or debug
    // flag bit 7) keeps the 8-wave kernel everywhere (same-box A/B, tests of both kernels).
    static const bool want_w64 = [] { const char *e = getenv("DUO_PREFILL_W64"); return !e || atoi(e) != 0; }();
    {
        bool w64_ok = want_w64 && tr && !(g_debug_flags & 128u);
        // (with the generated bulk schedule it wins on every launch shape, first chunks and streaming-only launches
        // included: +9 ... +14 %, profiles/r2_prefill_w64.md; debug bit 8 = never split the key range, so tests reach it
        // on short launches too)
        if (w64_ok) {
            static std::atomic<bool> w64_attr[64][2];
            const void *wfn = F16 ? (const void *)duo_prefill_w64_f16_kernel : (const void *)duo_prefill_w64_kernel;
            if (dev >= 64 || !w64_attr[dev][F16].load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute(wfn,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
                if (e != hipSuccess) return (int)e;
                if (dev < 64) w64_attr[dev][F16].store(true, std::memory_order_release);
            }
            // XCD-aware order of the retrieval class (unsplit launches; DUO_PREFILL_XMAP=0 / debug bit 10: plain order)
            static const bool want_xmap = [] { const char *e = getenv("DUO_PREFILL_XMAP"); return !e || atoi(e) != 0; }();
            if (want_xmap && !(g_debug_flags & 1024u) && P.ksplit == 1 && long_wgs > 0) {
                const int row_items = P.cls[0].n_kv_heads * group;
                int rows = 1;
                while ((rows * row_items) % 8 != 0) rows *= 2;       // 1, 2, 4 or 8 rows: the first multiple of 8 workgroups
                const int periods = (P.n_qtiles + rows - 1) / rows;
                P.xmap_rows = rows;
                P.xmap_q = rows * row_items / 8;
                nblk += periods * rows * row_items - P.nblk_full;     // the padded last period
                P.nblk_full = periods * rows * row_items;
            }
            if constexpr (F16) hipLaunchKernelGGL(duo_prefill_w64_f16_kernel, dim3(nblk, n_batch), dim3(256), LDS_BYTES, st, P);
            else hipLaunchKernelGGL(duo_prefill_w64_kernel, dim3(nblk, n_batch), dim3(256), LDS_BYTES, st, P);
            DUO_HIP_CHECK_LAUNCH();
            if (P.ksplit > 1) {
                hipLaunchKernelGGL((duo_prefill_merge_kernel<F16>), dim3(long_wgs, n_batch), dim3(256), 0, st, P);
                DUO_HIP_CHECK_LAUNCH();
            }
            return 0;
        }
    }
    {       // debug / cross-check paths: the 8-wave x 32-row kernel (duo_prefill_w32_debug.hip)
        const int rc = duo_prefill_w32_launch(&P, tr, F16, nblk, n_batch, dev, st);
        if (rc) return rc;
    }
    DUO_HIP_CHECK_LAUNCH();
    if (P.ksplit > 1) {
        hipLaunchKernelGGL((duo_prefill_merge_kernel<F16>), dim3(long_wgs, n_batch), dim3(256), 0, st, P);
        DUO_HIP_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int duo_attn_prefill_bf16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                     void *out, int64_t out_token_stride, int64_t out_head_stride,
                                     int32_t n_tokens, int32_t group, const duo_head_class *full,
                                     const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                     void *stream) {
    return prefill_impl<false>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                               group, full, stream_cls, scale, head_dim, nullptr, 0, stream);
}

// Same, with a caller-owned workspace (duo_attn_prefill_workspace_bytes()) that lets the launcher split the
// retrieval class over key ranges when its workgroups would not fill the chip.
extern "C" int duo_attn_prefill_ws_bf16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                        void *out, int64_t out_token_stride, int64_t out_head_stride,
                                        int32_t n_tokens, int32_t group, const duo_head_class *full,
                                        const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                        void *workspace, int64_t workspace_bytes, void *stream) {
    return prefill_impl<false>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                               group, full, stream_cls, scale, head_dim, workspace, workspace_bytes, stream);
}

// fp16 twin (q, K, V, out all fp16): the attention of the INT4 path's chunked prefill over dequantised pools
// (demo/w8a8kv4_llama.py:226-274) and of fp16 models.
// Batched forms (n_batch rows of equal length — the reference's pools and forward carry a batch dimension,
// static_kv_cache.py:60-99, and flash_attn_func batches natively): the batch row is grid.y of the same launches.
extern "C" int duo_attn_prefill_batched_bf16(const void *q, int64_t q_batch_stride, int64_t q_token_stride,
                                             int64_t q_head_stride, void *out, int64_t out_batch_stride,
                                             int64_t out_token_stride, int64_t out_head_stride, int32_t n_batch,
                                             int32_t n_tokens, int32_t group, const duo_head_class *full,
                                             const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                             void *workspace, int64_t workspace_bytes, void *stream) {
    return prefill_impl<false>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                               group, full, stream_cls, scale, head_dim, workspace, workspace_bytes, stream, n_batch,
                               q_batch_stride, out_batch_stride);
}
extern "C" int duo_attn_prefill_batched_f16(const void *q, int64_t q_batch_stride, int64_t q_token_stride,
                                            int64_t q_head_stride, void *out, int64_t out_batch_stride,
                                            int64_t out_token_stride, int64_t out_head_stride, int32_t n_batch,
                                            int32_t n_tokens, int32_t group, const duo_head_class *full,
                                            const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                            void *workspace, int64_t workspace_bytes, void *stream) {
    return prefill_impl<true>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                              group, full, stream_cls, scale, head_dim, workspace, workspace_bytes, stream, n_batch,
                              q_batch_stride, out_batch_stride);
}

extern "C" int duo_attn_prefill_f16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                    void *out, int64_t out_token_stride, int64_t out_head_stride,
                                    int32_t n_tokens, int32_t group, const duo_head_class *full,
                                    const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                    void *stream) {
    return prefill_impl<true>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                              group, full, stream_cls, scale, head_dim, nullptr, 0, stream);
}

extern "C" int duo_attn_prefill_ws_f16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                       void *out, int64_t out_token_stride, int64_t out_head_stride,
                                       int32_t n_tokens, int32_t group, const duo_head_class *full,
                                       const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                       void *workspace, int64_t workspace_bytes, void *stream) {
    return prefill_impl<true>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                              group, full, stream_cls, scale, head_dim, workspace, workspace_bytes, stream);
}

#ifdef W64_TIMING      /* measurement builds only (tools/debug): per-phase cycle sums of the last w64 launch */
extern "C" int duo_debug_w64_timing(uint32_t *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(w64_timing), 8 * sizeof(uint32_t));
}
#endif
