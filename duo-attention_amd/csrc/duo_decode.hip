// duo_decode.hip — single-token decode attention for both DuoAttention head
// classes in one launch (gfx950).
//
// Replaces the two flash_attn_func calls of the decode branch of the reference
// (duo_attn/patch/llama.py:392-421 with q_len == 1): retrieval heads scan the
// whole full-KV pool, streaming heads scan sink+recent pool rows plus the new
// row.  The step is HBM-bound (512 B of K+V per token per kv head against
// ~2 KFLOP), so the design is a split-KV stream:
//
//   * one 256-thread workgroup per (kv head, token chunk); the G q heads of the
//     GQA group share every K/V row that is fetched;
//   * 16 lanes per token row, 16 B per lane -> every global_load_dwordx4 of a
//     wave covers 4 whole 256-B rows (full-line coalescing on either pool
//     layout), 8 such loads (4 K + 4 V) in flight per wave and the next 8
//     prefetched behind the current compute;
//   * fp32 scalar FMA for q.k and p.v, DPP row all-reduce for the 16-lane dot
//     product (no LDS), exp2-domain online softmax per 16-lane token group;
//   * per-workgroup partial (m, l, acc[128]) per q head -> fp32 workspace,
//     merged by duo_decode_merge_kernel (or written straight to `out` when a
//     class needs a single split).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "duo_kv_ops.h"

namespace {

constexpr float kNegSentinel = -1.0e30f;
constexpr int kTokPerIter = 16;  // tokens per wave per iteration (4 loads x 4 rows)

struct DecodeParams {
    const bf16_t *q;
    int64_t q_head_stride;
    bf16_t *out;
    int64_t out_head_stride;
    DuoClassDev cls[2];      // 0 = retrieval (full), 1 = streaming
    int32_t splits[2];       // token chunks per kv head
    int32_t chunk[2];        // tokens per chunk (multiple of 64)
    int32_t nblk_full;       // cls[0].n_kv_heads * splits[0]
    int32_t group;           // q heads per kv head
    float scale_log2e;
    float *ws_ml;            // [n_q_heads][max_splits][2]
    float *ws_acc;           // [n_q_heads][max_splits][128]
    int32_t max_splits;
    // ---- fused decode step (duo_decode_layer_bf16): q and the new k row arrive UN-rotated, the new
    //      token is not in the pools yet.  cls[c].a = pool rows cached before this token; cls[c].b =
    //      the new row (len 1, token stride 0).  The workgroup of a kv head's LAST split rotates the
    //      new k row, scores it and — for retrieval heads — appends k,v to the pool at row app_row.
    //      q and k are never written (the epilogue's pool update rotates the streaming rows itself).
    int32_t fused;
    uint32_t dbg;            // debug flags copy (bit 6: loads only — the memory-side ceiling of the scan)
    // device-side step state {full_len, str_len, pos, _} of this layer (duo_decode_layer_dev_bf16): when
    // set, the lengths and the position are read from here instead of the launch parameters, so a
    // captured launch stays valid while the cache grows
    const int32_t *dev_state;
    int32_t app_row;
    bf16_t *app_k, *app_v;   // full pool bases (head 0, row 0)
    int64_t app_ts, app_hs;
    float pos;
    float inv_freq[64];
    // ---- single-launch step (duo_decode_step_bf16): merge + streaming-pool update folded into this kernel.
    //      tickets[2*h] counts the workgroups of kv head h (global index: retrieval heads, then streaming heads)
    //      that have published their partial, tickets[2*h+1] the mergers that have finished; tickets[1023] is
    //      a give-up flag.  All zero on entry, all zero again on exit.
    int32_t one_launch;
    int32_t *tickets;
    uint32_t ws_bytes;       // bytes of the partial workspace behind ws_ml (buffer-descriptor bound)
    int32_t odd_dw;          // duo_decode_scan_kernel: 64ths of a share that odd-XCD workgroups give up (0 = even deal)
    // ---- batched launch: grid.z = batch row (all rows at the same lengths / position).  q / out rows are q_bs / out_bs
    //      elements apart, the segments carry their own batch strides, every row has its own partial area
    int64_t q_bs, out_bs, app_bs;
    int64_t ws_row_floats;
    float pos_delta;         // added to the device-side position (dev_state): this batch row's offset from row 0
};

constexpr int kTicketWords = DUO_DECODE_TICKET_BYTES / 4;
constexpr int kSpinLimit = 1 << 22;   // polls (with s_sleep) before a merger gives up: far beyond any real wait

__device__ __forceinline__ void unpack8(const u32x4 &w, float (&f)[8]) {
    f[0] = bf16_lo(w.x); f[1] = bf16_hi(w.x);
    f[2] = bf16_lo(w.y); f[3] = bf16_hi(w.y);
    f[4] = bf16_lo(w.z); f[5] = bf16_hi(w.z);
    f[6] = bf16_lo(w.w); f[7] = bf16_hi(w.w);
}

struct RowSrc {
    const bf16_t *ka, *va;   // segment A base for this kv head (+ lane dim offset)
    const bf16_t *kb, *vb;   // segment B base
    int64_t tsa, tsb;
    int32_t lenA;
};

// A wave-uniform global address pinned to an SGPR pair (readfirstlane), typed as a GLOBAL pointer: an
// integer-to-pointer cast alone would make it generic and the loads flat_load (vmcnt AND lgkmcnt).
typedef __attribute__((address_space(1))) const char gchar_t;
typedef __attribute__((address_space(1))) const u32x4 gu32x4_t;
__device__ __forceinline__ gchar_t *uniform_gptr(const char *p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (gchar_t *)(((uint64_t)hi << 32) | lo);
}
template <bool NT>
__device__ __forceinline__ u32x4 ld16g(gchar_t *p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(p));
    else return *reinterpret_cast<gu32x4_t *>(p);
}

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const bf16_t *p) {
    // K/V rows are read exactly once per token: the non-temporal policy keeps them from
    // displacing q / partials / the next kernel's working set in L2 and MALL
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    else return *reinterpret_cast<const u32x4 *>(p);
}

template <bool NT>
__device__ __forceinline__ void load_rows(const RowSrc &src, int tok0, int tg, int tok_end,
                                          u32x4 (&kbuf)[4], u32x4 (&vbuf)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int tok = tok0 + 4 * u + tg;
        tok = tok < tok_end ? tok : tok_end - 1;  // clamp: masked later, never OOB
        const bool inA = tok < src.lenA;
        const int64_t off = inA ? (int64_t)tok * src.tsa : (int64_t)(tok - src.lenA) * src.tsb;
        const bf16_t *kp = (inA ? src.ka : src.kb) + off;
        const bf16_t *vp = (inA ? src.va : src.vb) + off;
        kbuf[u] = ld16<NT>(kp);
        vbuf[u] = ld16<NT>(vp);
    }
}

// FULL: all 16 tokens of the group are inside the range (no masks)
template <int GT, bool FULL = false>
__device__ __forceinline__ void consume_rows(const u32x4 (&kbuf)[4], const u32x4 (&vbuf)[4],
                                             int tok0, int tg, int tok_end,
                                             const float (&qf)[GT][8], float (&m)[GT],
                                             float (&l)[GT], float (&acc)[GT][8]) {
    float s[4][GT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float kf[8];
        unpack8(kbuf[u], kf);
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            float d = qf[g][0] * kf[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) d = fmaf(qf[g][e], kf[e], d);
            s[u][g] = d;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < GT; ++g) s[u][g] = row16_allreduce_sum(s[u][g]);

    bool valid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) valid[u] = FULL || (tok0 + 4 * u + tg) < tok_end;

    float p[4][GT];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        float mn = m[g];
#pragma unroll
        for (int u = 0; u < 4; ++u) mn = fmaxf(mn, valid[u] ? s[u][g] : kNegSentinel);
        const float alpha = fast_exp2(m[g] - mn);
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            p[u][g] = valid[u] ? fast_exp2(s[u][g] - mn) : 0.f;
            psum += p[u][g];
        }
        l[g] = fmaf(l[g], alpha, psum);
        m[g] = mn;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] *= alpha;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float vf[8];
        unpack8(vbuf[u], vf);
#pragma unroll
        for (int g = 0; g < GT; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p[u][g], vf[e], acc[g][e]);
    }
}

struct MergeParams {
    const float *ws_ml;
    const float *ws_acc;
    bf16_t *out;
    int64_t out_head_stride;
    int32_t max_splits;
    // q-head ranges [begin,end) with their split counts; ranges with <=1 split are skipped
    int32_t qh_begin[2], qh_end[2], splits[2];
    int64_t ws_row_floats, out_bs;     // batched launch (grid.y = batch row)
};

// Four workgroups (256 threads) per q head, one per 32-dim quarter: 32 split lanes x 8 dim quads, so the
// ~128 partials of a retrieval head are ONE batch of loads per thread (m, l and four accumulator dims of its
// four splits, issued together).  No separate pass for the global maximum: every split lane reduces its own
// splits against its own running maximum, and the 32 lanes are combined through LDS with their maxima.
// SC1: the partials were published write-through by other workgroups of the SAME launch (single-launch
// step) — read them with sc1 loads (served by L2 / memory, never by this CU's possibly stale L1).
template <bool SC1>
__device__ __forceinline__ void duo_decode_merge_task(const float *ws_ml, const float *ws_acc, uint32_t ws_bytes,
                                                      bf16_t *out, int64_t out_head_stride, int max_splits, int qh,
                                                      int quarter, int splits) {
    const int sl = threadIdx.x >> 3;  // 0..31
    const int dq = threadIdx.x & 7;   // dims 32*quarter + 4dq .. +3
    const int d0 = 32 * quarter + 4 * dq;
    const float *ml = ws_ml + (int64_t)qh * max_splits * 2;
    const float *ac = ws_acc + (int64_t)qh * max_splits * DUO_HEAD_DIM + d0;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ws_ml, 0, ws_bytes, 0x00020000);
    const uint32_t ml_off = (uint32_t)((const char *)ml - (const char *)ws_ml);
    const uint32_t ac_off = (uint32_t)((const char *)ac - (const char *)ws_ml);
    auto ld_ml = [&](int s_) -> u32x2 {
        if constexpr (SC1) return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, ml_off + s_ * 8, 0, 16));
        else return *reinterpret_cast<const u32x2 *>(ml + s_ * 2);
    };
    auto ld_ac = [&](int s_) -> f32x4 {
        if constexpr (SC1) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ac_off + s_ * (DUO_HEAD_DIM * 4), 0, 16));
        else return *reinterpret_cast<const f32x4 *>(ac + (int64_t)s_ * DUO_HEAD_DIM);
    };

    __shared__ float sm[32];
    __shared__ float slm[32][8];
    __shared__ f32x4 so[32][8];

    float m = kNegSentinel, Lsum = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    // 4 splits (stride 32) per step with all 8 loads in flight together, for EVERY split count: the usual 60-odd splits of a
    // retrieval head are ONE round trip per thread (round 2 walked a head with fewer than 128 splits one split at a time —
    // two dependent round trips at 63 splits, most of the launch).  Lanes past the end re-read a valid split with weight 0.
    for (int s = sl; s < splits; s += 128) {
        u32x2 w[4];
        f32x4 a[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = s + 32 * u < splits;
            const int si = ok[u] ? s + 32 * u : s;
            w[u] = ld_ml(si);
            a[u] = ld_ac(si);
        }
        float mx = m;
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, ok[u] ? __uint_as_float(w[u].x) : kNegSentinel);
        const float f = fast_exp2(m - mx);
        Lsum *= f;
        o = o * f;
        m = mx;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float wu = ok[u] ? fast_exp2(__uint_as_float(w[u].x) - m) : 0.f;
            Lsum = fmaf(__uint_as_float(w[u].y), wu, Lsum);
            o = o + a[u] * wu;
        }
    }
    if (dq == 0) sm[sl] = m;
    slm[sl][dq] = Lsum;
    so[sl][dq] = o;
    __syncthreads();
    if (sl == 0) {
        float M = sm[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) M = fmaxf(M, sm[i]);
        float LL = 0.f;
        f32x4 oo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float f = fast_exp2(sm[i] - M);   // lanes without a split: exp2(-1e30 - M) = 0
            LL = fmaf(slm[i][dq], f, LL);
            oo = oo + so[i][dq] * f;
        }
        const float inv = 1.f / LL;
        u32x2 w;
        w.x = pack_bf16x2(oo.x * inv, oo.y * inv);
        w.y = pack_bf16x2(oo.z * inv, oo.w * inv);
        *reinterpret_cast<u32x2 *>(out + (int64_t)qh * out_head_stride + d0) = w;
    }
}

// block of the stand-alone epilogue launch -> (q head, 32-dim quarter)
__device__ __forceinline__ void duo_decode_merge_block(const MergeParams &P, int blk, int row) {
    int qh = blk >> 2;
    int splits;
    const int n0 = P.splits[0] > 1 ? P.qh_end[0] - P.qh_begin[0] : 0;
    if (qh < n0) {
        qh += P.qh_begin[0];
        splits = P.splits[0];
    } else {
        qh = qh - n0 + P.qh_begin[1];
        splits = P.splits[1];
    }
    duo_decode_merge_task<false>(P.ws_ml + (int64_t)row * P.ws_row_floats, P.ws_acc + (int64_t)row * P.ws_row_floats, 0,
                                 P.out + (int64_t)row * P.out_bs, P.out_head_stride, P.max_splits, qh, blk & 3, splits);
}


#ifdef DUO_DECODE_TIMING   /* measurement builds only (tools/debug/decode_timing.py): s_memtime of wave 0 per workgroup */
__device__ unsigned long long duo_decode_timing[2048][12];
#define DUO_DT(k)                                                                                  \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 2048)                              \
            duo_decode_timing[blockIdx.x][k] = __builtin_amdgcn_s_memtime();                       \
    } while (0)
/* constant-rate clock (100 MHz): the only one that compares across XCDs */
#define DUO_DRT(k)                                                                                 \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 2048)                              \
            duo_decode_timing[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime();                   \
    } while (0)
#else
#define DUO_DT(k) do { } while (0)
#define DUO_DRT(k) do { } while (0)
#endif

// grid.x = (kv head, split) pairs of the full class then of the streaming class
// grid.y = group / GT

template <int GT, bool NT, bool PREFETCH, bool FUSED>
__global__ __launch_bounds__(256) void duo_decode_split_kernel(const DecodeParams P, const CompressParams CP) {
    DUO_DT(0);
    DUO_DRT(8);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: scalar loop control
    const int sub = lane & 15;   // which 8-dim slice of the 128-dim row
    const int tg = lane >> 4;    // which of the 4 rows a wave-load covers

    int b = blockIdx.x;
    const int ci = b < P.nblk_full ? 0 : 1;
    if (ci) b -= P.nblk_full;
    DuoClassDev C = duo_select(P.cls[0], P.cls[1], ci != 0);
    const int bz = blockIdx.z;      // batch row
    duo_class_batch_row(C, bz);
    bf16_t *const out_row = P.out + (int64_t)bz * P.out_bs;
    float *const ws_ml_row = P.ws_ml + (int64_t)bz * P.ws_row_floats, *const ws_acc_row = P.ws_acc + (int64_t)bz * P.ws_row_floats;
    int app_row = P.app_row;
    float pos = P.pos;
    if constexpr (FUSED) {
        if (P.dev_state) {   // uniform scalar loads
            app_row = P.dev_state[0];
            C.a.len = ci ? P.dev_state[1] : app_row;
            pos = (float)P.dev_state[2] + P.pos_delta;
        }
    }
    const int splits = ci ? P.splits[1] : P.splits[0];
    const int kvh = b / splits;
    const int split = b - kvh * splits;
    const int qh0 = C.q_head_offset + kvh * P.group + blockIdx.y * GT;

    // balanced static partition: the head's ceil(L/64) 64-token units are dealt to the splits as
    // evenly as possible, so every workgroup of the (single-round) grid streams the same bytes
    // FUSED: the scan covers the cached rows only; the new token is handled after the loop
    const int L = FUSED ? C.a.len : C.a.len + C.b.len;
    const int units = (L + 63) >> 6;
    const int uq = units / splits, ur = units - uq * splits;
    const int u0 = split * uq + min(split, ur);
    const int un = uq + (split < ur ? 1 : 0);
    const int c0 = u0 << 6;
    const int c1 = min((u0 + un) << 6, L);
    const int per_wave = (((c1 - c0 + 3) >> 2) + 15) & ~15;  // quarter of the chunk, multiple of 16
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + wave * per_wave);
    const int w1 = __builtin_amdgcn_readfirstlane(min(w0 + per_wave, c1));

    RowSrc src;
    src.ka = C.a.k + kvh * C.a.head_stride + sub * 8;
    src.va = C.a.v + kvh * C.a.head_stride + sub * 8;
    src.kb = C.b.k + kvh * C.b.head_stride + sub * 8;
    src.vb = C.b.v + kvh * C.b.head_stride + sub * 8;
    src.tsa = C.a.token_stride;
    src.tsb = C.b.token_stride;
    src.lenA = C.a.len;
    DUO_DT(4);

    // RoPE factors of this lane's 8 dims (FUSED): dims 8*sub+e pair with dims (8*sub+e) ^ 64, i.e.
    // with the slice of lane sub ^ 8; both slices of a pair use frequency index (8*sub+e) & 63
    float cs[8], sn[8];
    if constexpr (FUSED) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sincos_rev(pos * P.inv_freq[((sub & 7) << 3) + e], sn[e], cs[e]);
    }
    DUO_DT(5);
    // x: own slice, y: partner slice -> rotated own slice (first half: x*c - y*s, second half: x*c + y*s)
    auto rope8 = [&](const u32x4 &own, const u32x4 &partner, float (&o)[8]) {
        float x[8], y[8];
        unpack8(own, x);
        unpack8(partner, y);
        const float sgn = sub < 8 ? -1.f : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // round to bf16 exactly like the standalone RoPE kernel does before the attention reads it
            o[e] = __uint_as_float(f32_to_bf16_bits(x[e] * cs[e] + sgn * y[e] * sn[e]) << 16);
        }
    };

    float qf[GT][8];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        const bf16_t *qrow = P.q + (int64_t)bz * P.q_bs + (int64_t)(qh0 + g) * P.q_head_stride;
        const u32x4 w = *reinterpret_cast<const u32x4 *>(qrow + sub * 8);
        if constexpr (FUSED) {
            const u32x4 wp = *reinterpret_cast<const u32x4 *>(qrow + (sub ^ 8) * 8);
            rope8(w, wp, qf[g]);
        } else {
            unpack8(w, qf[g]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[g][e] *= P.scale_log2e;
    }

    float m[GT], l[GT], acc[GT][8];
    DUO_DT(1);
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        m[g] = kNegSentinel;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }

    // Main loop: the 16-token groups that lie wholly in segment A and inside the wave's range.  Their
    // addresses are a uniform base (SGPR pair, advanced per group) + four per-lane 32-bit byte offsets that
    // never change: no per-load address arithmetic, and straight-line code with a FIXED number of loads
    // in flight at every use — with a conditional prefetch hipcc has to wait with the smaller count, which
    // on the common path means waiting for the loads it has just issued.  The prefetch of the group after
    // the last one re-reads the last one (never consumed).  What is left — a partial group, segment B
    // rows — goes through the general clamped loader, one group at a time.
    const char *kA = reinterpret_cast<const char *>(C.a.k + kvh * C.a.head_stride);
    const char *vA = reinterpret_cast<const char *>(C.a.v + kvh * C.a.head_stride);
    uint32_t roff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) roff[u] = (uint32_t)(((4 * u + tg) * src.tsa + sub * 8) * 2);
    auto load_fast = [&](int t, u32x4 (&kb)[4], u32x4 (&vb)[4]) __attribute__((always_inline)) {
        // (readfirstlane: keeps the group base in SGPRs — loop strength reduction otherwise turns it into
        // a per-lane 64-bit pointer whose registers then collide with loads still in flight)
        gchar_t *kt = uniform_gptr(kA + (int64_t)t * src.tsa * 2), *vt = uniform_gptr(vA + (int64_t)t * src.tsa * 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // (empty asm: the 32-bit offset is re-materialised as such in THIS basic block, so instruction
            // selection sees sgpr base + zext(vgpr32) and emits the saddr form; hoisted out of the loop the
            // zero-extension becomes a 64-bit VGPR pair and every load pays a v_lshl_add_u64)
            // (in place, no copy: a temporary would be a fresh register — possibly one a load still in
            // flight is about to write, which costs a vmcnt(0))
            asm volatile("" : "+v"(roff[u]));
            kb[u] = ld16g<NT>(kt + roff[u]);
            vb[u] = ld16g<NT>(vt + roff[u]);
        }
    };

    int t_rest = w0;
    if constexpr (PREFETCH) {
        const int fast_end = min(w1, C.a.len);
        const int nfast = fast_end > w0 ? (fast_end - w0) / kTokPerIter : 0;
        if (nfast > 0) {
            const int t_last = w0 + (nfast - 1) * kTokPerIter;
            u32x4 k0[4], v0[4], k1[4], v1[4];
            load_fast(w0, k0, v0);
            // (sched_barrier: the loads stay ahead of the arithmetic on the other buffer — the scheduler
            // would otherwise sink them next to their first use, one iteration later.)
            for (int t = w0;; t += 2 * kTokPerIter) {
                load_fast(min(t + kTokPerIter, t_last), k1, v1);
                __builtin_amdgcn_sched_barrier(0);
                if (P.dbg & 64u) l[0] += __uint_as_float((k0[0].x ^ k0[1].y ^ k0[2].z ^ k0[3].w ^ v0[0].x ^ v0[1].y ^ v0[2].z ^ v0[3].w) & 1u);
                else consume_rows<GT, true>(k0, v0, t, tg, w1, qf, m, l, acc);
                if (t >= t_last) break;
                load_fast(min(t + 2 * kTokPerIter, t_last), k0, v0);
                __builtin_amdgcn_sched_barrier(0);
                if (P.dbg & 64u) l[0] += __uint_as_float((k1[0].x ^ k1[1].y ^ k1[2].z ^ k1[3].w ^ v1[0].x ^ v1[1].y ^ v1[2].z ^ v1[3].w) & 1u);
                else consume_rows<GT, true>(k1, v1, t + kTokPerIter, tg, w1, qf, m, l, acc);
                if (t + kTokPerIter >= t_last) break;
            }
            t_rest = w0 + nfast * kTokPerIter;
        }
    }
    {
        u32x4 k0[4], v0[4];
        for (int t = t_rest; t < w1; t += kTokPerIter) {
            load_rows<NT>(src, t, tg, w1, k0, v0);
            consume_rows<GT>(k0, v0, t, tg, w1, qf, m, l, acc);
        }
    }

    DUO_DT(2);
    if constexpr (FUSED) {
        // ---- the new token: last split of the kv head, wave 0, token group 0 (one 16-lane DPP row) ----
        if (split == splits - 1 && wave == 0 && tg == 0) {
            const bf16_t *krow = C.b.k + (int64_t)kvh * C.b.head_stride;
            const bf16_t *vrow = C.b.v + (int64_t)kvh * C.b.head_stride;
            const u32x4 kw = *reinterpret_cast<const u32x4 *>(krow + sub * 8);
            const u32x4 kp = *reinterpret_cast<const u32x4 *>(krow + (sub ^ 8) * 8);
            const u32x4 vw = *reinterpret_cast<const u32x4 *>(vrow + sub * 8);
            float kf[8], vf[8];
            rope8(kw, kp, kf);
            unpack8(vw, vf);
            u32x4 kr;   // rotated slice, bf16
            kr.x = (__float_as_uint(kf[0]) >> 16) | (__float_as_uint(kf[1]) & 0xffff0000u);
            kr.y = (__float_as_uint(kf[2]) >> 16) | (__float_as_uint(kf[3]) & 0xffff0000u);
            kr.z = (__float_as_uint(kf[4]) >> 16) | (__float_as_uint(kf[5]) & 0xffff0000u);
            kr.w = (__float_as_uint(kf[6]) >> 16) | (__float_as_uint(kf[7]) & 0xffff0000u);
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                float d = qf[g][0] * kf[0];
#pragma unroll
                for (int e = 1; e < 8; ++e) d = fmaf(qf[g][e], kf[e], d);
                const float sc_ = row16_allreduce_sum(d);
                const float mn = fmaxf(m[g], sc_);
                const float alpha = fast_exp2(m[g] - mn);
                const float p_ = fast_exp2(sc_ - mn);
                l[g] = fmaf(l[g], alpha, p_);
                m[g] = mn;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p_, vf[e], acc[g][e] * alpha);
            }
            // retrieval heads: the rotated key and the value join the pool (row app_row is outside
            // every scan range of this launch)
            if (ci == 0 && blockIdx.y == 0) {
                const int64_t po = (int64_t)bz * P.app_bs + (int64_t)app_row * P.app_ts + (int64_t)kvh * P.app_hs + sub * 8;
                *reinterpret_cast<u32x4 *>(P.app_k + po) = kr;
                *reinterpret_cast<u32x4 *>(P.app_v + po) = vw;
            }
        }
    }

#include "duo_decode_tail.inc"
}

// ------------------------------------------------------------------------------------------------------------------
// duo_decode_scan_kernel — the same split-KV scan behind a SHORT prologue (round 3; duo_decode_split_kernel above is
// kept as the general fallback and as the same-box A/B, debug bit 9).
//
// What the anatomy of the launch showed (profiles/r2_decode_wgs.md): all 256 workgroups run their prologue at the
// same time, so HBM idles for its whole length, and the prologue was a chain of DEPENDENT memory round trips of
// ~1 us each — kernel arguments (scalar loads, the class descriptor picked through a computed address) -> q rows ->
// first K/V rows.  Here
//   * everything the retrieval class needs to form its q and K/V addresses is the first 14 dwords of the argument
//     list (plain scalars, one s_load batch, no class select through a computed address); streaming-class workgroups,
//     a handful per launch and short, read their descriptor from the by-value parameter block as before.  (Preloading
//     those 14 dwords into SGPRs at wave launch — -mllvm -amdgpu-kernarg-preload-count=14 — was built and measured:
//     1.582 / 1.590 vs 1.589 / 1.589 ms per token, no difference, so the build does not use it.)
//   * the loads go out back to back in the order they are consumed — RoPE frequencies, q rows, then the FIRST K/V
//     group (clamped per-lane addresses, so it is unconditional and the counted waits stay exact), then the new
//     token's rows — and the RoPE factors / q rotation are computed underneath them: one memory round trip instead
//     of three before the first FMA;
//   * the new token's k/v rows sit in registers through the scan, so the epilogue of the last split has no load.
// pack0 = splits of the retrieval class (bits 0-11) | q heads per kv head (12-17) | first q head of the class (18-31);
// pack1 = q head stride in elements (0-15) | rows of the retrieval class's segment B (16-31).
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t cvt_pk_bf16_rne(float lo, float hi) {
    f32x2_t v = {lo, hi};
    hw_bf16x2_t r = __builtin_convertvector(v, hw_bf16x2_t);   // v_cvt_pk_bf16_f32: round to nearest even
    return *reinterpret_cast<uint32_t *>(&r);
}

template <int GT, bool NT, bool FUSED>
__global__ __launch_bounds__(256) void duo_decode_scan_kernel(const bf16_t *__restrict__ q, const bf16_t *k0,
                                                              const bf16_t *v0, const int32_t *dev_state, int32_t len0,
                                                              uint32_t pack0, int32_t nblk_full, uint32_t hs0,
                                                              uint32_t ts0, uint32_t pack1, const DecodeParams P,
                                                              const CompressParams CP) {
    DUO_DT(0);
    DUO_DRT(8);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: scalar loop control
    const int sub = lane & 15;   // which 8-dim slice of the 128-dim row
    const int tg = lane >> 4;    // which of the 4 rows a wave-load covers

    // ---- class description: retrieval class from the preloaded SGPRs, streaming class from the kernarg segment -----
    int b = blockIdx.x;
    const int ci = b < nblk_full ? 0 : 1;
    const int group = (int)((pack0 >> 12) & 63u);
    const bf16_t *ak = k0, *av = v0;
    int64_t a_ts = ts0, a_hs = hs0;
    int a_len = len0, b_len = (int)(pack1 >> 16), splits = (int)(pack0 & 4095u), qoff = (int)(pack0 >> 18);
    int app_row = 0, st_full = 0, st_str = 0;
    float pos = 0.f;
    bool have_state = false;
    if constexpr (FUSED) {
        if (dev_state) {   // captured step: lengths / position live in device memory (uniform scalar loads)
            st_full = dev_state[0];
            st_str = dev_state[1];
            pos = (float)dev_state[2] + P.pos_delta;
            app_row = st_full;
            have_state = true;
        }
    }
    if (ci) {
        asm volatile("" ::: "memory");   // a real branch: retrieval workgroups must not wait for these scalar loads
        b -= nblk_full;
        ak = P.cls[1].a.k;
        av = P.cls[1].a.v;
        a_ts = P.cls[1].a.token_stride;
        a_hs = P.cls[1].a.head_stride;
        a_len = P.cls[1].a.len;
        b_len = P.cls[1].b.len;
        splits = P.splits[1];
        qoff = P.cls[1].q_head_offset;
    }
    if constexpr (FUSED) {
        if (have_state) a_len = ci ? st_str : st_full;
    }
    const int bz = blockIdx.z;      // batch row (batched launches): row 0 needs nothing from the parameter block here
    if (bz) {
        asm volatile("" ::: "memory");
        q += (int64_t)bz * P.q_bs;
        ak += (int64_t)bz * (ci ? P.cls[1].a.batch_stride : P.cls[0].a.batch_stride);
        av += (int64_t)bz * (ci ? P.cls[1].a.batch_stride : P.cls[0].a.batch_stride);
    }
    bf16_t *const out_row = P.out + (int64_t)bz * P.out_bs;
    float *const ws_ml_row = P.ws_ml + (int64_t)bz * P.ws_row_floats, *const ws_acc_row = P.ws_acc + (int64_t)bz * P.ws_row_floats;
    const int kvh = b / splits;
    const int split = b - kvh * splits;
    const int qh0 = qoff + kvh * group + blockIdx.y * GT;

    // balanced static partition (as duo_decode_split_kernel): the head's 64-token units dealt evenly to the splits
    const int L = FUSED ? a_len : a_len + b_len;
    const int units = (L + 63) >> 6;
    int u0, u1;
    if (P.odd_dw > 0 && splits > 1 && gridDim.y == 1) {
        // XCD-weighted deal (experiment, profiles/r3_decode.md): a workgroup on an odd XCD (block id odd: XCD = id % 8)
        // takes (64 - odd_dw) / 64 of an even one's share.  Same closed form in every workgroup of the head, so the
        // ranges tile the head exactly; the results stay a fixed function of the launch shape.
        const int bfirst = (int)blockIdx.x - split;                  // block id of this head's split 0
        const int wt0 = (bfirst & 1) ? 64 - P.odd_dw : 64;
        const int pair = 128 - P.odd_dw;
        auto cum = [&](int s_) { return (s_ >> 1) * pair + ((s_ & 1) ? wt0 : 0); };
        const float per = (float)units / (float)cum(splits);
        u0 = split == 0 ? 0 : (int)((float)cum(split) * per);
        u1 = split + 1 == splits ? units : (int)((float)cum(split + 1) * per);
    } else {
        const int uq = units / splits, ur = units - uq * splits;
        u0 = split * uq + min(split, ur);
        u1 = u0 + uq + (split < ur ? 1 : 0);
    }
    const int c0 = u0 << 6;
    const int c1 = min(u1 << 6, L);
    const int per_wave = (((c1 - c0 + 3) >> 2) + 15) & ~15;  // quarter of the chunk, multiple of 16
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + wave * per_wave);
    const int w1 = __builtin_amdgcn_readfirstlane(min(w0 + per_wave, c1));

    const char *kA = reinterpret_cast<const char *>(ak + (int64_t)kvh * a_hs);
    const char *vA = reinterpret_cast<const char *>(av + (int64_t)kvh * a_hs);
    uint32_t roff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) roff[u] = ((uint32_t)(4 * u + tg) * (uint32_t)a_ts + (uint32_t)sub * 8u) * 2u;

    // ================= loads, in the order they are consumed =======================================================
    // (a) RoPE frequencies of this lane's 8 dims (FUSED): dims 8*sub+e pair with dims (8*sub+e) ^ 64; both slices of a
    //     pair use frequency index (8*sub+e) & 63.  Vector loads from the kernarg segment (per-lane index).
    f32x4 fr0 = {0.f, 0.f, 0.f, 0.f}, fr1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FUSED) {
        const float *fp = &P.inv_freq[(sub & 7) << 3];
        fr0 = *reinterpret_cast<const f32x4 *>(fp);
        fr1 = *reinterpret_cast<const f32x4 *>(fp + 4);
    }
    // (b) q rows (own slice, and the rotation partner's when q arrives un-rotated)
    const uint32_t qhs = pack1 & 0xffffu;
    u32x4 qw[GT], qp[GT];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        const bf16_t *qrow = q + (int64_t)(qh0 + g) * qhs;
        qw[g] = *reinterpret_cast<const u32x4 *>(qrow + sub * 8);
        if constexpr (FUSED) qp[g] = *reinterpret_cast<const u32x4 *>(qrow + (sub ^ 8) * 8);
    }
    // (c) the first K/V group of this wave: per-lane row index clamped into segment A (which always has a readable
    //     row 0: an empty segment A is given the base of segment B by the launcher), so the eight loads are
    //     unconditional.  When the wave has a full first group (nfast > 0) no lane is clamped and the registers hold
    //     exactly what load_fast(t_first) would have fetched.
    // (Visiting the wave's groups in a per-workgroup ROTATED order — so that the 256 workgroups do not walk their equally
    // long chunks in lockstep — was built and measured: 1.595-1.599 vs 1.582-1.584 ms per token, i.e. slower; the plain
    // ascending order stays.  profiles/r3_decode.md)
    const int fast_end = min(w1, a_len);
    const int nfast = fast_end > w0 ? (fast_end - w0) / kTokPerIter : 0;
    const int t_first = w0;
    u32x4 k0r[4], v0r[4], k1r[4], v1r[4];
    {
        const int last = max(a_len - 1, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tok = min(t_first + 4 * u + tg, last);
            const int64_t off = ((int64_t)tok * a_ts + sub * 8) * 2;
            k0r[u] = ld16<NT>(reinterpret_cast<const bf16_t *>(kA + off));
            v0r[u] = ld16<NT>(reinterpret_cast<const bf16_t *>(vA + off));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    DUO_DT(4);

    // ================= everything below runs underneath those loads ================================================
    DuoSegDev Bseg = duo_select(P.cls[0].b, P.cls[1].b, ci != 0);
    Bseg.k += (int64_t)bz * Bseg.batch_stride;
    Bseg.v += (int64_t)bz * Bseg.batch_stride;
    if constexpr (FUSED) {
        if (!have_state) {
            app_row = P.app_row;
            pos = P.pos;
        }
    }
    // (d) the new token's rows (FUSED): k slice, its rotation partner, v slice — every wave loads them (L2 hits), only
    //     the last split's wave 0 uses them after the scan
    u32x4 nkw = {0u, 0u, 0u, 0u}, nkp = {0u, 0u, 0u, 0u}, nvw = {0u, 0u, 0u, 0u};
    if constexpr (FUSED) {
        const bf16_t *krow = Bseg.k + (int64_t)kvh * Bseg.head_stride;
        const bf16_t *vrow = Bseg.v + (int64_t)kvh * Bseg.head_stride;
        nkw = *reinterpret_cast<const u32x4 *>(krow + sub * 8);
        nkp = *reinterpret_cast<const u32x4 *>(krow + (sub ^ 8) * 8);
        nvw = *reinterpret_cast<const u32x4 *>(vrow + sub * 8);
        __builtin_amdgcn_sched_barrier(0);
    }
    float cs[8], sn[8];
    if constexpr (FUSED) {
        const float fr[8] = {fr0.x, fr0.y, fr0.z, fr0.w, fr1.x, fr1.y, fr1.z, fr1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) sincos_rev(pos * fr[e], sn[e], cs[e]);
    }
    DUO_DT(5);
    // x: own slice, y: partner slice -> rotated own slice (first half: x*c - y*s, second half: x*c + y*s), rounded to
    // bf16 exactly like the standalone RoPE kernel does before the attention reads it; `packed` = the bf16 bits
    auto rope8 = [&](const u32x4 &own, const u32x4 &partner, float (&o)[8], u32x4 &packed) {
        float x[8], y[8];
        unpack8(own, x);
        unpack8(partner, y);
        const float sgn = sub < 8 ? -1.f : 1.f;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            w[e >> 1] = cvt_pk_bf16_rne(x[e] * cs[e] + sgn * y[e] * sn[e], x[e + 1] * cs[e + 1] + sgn * y[e + 1] * sn[e + 1]);
            o[e] = bf16_lo(w[e >> 1]);
            o[e + 1] = bf16_hi(w[e >> 1]);
        }
        packed.x = w[0]; packed.y = w[1]; packed.z = w[2]; packed.w = w[3];
    };

    float qf[GT][8];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        if constexpr (FUSED) {
            u32x4 unused;
            rope8(qw[g], qp[g], qf[g], unused);
        } else {
            unpack8(qw[g], qf[g]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[g][e] *= P.scale_log2e;
    }

    float m[GT], l[GT], acc[GT][8];
    DUO_DT(1);
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        m[g] = kNegSentinel;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }

    // ---- main loop: as duo_decode_split_kernel (uniform SGPR base + constant per-lane offsets, a fixed number of
    //      loads in flight at every use); its first group is already on the way -------------------------------------
    auto load_fast = [&](int t, u32x4 (&kb)[4], u32x4 (&vb)[4]) __attribute__((always_inline)) {
        gchar_t *kt = uniform_gptr(kA + (int64_t)t * a_ts * 2), *vt = uniform_gptr(vA + (int64_t)t * a_ts * 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(roff[u]));
            kb[u] = ld16g<NT>(kt + roff[u]);
            vb[u] = ld16g<NT>(vt + roff[u]);
        }
    };
    int t_rest = w0;
    if (nfast > 0) {
        const int t_last = w0 + (nfast - 1) * kTokPerIter;
        for (int t = w0;; t += 2 * kTokPerIter) {
            load_fast(min(t + kTokPerIter, t_last), k1r, v1r);      // (behind the last group: re-reads it, never consumed)
            __builtin_amdgcn_sched_barrier(0);
            if (P.dbg & 64u) l[0] += __uint_as_float((k0r[0].x ^ k0r[1].y ^ k0r[2].z ^ k0r[3].w ^ v0r[0].x ^ v0r[1].y ^ v0r[2].z ^ v0r[3].w) & 1u);
            else consume_rows<GT, true>(k0r, v0r, t, tg, w1, qf, m, l, acc);
            if (t >= t_last) break;
            load_fast(min(t + 2 * kTokPerIter, t_last), k0r, v0r);
            __builtin_amdgcn_sched_barrier(0);
            if (P.dbg & 64u) l[0] += __uint_as_float((k1r[0].x ^ k1r[1].y ^ k1r[2].z ^ k1r[3].w ^ v1r[0].x ^ v1r[1].y ^ v1r[2].z ^ v1r[3].w) & 1u);
            else consume_rows<GT, true>(k1r, v1r, t + kTokPerIter, tg, w1, qf, m, l, acc);
            if (t + kTokPerIter >= t_last) break;
        }
        t_rest = w0 + nfast * kTokPerIter;
    }
    {
        // what is left: a partial group of segment A, segment B rows (non-fused form) — general clamped loader
        RowSrc src;
        src.ka = reinterpret_cast<const bf16_t *>(kA) + sub * 8;
        src.va = reinterpret_cast<const bf16_t *>(vA) + sub * 8;
        src.kb = Bseg.k + (int64_t)kvh * Bseg.head_stride + sub * 8;
        src.vb = Bseg.v + (int64_t)kvh * Bseg.head_stride + sub * 8;
        src.tsa = a_ts;
        src.tsb = Bseg.token_stride;
        src.lenA = a_len;
        u32x4 kt[4], vt[4];
        for (int t = t_rest; t < w1; t += kTokPerIter) {
            load_rows<NT>(src, t, tg, w1, kt, vt);
            consume_rows<GT>(kt, vt, t, tg, w1, qf, m, l, acc);
        }
    }

    DUO_DT(2);
    if constexpr (FUSED) {
        // ---- the new token: last split of the kv head, wave 0, token group 0 (one 16-lane DPP row) ----
        if (split == splits - 1 && wave == 0 && tg == 0) {
            float kf[8], vf[8];
            u32x4 kr;   // rotated slice, bf16
            rope8(nkw, nkp, kf, kr);
            unpack8(nvw, vf);
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                float d = qf[g][0] * kf[0];
#pragma unroll
                for (int e = 1; e < 8; ++e) d = fmaf(qf[g][e], kf[e], d);
                const float sc_ = row16_allreduce_sum(d);
                const float mn = fmaxf(m[g], sc_);
                const float alpha = fast_exp2(m[g] - mn);
                const float p_ = fast_exp2(sc_ - mn);
                l[g] = fmaf(l[g], alpha, p_);
                m[g] = mn;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p_, vf[e], acc[g][e] * alpha);
            }
            // retrieval heads: the rotated key and the value join the pool (row app_row is outside every scan range
            // of this launch)
            if (ci == 0 && blockIdx.y == 0) {
                const int64_t po = (int64_t)bz * P.app_bs + (int64_t)app_row * P.app_ts + (int64_t)kvh * P.app_hs + sub * 8;
                *reinterpret_cast<u32x4 *>(P.app_k + po) = kr;
                *reinterpret_cast<u32x4 *>(P.app_v + po) = nvw;
            }
        }
    }
#include "duo_decode_tail.inc"
}

// Epilogue launch of a decode step: blocks [0, n_merge) merge the split-KV partials of one q-head quarter
// each (n_merge = 4 x q heads with more than one split); blocks [n_merge, n_merge + n_compress) run the streaming-pool sink+recent update of one
// (head, K|V) each (absent when the caller updates the pool separately).
__global__ __launch_bounds__(256) void duo_decode_post_kernel(const MergeParams M, int n_merge,
                                                             const CompressParams C) {
    if ((int)blockIdx.x < n_merge) duo_decode_merge_block(M, blockIdx.x, blockIdx.y);
    else duo_stream_compress_block(C, blockIdx.x - n_merge, blockIdx.y);
}

}  // namespace

extern "C" int64_t duo_attn_decode_workspace_bytes(int32_t n_q_heads, int32_t max_splits) {
    if (n_q_heads <= 0 || max_splits <= 0) return 0;
    return (int64_t)n_q_heads * max_splits * (DUO_HEAD_DIM + 2) * (int64_t)sizeof(float);
}

static int decode_target_wgs() {
    // workgroups the split kernel aims for across both classes (tuning knob, read once)
    static int v = [] {
        const char *e = getenv("DUO_DECODE_TARGET_WGS");
        const int x = e ? atoi(e) : 0;
        return x > 0 ? x : 256;
    }();
    return v;
}

// Split policy: one resident round of ONE workgroup per CU (256 in total); each kv head's 64-token units are dealt
// evenly to its splits (balanced partition in the kernel).  Never more splits than units or than the workspace
// holds.  Measured at 131072 context, 32 layers (profiles/r2_decode_wgs.md): 256 workgroups 1.46-1.49 ms per step of
// scans, 512 (two per CU, the round-1 choice) 1.54 ms, 384 / 640 (not a multiple of the CU count) 1.63-1.66 ms —
// fewer, longer workgroups amortise the per-workgroup prologue / epilogue and halve the partials to merge.
//
// The split count is a function of the length's BUCKET (64-token units rounded up to a power of two), not of the length:
// every length inside a bucket gets the same grid.  That is what lets a captured decode step (device-side lengths,
// duo_decode_layer_dev_bf16) stand for the eager one bit for bit while the cache grows — the kernel deals the CURRENT units
// to the captured workgroups, and the eager plan of any length in the bucket is that same grid — and it tells the graph's
// owner exactly when a re-capture pays: when duo_decode_plan_bucket() of the length changes (duo_attn/graph.py does).
static int plan_bucket_units(int L) {
    int units = (L + 63) / 64, b = 1;
    while (b < units) b <<= 1;
    return b;
}
extern "C" int32_t duo_decode_plan_bucket(int32_t n_tokens) { return n_tokens <= 0 ? 0 : plan_bucket_units(n_tokens); }

static void choose_splits(int n_kv_heads, int L, int max_splits, int budget_wgs, int &splits) {
    if (n_kv_heads <= 0 || L <= 0) {
        splits = 0;
        return;
    }
    const int bucket = plan_bucket_units(L);        // units <= bucket < 2 * units
    int s = budget_wgs / n_kv_heads;
    // keep a few units per workgroup so its epilogue stays small: bucket / 8 (256-512 tokens per workgroup; ADVICE r5 asked
    // for bucket / 4 at short contexts — DUO_DECODE_SPLIT_DIV, measured in profiles/r6_decode_short.md)
    static const int div = [] { const char *e = getenv("DUO_DECODE_SPLIT_DIV"); const int x = e ? atoi(e) : 0; return x >= 1 && x <= 64 ? x : 8; }();
    s = std::min(s, std::max(1, bucket / div));
    s = std::max(1, std::min(s, max_splits));
    splits = s;
}

namespace {
struct DecodePlan {
    DecodeParams P;
    MergeParams M;
    int n_merge;
    int nblk;
    int gt;
    int n_batch;
};
}  // namespace

// validates the two head classes, sizes the splits and fills the kernel parameter blocks
static int decode_plan(const void *q, int64_t q_head_stride, void *out, int64_t out_head_stride,
                       int32_t group, const duo_head_class *full, const duo_head_class *stream_cls,
                       float scale, void *workspace, int64_t workspace_bytes, DecodePlan &D,
                       int32_t n_batch = 1, int64_t q_batch_stride = 0, int64_t out_batch_stride = 0) {
    DecodeParams &P = D.P;
    if (n_batch < 1 || n_batch > 65535 || (n_batch > 1 && ((q_batch_stride & 7) || (out_batch_stride & 3)))) return DUO_EINVAL;
    D.n_batch = n_batch;
    P.q_bs = q_batch_stride;
    P.out_bs = out_batch_stride;
    P.app_bs = 0;
    P.ws_row_floats = 0;
    P.q = (const bf16_t *)q;
    P.q_head_stride = q_head_stride;
    P.out = (bf16_t *)out;
    P.out_head_stride = out_head_stride;
    P.cls[0] = duo_class_dev(full);
    P.cls[1] = duo_class_dev(stream_cls);
    P.group = group;
    P.dbg = duo_get_debug_flags();
    static const int odd_dw = [] { const char *e = getenv("DUO_DECODE_ODD_XCD_DW"); const int x = e ? atoi(e) : 0; return x > 0 && x < 32 ? x : 0; }();
    P.odd_dw = odd_dw;
    P.fused = 0;
    P.one_launch = 0;
    P.tickets = nullptr;
    P.dev_state = nullptr;
    P.pos_delta = 0.f;
    P.scale_log2e = scale * 1.4426950408889634f;
    const int n_q_heads = (P.cls[0].n_kv_heads + P.cls[1].n_kv_heads) * group;
    D.nblk = 0;
    D.n_merge = 0;
    if (n_q_heads <= 0) return 0;
    if ((q_head_stride & 7) || (out_head_stride & 3)) return DUO_EINVAL;   // 16-byte q loads, 8-byte out stores
    for (int c = 0; c < 2; ++c) {
        const DuoClassDev &C = P.cls[c];
        if (C.n_kv_heads <= 0) { P.cls[c].n_kv_heads = 0; continue; }
        if (C.a.len < 0 || C.b.len < 0 || C.a.len + C.b.len <= 0) return DUO_EINVAL;
        if ((C.a.len > 0 && (!C.a.k || !C.a.v)) || (C.b.len > 0 && (!C.b.k || !C.b.v))) return DUO_EINVAL;
        if (C.a.len == 0) { P.cls[c].a = P.cls[c].b; P.cls[c].a.len = 0; }  // valid base for clamped loads
        if (C.b.len == 0) { P.cls[c].b = P.cls[c].a; P.cls[c].b.len = 0; }
    }
    // workspace capacity -> max splits per head
    const int64_t per_split = (int64_t)n_q_heads * (DUO_HEAD_DIM + 2) * (int64_t)sizeof(float) * n_batch;   // every batch row has its own partials
    const int max_splits = workspace ? (int)std::min<int64_t>(workspace_bytes / per_split, 1024) : 0;
    {
        const int target = std::max(1, decode_target_wgs() / n_batch);     // one resident round over all rows
        const int ms = max_splits > 0 ? max_splits : 1;
        const int Ls = P.cls[1].a.len + P.cls[1].b.len, Lf = P.cls[0].a.len + P.cls[0].b.len;
        // the streaming class is a few hundred rows per head: one workgroup each unless it is alone
        choose_splits(P.cls[1].n_kv_heads, Ls, ms, P.cls[0].n_kv_heads > 0 ? P.cls[1].n_kv_heads : target, P.splits[1]);
        const int used = P.cls[1].n_kv_heads * P.splits[1];
        choose_splits(P.cls[0].n_kv_heads, Lf, ms, std::max(target - used, P.cls[0].n_kv_heads), P.splits[0]);
        P.chunk[0] = P.chunk[1] = 0;
    }
    const int need = std::max(P.splits[0], P.splits[1]);
    if (need > 1 && max_splits < need) return DUO_EWORKSPC;
    P.max_splits = need > 1 ? need : 1;
    P.ws_ml = (float *)workspace;
    P.ws_acc = P.ws_ml ? P.ws_ml + (int64_t)n_q_heads * P.max_splits * 2 : nullptr;
    P.ws_bytes = (uint32_t)std::min<int64_t>((int64_t)n_q_heads * P.max_splits * (DUO_HEAD_DIM + 2) * (int64_t)sizeof(float),
                                             0xffffffffll);
    P.nblk_full = P.cls[0].n_kv_heads * P.splits[0];
    D.nblk = P.nblk_full + P.cls[1].n_kv_heads * P.splits[1];
    D.gt = (group % 4 == 0) ? 4 : (group % 2 == 0) ? 2 : 1;
    P.ws_row_floats = (int64_t)n_q_heads * P.max_splits * (DUO_HEAD_DIM + 2);

    MergeParams &M = D.M;
    M.ws_row_floats = P.ws_row_floats;
    M.out_bs = out_batch_stride;
    M.ws_ml = P.ws_ml;
    M.ws_acc = P.ws_acc;
    M.out = P.out;
    M.out_head_stride = out_head_stride;
    M.max_splits = P.max_splits;
    for (int c = 0; c < 2; ++c) {
        M.qh_begin[c] = P.cls[c].q_head_offset;
        M.qh_end[c] = P.cls[c].q_head_offset + P.cls[c].n_kv_heads * group;
        M.splits[c] = P.splits[c];
        if (P.splits[c] > 1) D.n_merge += P.cls[c].n_kv_heads * group;
    }
    return 0;
}

// The short-prologue scan kernel takes the retrieval class's addressing in 14 preloaded dwords; it applies when those
// fields fit their packed widths (always, for the pools and activations this library's callers hand over).
static bool decode_scan_eligible(const DecodeParams &P) {
    const DuoClassDev &F = P.cls[0];
    auto u32ok = [](int64_t x) { return x >= 0 && x <= 0xffffffffll; };
    if (P.group <= 0 || P.group > 63 || P.q_head_stride < 0 || P.q_head_stride > 0xffff) return false;
    if (F.n_kv_heads > 0) {
        if (P.splits[0] <= 0 || P.splits[0] > 4095 || F.q_head_offset < 0 || F.q_head_offset > 16383) return false;
        if (!u32ok(F.a.token_stride) || !u32ok(F.a.head_stride) || F.b.len < 0 || F.b.len > 0xffff) return false;
    }
    const DuoClassDev &S = P.cls[1];
    if (S.n_kv_heads > 0 && (S.a.token_stride < 0 || S.a.head_stride < 0)) return false;
    return true;
}

template <bool FUSED>
static int decode_launch_split(const DecodePlan &D, const CompressParams &CP, hipStream_t st) {
    if (D.nblk <= 0) return 0;
    const DecodeParams &P = D.P;
    dim3 grid(D.nblk, P.group / D.gt, D.n_batch), block(256);
    const uint32_t fl = duo_get_debug_flags();
    const bool nt = !(fl & 4u);        // debug bit 2: plain (temporal) K/V loads
    const bool pf = !(fl & 8u);        // debug bit 3: no register prefetch of the next 16 tokens
    // debug bit 9: stay on duo_decode_split_kernel (the long-prologue form: same-box A/B of the two scans)
    if (pf && !(fl & 512u) && decode_scan_eligible(P)) {
        const DuoClassDev &F = P.cls[0];
        const bool hasF = F.n_kv_heads > 0;
        const uint32_t pack0 = hasF ? ((uint32_t)P.splits[0] | ((uint32_t)P.group << 12) | ((uint32_t)F.q_head_offset << 18))
                                    : (1u | ((uint32_t)P.group << 12));
        const uint32_t pack1 = (uint32_t)P.q_head_stride | (hasF ? (uint32_t)F.b.len << 16 : 0u);
        const uint32_t hs0 = hasF ? (uint32_t)F.a.head_stride : 0u, ts0 = hasF ? (uint32_t)F.a.token_stride : 0u;
        const int32_t len0 = hasF ? F.a.len : 0;
#define DUO_LAUNCH_SCAN(GT_)                                                                                          \
    do {                                                                                                              \
        if (nt) hipLaunchKernelGGL((duo_decode_scan_kernel<GT_, true, FUSED>), grid, block, 0, st, P.q, F.a.k, F.a.v,     \
                                   P.dev_state, len0, pack0, P.nblk_full, hs0, ts0, pack1, P, CP);                       \
        else hipLaunchKernelGGL((duo_decode_scan_kernel<GT_, false, FUSED>), grid, block, 0, st, P.q, F.a.k, F.a.v,       \
                                P.dev_state, len0, pack0, P.nblk_full, hs0, ts0, pack1, P, CP);                          \
    } while (0)
        if (D.gt == 4) DUO_LAUNCH_SCAN(4);
        else if (D.gt == 2) DUO_LAUNCH_SCAN(2);
        else DUO_LAUNCH_SCAN(1);
#undef DUO_LAUNCH_SCAN
        DUO_HIP_CHECK_LAUNCH();
        return 0;
    }
#define DUO_LAUNCH_DECODE(GT_)                                                                                      \
    do {                                                                                                            \
        if (nt && pf) hipLaunchKernelGGL((duo_decode_split_kernel<GT_, true, true, FUSED>), grid, block, 0, st, P, CP);   \
        else if (nt) hipLaunchKernelGGL((duo_decode_split_kernel<GT_, true, false, FUSED>), grid, block, 0, st, P, CP);   \
        else if (pf) hipLaunchKernelGGL((duo_decode_split_kernel<GT_, false, true, FUSED>), grid, block, 0, st, P, CP);   \
        else hipLaunchKernelGGL((duo_decode_split_kernel<GT_, false, false, FUSED>), grid, block, 0, st, P, CP);          \
    } while (0)
    if (D.gt == 4) DUO_LAUNCH_DECODE(4);
    else if (D.gt == 2) DUO_LAUNCH_DECODE(2);
    else DUO_LAUNCH_DECODE(1);
#undef DUO_LAUNCH_DECODE
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

static int attn_decode_impl(const void *q, int64_t q_batch_stride, int64_t q_head_stride, void *out,
                            int64_t out_batch_stride, int64_t out_head_stride, int32_t n_batch, int32_t group,
                            const duo_head_class *full, const duo_head_class *stream_cls, float scale, int32_t head_dim,
                            void *workspace, int64_t workspace_bytes, void *stream);
extern "C" int duo_attn_decode_bf16(const void *q, int64_t q_head_stride, void *out,
                                    int64_t out_head_stride, int32_t group,
                                    const duo_head_class *full, const duo_head_class *stream_cls,
                                    float scale, int32_t head_dim, void *workspace,
                                    int64_t workspace_bytes, void *stream) {
    return attn_decode_impl(q, 0, q_head_stride, out, 0, out_head_stride, 1, group, full, stream_cls, scale, head_dim,
                            workspace, workspace_bytes, stream);
}
// batched: q / out [B, n_q_heads, 128]; the segments of `full` / `stream_cls` carry their batch strides; every row at
// the same lengths (the reference's pools and counters, static_kv_cache.py:44-45,60-99); grid.z = batch row
extern "C" int duo_attn_decode_batched_bf16(const void *q, int64_t q_batch_stride, int64_t q_head_stride, void *out,
                                            int64_t out_batch_stride, int64_t out_head_stride, int32_t n_batch,
                                            int32_t group, const duo_head_class *full, const duo_head_class *stream_cls,
                                            float scale, int32_t head_dim, void *workspace, int64_t workspace_bytes,
                                            void *stream) {
    if (n_batch <= 0) return n_batch == 0 ? 0 : DUO_EINVAL;
    return attn_decode_impl(q, q_batch_stride, q_head_stride, out, out_batch_stride, out_head_stride, n_batch, group, full,
                            stream_cls, scale, head_dim, workspace, workspace_bytes, stream);
}
static int attn_decode_impl(const void *q, int64_t q_batch_stride, int64_t q_head_stride, void *out,
                            int64_t out_batch_stride, int64_t out_head_stride, int32_t n_batch, int32_t group,
                            const duo_head_class *full, const duo_head_class *stream_cls, float scale, int32_t head_dim,
                            void *workspace, int64_t workspace_bytes, void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (q == nullptr || out == nullptr || group <= 0) return DUO_EINVAL;
    DecodePlan D;
    int rc = decode_plan(q, q_head_stride, out, out_head_stride, group, full, stream_cls, scale, workspace,
                         workspace_bytes, D, n_batch, q_batch_stride, out_batch_stride);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    rc = decode_launch_split<false>(D, CompressParams{}, st);
    if (rc) return rc;
    // debug flag bit 1: leave the partials unmerged (profiling the split kernel alone)
    if (D.n_merge > 0 && !(duo_get_debug_flags() & 2u)) {
        CompressParams none{};
        hipLaunchKernelGGL(duo_decode_post_kernel, dim3(4 * D.n_merge, D.n_batch), dim3(256), 0, st, D.M, 4 * D.n_merge, none);
        DUO_HIP_CHECK_LAUNCH();
    }
    return 0;
}

// One decode step of one layer of the static dual-cache path in two launches: the split-KV scan of
// both head classes with RoPE of q / the new key row and the retrieval-pool append folded in, then
// merge + streaming-pool update.  See include/duo_attn_hip.h.
static int decode_layer_impl(const duo_decode_layer_args *a, int32_t *new_stream_len, const int32_t *dev_state,
                             void *workspace, int64_t workspace_bytes, void *tickets, void *stream,
                             const duo_decode_batch *B = nullptr, int64_t pos_delta = 0) {
    if (!a) return DUO_EINVAL;
    if (a->head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    const int n_batch = B ? B->n_batch : 1;
    if (n_batch < 1) return n_batch == 0 ? 0 : DUO_EINVAL;
    if (B && n_batch > 1) {
        if (((B->q_batch_stride | B->kv_batch_stride | B->out_batch_stride | B->full_batch_stride | B->str_batch_stride) & 7) != 0)
            return DUO_EINVAL;
        bool same = true;
        for (int b = 0; B->pos && b < n_batch; ++b) same = same && B->pos[b] == B->pos[0];
        if (!same) {
            // rows at different positions (left-padded batches): one launch pair per row — the lengths, and with them
            // the partition, are still common
            for (int b = 0; b < n_batch; ++b) {
                duo_decode_layer_args r = *a;
                r.q = (bf16_t *)a->q + b * B->q_batch_stride;
                r.k = (bf16_t *)a->k + b * B->kv_batch_stride;
                r.v = (const bf16_t *)a->v + b * B->kv_batch_stride;
                r.out = (bf16_t *)a->out + b * B->out_batch_stride;
                if (a->full_k) { r.full_k = (bf16_t *)a->full_k + b * B->full_batch_stride; r.full_v = (bf16_t *)a->full_v + b * B->full_batch_stride; }
                if (a->str_k) { r.str_k = (bf16_t *)a->str_k + b * B->str_batch_stride; r.str_v = (bf16_t *)a->str_v + b * B->str_batch_stride; }
                r.pos = B->pos[b];
                // (device-side lengths: the row runs at dev_state->pos plus its fixed offset from args->pos, the host's
                //  view of that counter)
                const int rc = decode_layer_impl(&r, new_stream_len, dev_state, workspace, workspace_bytes, nullptr, stream,
                                                 nullptr, dev_state ? B->pos[b] - a->pos : 0);
                if (rc) return rc;
            }
            return 0;
        }
    }
    const int64_t pos_all = (B && B->pos) ? B->pos[0] : a->pos;
    if (dev_state && B && B->pos && n_batch > 1) pos_delta = B->pos[0] - a->pos;      // all rows at one offset from the counter
    const bool batched = n_batch > 1;
    const int nf = a->n_full, nkv = a->n_kv_heads, ns = nkv - nf;
    if (!a->q || !a->k || !a->v || !a->out || nkv <= 0 || nf < 0 || ns < 0 || a->n_q_heads % nkv != 0)
        return DUO_EINVAL;
    if (a->rope_scale <= 0.f || a->rope_theta <= 0.f) return DUO_EINVAL;
    if (nf > 0 && (!a->full_k || !a->full_v || a->full_len < 0 || a->full_len + 1 > a->full_capacity)) return DUO_EINVAL;
    const int W = a->sink + a->recent;
    if (a->str_len < 0 || a->str_len > W) return DUO_EINVAL;
    if (ns > 0 && (!a->str_k || !a->str_v)) return DUO_EINVAL;
    if (((a->q_head_stride | a->kv_head_stride | a->out_head_stride | a->full_token_stride | a->full_head_stride |
          a->str_token_stride | a->str_head_stride) & 7) != 0)
        return DUO_EINVAL;
    const int group = a->n_q_heads / nkv;
    hipStream_t st = (hipStream_t)stream;

    // ---- launch 1: split-KV scan (+ RoPE, + retrieval append) -----------------------------------
    const bf16_t *kn = (const bf16_t *)a->k, *vn = (const bf16_t *)a->v;
    duo_head_class fc{}, sc{};
    fc.n_kv_heads = nf;
    fc.q_head_offset = 0;
    fc.segA = duo_kv_seg{a->full_k, a->full_v, a->full_token_stride, a->full_head_stride, a->full_len, 0};
    fc.segB = duo_kv_seg{kn, vn, 0, a->kv_head_stride, 1, 0};
    sc.n_kv_heads = ns;
    sc.q_head_offset = nf * group;
    sc.segA = duo_kv_seg{a->str_k, a->str_v, a->str_token_stride, a->str_head_stride, a->str_len, 0};
    sc.segB = duo_kv_seg{kn + (int64_t)nf * a->kv_head_stride, vn + (int64_t)nf * a->kv_head_stride, 0,
                         a->kv_head_stride, 1, 0};
    if (batched) {
        fc.segA.batch_stride = B->full_batch_stride;
        sc.segA.batch_stride = B->str_batch_stride;
        fc.segB.batch_stride = sc.segB.batch_stride = B->kv_batch_stride;
    }
    DecodePlan D;
    int rc = decode_plan(a->q, a->q_head_stride, a->out, a->out_head_stride, group, nf ? &fc : nullptr,
                         ns ? &sc : nullptr, a->scale, workspace, workspace_bytes, D, n_batch,
                         batched ? B->q_batch_stride : 0, batched ? B->out_batch_stride : 0);
    if (rc) return rc;
    float inv_freq[64];
    for (int i = 0; i < 64; ++i)
        inv_freq[i] = (float)(pow((double)a->rope_theta, -2.0 * i / 128.0) / (double)a->rope_scale);
    D.P.fused = 1;
    D.P.app_row = a->full_len;
    D.P.app_k = (bf16_t *)a->full_k;
    D.P.app_v = (bf16_t *)a->full_v;
    D.P.app_ts = a->full_token_stride;
    D.P.app_hs = a->full_head_stride;
    D.P.app_bs = batched ? B->full_batch_stride : 0;
    D.P.pos = (float)pos_all;
    D.P.dev_state = dev_state;
    D.P.pos_delta = (float)pos_delta;
    memcpy(D.P.inv_freq, inv_freq, sizeof(inv_freq));
    // ---- streaming-pool update parameters (launch 2, or folded into launch 1) -----------------------
    CompressParams C{};
    int n_compress = 0;
    const int T = a->str_len + 1;
    // the counter advances even for a layer without streaming heads, as the reference's
    // compress_and_replace_streaming_kv does on its zero-head tensors (static_kv_cache.py:127-167)
    if (new_stream_len) *new_stream_len = T <= W ? T : W;
    if (ns > 0) {
        C = CompressParams{(bf16_t *)a->str_k, (bf16_t *)a->str_v, a->str_token_stride, a->str_head_stride,
                           (const bf16_t *)a->k + (int64_t)nf * a->kv_head_stride,
                           (const bf16_t *)a->v + (int64_t)nf * a->kv_head_stride, 0, a->kv_head_stride,
                           ns, a->str_len, 1, a->sink, a->recent, 1, (float)pos_all, {}, dev_state};
        memcpy(C.inv_freq, inv_freq, sizeof(inv_freq));
        C.pos_delta = (float)pos_delta;
        C.p_bs = batched ? B->str_batch_stride : 0;
        C.n_bs = batched ? B->kv_batch_stride : 0;
        n_compress = 2 * ns;
    }
    // Single launch (duo_decode_step_bf16): the kernel merges and updates the streaming pool itself.  Needs one
    // workgroup row per kv head (group == GT), the ticket area, and room for every head's two counters.
    // (... and a grid that is certainly resident at once — one workgroup per CU: the mergers spin while they hold their CU,
    // so a workgroup that has not been dispatched yet must never be waited for)
    const bool one = tickets != nullptr && !batched && group == D.gt && 2 * nkv < kTicketWords - 1 && D.nblk <= 256 &&
                     !(duo_get_debug_flags() & 2u);
    if (one) {
        D.P.one_launch = 1;
        D.P.tickets = (int32_t *)tickets;
        return decode_launch_split<true>(D, C, st);
    }
    // two launches: fold the streaming-pool update into the scan when every streaming head is one workgroup (grid.y == 1)
    static const bool fold_ok = [] { const char *e = getenv("DUO_DECODE_FOLD_COMPRESS"); return !e || atoi(e) != 0; }();
    const bool fold = fold_ok && ns > 0 && group == D.gt && D.P.splits[1] == 1 && !(duo_get_debug_flags() & 2u);
    if (fold) {
        D.P.one_launch = 2;
        n_compress = 0;
    }
    rc = decode_launch_split<true>(D, fold ? C : CompressParams{}, st);
    if (rc) return rc;

    // ---- launch 2: merge (+ streaming-pool update when it was not folded) ----------------------------
    if (D.n_merge + n_compress > 0) {
        hipLaunchKernelGGL(duo_decode_post_kernel, dim3(4 * D.n_merge + n_compress, n_batch), dim3(256), 0, st, D.M, 4 * D.n_merge, C);
        DUO_HIP_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int duo_decode_layer_bf16(const duo_decode_layer_args *a, int32_t *new_stream_len,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    return decode_layer_impl(a, new_stream_len, nullptr, workspace, workspace_bytes, nullptr, stream);
}

// The step for n_batch rows at once (q [B, Hq, 128], new k / v rows [B, Hkv, 128], pools with a batch stride; every row at
// the same cache lengths — the reference's scalar counters, static_kv_cache.py:44-45): the batch row is grid.z of the scan
// and grid.y of the merge launch.  `args` describes row 0.
extern "C" int duo_decode_layer_batched_bf16(const duo_decode_layer_args *a, const duo_decode_batch *batch,
                                             int32_t *new_stream_len, void *workspace, int64_t workspace_bytes,
                                             void *stream) {
    if (!batch) return DUO_EINVAL;
    return decode_layer_impl(a, new_stream_len, nullptr, workspace, workspace_bytes, nullptr, stream, batch);
}

// Same step with the lengths and the position read on the device (see include/duo_attn_hip.h): the
// values in `a` only size the grid, so the two launches can sit in a captured HIP graph and be replayed
// while the cache grows.
extern "C" int duo_decode_layer_dev_bf16(const duo_decode_layer_args *a, const duo_decode_state *dev_state,
                                         void *workspace, int64_t workspace_bytes, void *stream) {
    if (!a || !dev_state) return DUO_EINVAL;
    duo_decode_layer_args plan = *a;
    // planning lengths: at least one cached row per class keeps the segment descriptors on the pools
    if (plan.full_len < 1) plan.full_len = 1;
    if (plan.n_full > 0 && plan.full_len + 1 > plan.full_capacity) plan.full_len = plan.full_capacity - 1;
    if (plan.str_len < 1) plan.str_len = 1;
    return decode_layer_impl(&plan, nullptr, reinterpret_cast<const int32_t *>(dev_state), workspace, workspace_bytes,
                             nullptr, stream);
}

// The batched step with device-side lengths (include/duo_attn_hip.h): one duo_decode_state for all rows of the layer.
extern "C" int duo_decode_layer_batched_dev_bf16(const duo_decode_layer_args *a, const duo_decode_batch *batch,
                                                 const duo_decode_state *dev_state, void *workspace,
                                                 int64_t workspace_bytes, void *stream) {
    if (!a || !batch || !dev_state) return DUO_EINVAL;
    duo_decode_layer_args plan = *a;
    if (plan.full_len < 1) plan.full_len = 1;
    if (plan.n_full > 0 && plan.full_len + 1 > plan.full_capacity) plan.full_len = plan.full_capacity - 1;
    if (plan.str_len < 1) plan.str_len = 1;
    return decode_layer_impl(&plan, nullptr, reinterpret_cast<const int32_t *>(dev_state), workspace, workspace_bytes,
                             nullptr, stream, batch);
}

// The whole step in ONE launch: the split-KV scan as above, and behind an arrival ticket per kv head the last
// workgroups to finish merge the partials and run the streaming pool's sink+recent update (see the end of
// duo_decode_split_kernel).  dev_state == NULL: lengths from `a` (eager); else read on the device (graph replay).
extern "C" int duo_decode_step_bf16(const duo_decode_layer_args *a, int32_t *new_stream_len,
                                    const duo_decode_state *dev_state, void *workspace, int64_t workspace_bytes,
                                    void *tickets, void *stream) {
    if (!a || !tickets) return DUO_EINVAL;
    if (!dev_state) return decode_layer_impl(a, new_stream_len, nullptr, workspace, workspace_bytes, tickets, stream);
    duo_decode_layer_args plan = *a;
    if (plan.full_len < 1) plan.full_len = 1;
    if (plan.n_full > 0 && plan.full_len + 1 > plan.full_capacity) plan.full_len = plan.full_capacity - 1;
    if (plan.str_len < 1) plan.str_len = 1;
    return decode_layer_impl(&plan, new_stream_len, reinterpret_cast<const int32_t *>(dev_state), workspace,
                             workspace_bytes, tickets, stream);
}

namespace {
__global__ void duo_decode_state_add_kernel(duo_decode_state *st, int n, int d_full, int d_str, int d_pos, int str_cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    duo_decode_state s = st[i];
    s.full_len = max(0, s.full_len + d_full);
    s.str_len = min(str_cap, max(0, s.str_len + d_str));
    s.pos = max(0, s.pos + d_pos);
    st[i] = s;
}
}  // namespace

extern "C" int duo_decode_state_add(duo_decode_state *dev_states, int32_t n_layers, int32_t d_full, int32_t d_str,
                                    int32_t d_pos, int32_t str_cap, void *stream) {
    if (!dev_states || n_layers <= 0 || str_cap < 0) return DUO_EINVAL;
    hipLaunchKernelGGL(duo_decode_state_add_kernel, dim3((n_layers + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       dev_states, n_layers, d_full, d_str, d_pos, str_cap);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

#ifdef DUO_DECODE_TIMING
extern "C" int duo_debug_decode_timing(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(duo_decode_timing), sizeof(unsigned long long) * 2048 * 12);
}
#endif
