// Small shared device pieces of the KV-pool kernels (included by duo_rope_kv.hip and
// duo_decode.hip; everything is inline / internal linkage).
#pragma once
#include "duo_common.h"

namespace {

__device__ __forceinline__ void unpack8f(const u32x4 &w, float (&f)[8]) {
    f[0] = bf16_lo(w.x); f[1] = bf16_hi(w.x);
    f[2] = bf16_lo(w.y); f[3] = bf16_hi(w.y);
    f[4] = bf16_lo(w.z); f[5] = bf16_hi(w.z);
    f[6] = bf16_lo(w.w); f[7] = bf16_hi(w.w);
}
__device__ __forceinline__ u32x4 pack8f(const float (&f)[8]) {
    u32x4 w;
    w.x = pack_bf16x2(f[0], f[1]);
    w.y = pack_bf16x2(f[2], f[3]);
    w.z = pack_bf16x2(f[4], f[5]);
    w.w = pack_bf16x2(f[6], f[7]);
    return w;
}

// sin/cos of an fp32 angle: the reduction to [-0.5, 0.5) revolutions is done in fp64 (exact to
// ~1e-16 of a revolution even at angle 1e6), then the hardware v_sin/v_cos (input in revolutions,
// abs error ~1e-6 — three orders below a bf16 ulp).  ~10x cheaper than the libm sincosf slow path
// that large positions take.
__device__ __forceinline__ void sincos_rev(float angle, float &s, float &c) {
    double r = (double)angle * 0.15915494309189535;   // 1 / (2 pi)
    r -= rint(r);
    const float rf = (float)r;
    s = __builtin_amdgcn_sinf(rf);
    c = __builtin_amdgcn_cosf(rf);
}

// ---------------------------------------------------------------------------
// Streaming pool update (compress_and_replace_streaming_kv,
// static_kv_cache.py:127-167, input = torch.cat([pool[:cur], new]) of
// llama.py:385-390).  X = pool[:cur] ++ new[:n_new], T = cur + n_new.
//   T <= W : pool[cur:T] = new
//   T >  W : pool[r] = X[r] (r < sink);  pool[sink+j] = X[T-recent+j] (j < recent)
// Every source row index is >= its destination row index, so one workgroup per
// (head, K|V) walks the destination rows upward in batches: load a batch into
// registers, barrier, store.  A later batch only reads rows above anything
// already written.
// ---------------------------------------------------------------------------
struct CompressParams {
    bf16_t *kp, *vp;
    int64_t p_ts, p_hs;
    const bf16_t *kn, *vn;
    int64_t n_ts, n_hs;
    int32_t n_heads, cur, n_new, sink, recent;
    // rope != 0 (fused decode step, n_new == 1): the new K row arrives un-rotated and is rotated
    // at position `pos` on its way into the pool (V rows are copied as they are)
    int32_t rope;
    float pos;
    float inv_freq[64];
    const int32_t *dev_state;   // {full_len, str_len, pos, _}: overrides cur / pos when set (captured decode step)
    int64_t p_bs, n_bs;         // batched launches: elements between the batch rows of the pool / of the new rows
    float pos_delta;            // added to the device-side position (a batch row's fixed offset from row 0)
};

constexpr int CMP_ROWS = 64;   // destination rows per batch: 64 rows x 16 chunks / 256 thr = 4 chunks each

__device__ __forceinline__ void duo_stream_compress_block(const CompressParams &P, int blk, int row = 0) {
    int cur = P.cur;
    float pos = P.pos;
    if (P.dev_state) {
        cur = P.dev_state[1];
        pos = (float)P.dev_state[2] + P.pos_delta;
    }
    const int h = blk >> 1;
    const bool is_v = blk & 1;
    bf16_t *pool = (is_v ? P.vp : P.kp) + (int64_t)h * P.p_hs + (int64_t)row * P.p_bs;
    const bf16_t *nw = (is_v ? P.vn : P.kn) + (int64_t)h * P.n_hs + (int64_t)row * P.n_bs;
    const int T = cur + P.n_new;
    const int W = P.sink + P.recent;
    const int ch = threadIdx.x & 15;
    const int r_in = threadIdx.x >> 4;   // 0..15

    int d_begin, d_end;
    if (T <= W) { d_begin = cur; d_end = T; }
    else { d_begin = 0; d_end = W; }

    for (int d0 = d_begin; d0 < d_end; d0 += CMP_ROWS) {
        u32x4 buf[4];
        bool act[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = d0 + r_in + 16 * j;
            act[j] = d < d_end;
            int x = d;                                   // source index in X
            if (T > W && d >= P.sink) x = T - P.recent + (d - P.sink);
            // rows that stay where they are need no traffic
            if (act[j] && x == d && x < cur) act[j] = false;
            if (act[j]) {
                const bf16_t *src = x < cur ? pool + (int64_t)x * P.p_ts : nw + (int64_t)(x - cur) * P.n_ts;
                buf[j] = *reinterpret_cast<const u32x4 *>(src + ch * 8);
                if (P.rope && !is_v && x >= cur) {
                    // chunk ch holds dims 8ch..8ch+7; its rotation partner is chunk ch ^ 8
                    float xs[8], ys[8];
                    unpack8f(buf[j], xs);
                    unpack8f(*reinterpret_cast<const u32x4 *>(src + (ch ^ 8) * 8), ys);
                    const float sgn = ch < 8 ? -1.f : 1.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float sn, cs;
                        sincos_rev(pos * P.inv_freq[((ch & 7) << 3) + e], sn, cs);
                        xs[e] = xs[e] * cs + sgn * ys[e] * sn;
                    }
                    buf[j] = pack8f(xs);
                }
            }
        }
        __syncthreads();   // all loads of this batch complete before any store of it
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = d0 + r_in + 16 * j;
            if (act[j]) *reinterpret_cast<u32x4 *>(pool + (int64_t)d * P.p_ts + ch * 8) = buf[j];
        }
        __syncthreads();
    }
}


}  // namespace
