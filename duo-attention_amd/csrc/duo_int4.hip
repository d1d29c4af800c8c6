// duo_int4.hip — INT4 KV pools (BASELINE config 5 / SURVEY §8f rank 1), gfx950.
//
// The reference's only native code is demo/quantize_int4.cu: a quantiser that walks 128 elements
// serially per thread (:73-144) and a dequantiser launched as one 8-thread block per row (:9-71) that
// rewrites the WHOLE pools to fp16 scratch every layer of every step (demo/int4_kv.py:373-436), after
// which flash_attn_func reads the scratch (demo/w8a8kv4_llama.py:240-274).  Here:
//   duo_int4_quantize_kernel    16 lanes per row, 16-B loads, DPP min/max, 4 packed bytes per lane;
//                               rows go straight to their place in the pool (no staging buffer, no copy_)
//   duo_int4_dequantize_kernel  16 lanes per row, 16-B stores (kept for the reference's get() API)
//   duo_int4_decode_split_kernel  the decode attention reads the packed nibbles + fp16 (scale, zero) in
//                               place and dequantises in registers: 136 B per K or V row-pair... per
//                               token and head: 2 x (64 + 4) = 136 B instead of 512 B of fp16 scratch
//                               written and read again.
// Semantics kept bit for bit (oracle/int4_oracle.py): scale = (max-min)/15 + 1e-8 in fp32, zero = min,
// q = clamp(roundf((x-zero)/scale), 0, 15), even element in the high nibble, scale/zero stored as
// fp16, dequantised value = hadd(hmul(half(q), scale), zero) with both fp16 roundings.
// Pool layout (this repo): packed [h][T][64] u8, sz [h][T][2] f16 = (scale, zero) interleaved; the
// strides are arguments, so the reference's token-major pools work too.
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdlib>
#include "duo_common.h"

namespace {

constexpr float kNegSentinelI4 = -1.0e30f;

// ----------------------------------------------------------------------------- quantise
template <bool BF16>
__device__ __forceinline__ void load8_as_f32(const void *p, float (&f)[8]) {
    const u32x4 w = *reinterpret_cast<const u32x4 *>(p);
    if constexpr (BF16) {
        f[0] = bf16_lo(w.x); f[1] = bf16_hi(w.x); f[2] = bf16_lo(w.y); f[3] = bf16_hi(w.y);
        f[4] = bf16_lo(w.z); f[5] = bf16_hi(w.z); f[6] = bf16_lo(w.w); f[7] = bf16_hi(w.w);
    } else {
        const __half2 *h = reinterpret_cast<const __half2 *>(&w);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 t = __half22float2(h[i]);
            f[2 * i] = t.x;
            f[2 * i + 1] = t.y;
        }
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_min(float x) {
    x = fminf(x, dpp_f<DUO_DPP_QUAD_XOR1>(x));
    x = fminf(x, dpp_f<DUO_DPP_QUAD_XOR2>(x));
    x = fminf(x, dpp_f<DUO_DPP_ROW_HALF_MIRROR>(x));
    return fminf(x, dpp_f<DUO_DPP_ROW_MIRROR>(x));
}
__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, dpp_f<DUO_DPP_QUAD_XOR1>(x));
    x = fmaxf(x, dpp_f<DUO_DPP_QUAD_XOR2>(x));
    x = fmaxf(x, dpp_f<DUO_DPP_ROW_HALF_MIRROR>(x));
    return fmaxf(x, dpp_f<DUO_DPP_ROW_MIRROR>(x));
}

struct QuantParams {
    const void *src;            // [T, h, 128] f16 / bf16
    int64_t s_ts, s_hs;         // element strides
    uint8_t *q;                 // packed pool, row (t, h) at q + (t*q_ts + h*q_hs) * 64
    __half *sz;                 // (scale, zero) pool, row (t, h) at sz + (t*q_ts + h*q_hs) * 2
    int64_t q_ts, q_hs;         // ROW strides of the pool
    int32_t n_tokens, n_heads, dst_row0;
};

// 16 lanes per (token, head) row, 16 rows per 256-thread block
template <bool BF16>
__global__ __launch_bounds__(256) void duo_int4_quantize_kernel(const QuantParams P) {
    const int sub = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t n_rows = (int64_t)P.n_tokens * P.n_heads;
    const bool live = row < n_rows;
    const int64_t r = live ? row : n_rows - 1;       // keep every lane in the DPP reductions
    const int h = (int)(r % P.n_heads);
    const int64_t t = r / P.n_heads;
    float x[8];
    load8_as_f32<BF16>((const char *)P.src + (t * P.s_ts + (int64_t)h * P.s_hs + sub * 8) * 2, x);
    float mn = x[0], mx = x[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) {
        mn = fminf(mn, x[e]);
        mx = fmaxf(mx, x[e]);
    }
    mn = row16_min(mn);
    mx = row16_max(mx);
    const float scale = __fdiv_rn(mx - mn, 15.0f) + 1e-8f;
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float qf = roundf(__fdiv_rn(x[e] - mn, scale));     // roundf: half away from zero, as the reference
        qf = fminf(fmaxf(qf, 0.0f), 15.0f);
        const uint32_t qi = (uint32_t)qf;
        // byte e/2 of the lane's word: even element -> high nibble
        packed |= qi << (8 * (e >> 1) + ((e & 1) ? 0 : 4));
    }
    if (live) {
        const int64_t drow = (P.dst_row0 + t) * P.q_ts + (int64_t)h * P.q_hs;
        reinterpret_cast<uint32_t *>(P.q + drow * 64)[sub] = packed;
        if (sub == 0) {
            P.sz[drow * 2 + 0] = __float2half(scale);
            P.sz[drow * 2 + 1] = __float2half(mn);
        }
    }
}

// ----------------------------------------------------------------------------- dequantise
// Two elements at a time in packed fp16: (hi nibble, lo nibble) of a byte -> half2(q_even, q_odd) via the
// 0x6400 | n == 1024 + n trick (exact), then hmul and hadd as TWO instructions with two roundings — the
// reference is __hadd(__hmul(half(q), s), z) (quantize_int4.cu:36-39).  Inline asm because hipcc contracts
// the HIP header's __hmul/__hadd pair into one v_pk_fma_f16 (single rounding: 1-ulp differences).
__device__ __forceinline__ uint32_t pk_mul_f16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add_f16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// w: 4 packed bytes = elements e0..e7 (byte b: e(2b) high nibble, e(2b+1) low nibble); s2 / z2 = the row's
// scale / zero broadcast to both halves.  o2[b] = half2(dequant(e(2b)), dequant(e(2b+1))).
__device__ __forceinline__ void dequant8_pk(uint32_t w, uint32_t s2, uint32_t z2, uint32_t (&o2)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t byte = (w >> (8 * b)) & 0xffu;
        const uint32_t n2 = (byte >> 4) | ((byte & 0xfu) << 16) | 0x64006400u;   // half2(1024+hi, 1024+lo)
        const uint32_t q2 = pk_add_f16(n2, 0xE400E400u);                         // - 1024: exact
        o2[b] = pk_add_f16(pk_mul_f16(q2, s2), z2);
    }
}
__device__ __forceinline__ void dequant8(uint32_t w, __half s, __half z, __half (&o)[8]) {
    const uint32_t sb = __half_as_ushort(s), zb = __half_as_ushort(z);
    uint32_t o2[4];
    dequant8_pk(w, sb | (sb << 16), zb | (zb << 16), o2);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        o[2 * b] = __ushort_as_half((unsigned short)(o2[b] & 0xffffu));
        o[2 * b + 1] = __ushort_as_half((unsigned short)(o2[b] >> 16));
    }
}

struct DequantParams {
    const uint8_t *q;
    const __half *sz;
    int64_t q_ts, q_hs;         // ROW strides
    __half *out;                // [T, h, 128] contiguous
    int32_t n_tokens, n_heads;
};

__global__ __launch_bounds__(256) void duo_int4_dequantize_kernel(const DequantParams P) {
    const int sub = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= (int64_t)P.n_tokens * P.n_heads) return;
    const int h = (int)(row % P.n_heads);
    const int64_t t = row / P.n_heads;
    const int64_t srow = t * P.q_ts + (int64_t)h * P.q_hs;
    const uint32_t w = reinterpret_cast<const uint32_t *>(P.q + srow * 64)[sub];
    __half o[8];
    dequant8(w, P.sz[srow * 2], P.sz[srow * 2 + 1], o);
    *reinterpret_cast<u32x4 *>(P.out + row * 128 + sub * 8) = *reinterpret_cast<const u32x4 *>(o);
}

// ----------------------------------------------------------------------------- pool compaction
// streaming pool: rows [len-recent, len) -> [sink, sink+recent)   (demo/int4_kv.py:438-492), in place.
// Source row index >= destination row index, so batches of destination rows are loaded, barriered,
// stored (same argument as duo_stream_compress_kernel).  One workgroup per (head, K|V).
struct Int4CompressParams {
    uint8_t *kq, *vq;
    __half *ksz, *vsz;
    int64_t q_ts, q_hs;
    int32_t n_heads, len, sink, recent;
};

__global__ __launch_bounds__(256) void duo_int4_compress_kernel(const Int4CompressParams P) {
    const int h = blockIdx.x >> 1;
    const bool is_v = blockIdx.x & 1;
    uint8_t *q = (is_v ? P.vq : P.kq) + (int64_t)h * P.q_hs * 64;
    __half *sz = (is_v ? P.vsz : P.ksz) + (int64_t)h * P.q_hs * 2;
    const int shift = P.len - P.recent - P.sink;     // > 0
    const int sub = threadIdx.x & 15;                // 4 bytes of the 64-byte row
    const int r_in = threadIdx.x >> 4;               // 16 rows per pass
    for (int d0 = P.sink; d0 < P.sink + P.recent; d0 += 16) {
        const int d = d0 + r_in;
        const bool act = d < P.sink + P.recent;
        uint32_t w = 0, s2 = 0;
        if (act) {
            const int64_t srow = (int64_t)(d + shift) * P.q_ts;
            w = reinterpret_cast<const uint32_t *>(q + srow * 64)[sub];
            if (sub == 0) s2 = *reinterpret_cast<const uint32_t *>(sz + srow * 2);
        }
        __syncthreads();
        if (act) {
            const int64_t drow = (int64_t)d * P.q_ts;
            reinterpret_cast<uint32_t *>(q + drow * 64)[sub] = w;
            if (sub == 0) *reinterpret_cast<uint32_t *>(sz + drow * 2) = s2;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------- fused decode
struct Int4SegDev {
    const uint8_t *kq, *vq;
    const __half *ksz, *vsz;
    int64_t ts, hs;            // ROW strides
    int32_t len;
    int32_t n_kv_heads;
    int32_t q_head_offset;
};

struct Int4DecodeParams {
    const __half *q;
    int64_t q_head_stride;
    __half *out;
    int64_t out_head_stride;
    Int4SegDev cls[2];
    int32_t splits[2];
    int32_t nblk_full;
    int32_t group;
    float scale_log2e;
    float *ws_ml, *ws_acc;
    int32_t max_splits;
};

__device__ __forceinline__ Int4SegDev i4_select(const Int4SegDev &a, const Int4SegDev &b, bool pb) {
    Int4SegDev r;
    r.kq = pb ? b.kq : a.kq; r.vq = pb ? b.vq : a.vq;
    r.ksz = pb ? b.ksz : a.ksz; r.vsz = pb ? b.vsz : a.vsz;
    r.ts = pb ? b.ts : a.ts; r.hs = pb ? b.hs : a.hs;
    r.len = pb ? b.len : a.len;
    r.n_kv_heads = pb ? b.n_kv_heads : a.n_kv_heads;
    r.q_head_offset = pb ? b.q_head_offset : a.q_head_offset;
    return r;
}

__device__ __forceinline__ void dequant8_f32(uint32_t w, uint32_t sz2, float (&f)[8]) {
    const uint32_t sb = sz2 & 0xffffu, zb = sz2 >> 16;    // (scale, zero) pair as stored
    uint32_t o2[4];
    dequant8_pk(w, sb | (sb << 16), zb | (zb << 16), o2);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float2 t = __half22float2(*reinterpret_cast<const __half2 *>(&o2[b]));
        f[2 * b] = t.x;
        f[2 * b + 1] = t.y;
    }
}

// Same decomposition as duo_decode_split_kernel: 256-thread workgroup per (kv head, balanced token
// chunk); 16 lanes per row — 4 packed bytes (8 dims) per lane, so one wave-load covers 4 rows = 256 B;
// 16 tokens of K and V (+ their (scale, zero) words) in flight per wave.
template <int GT>
__global__ __launch_bounds__(256) void duo_int4_decode_split_kernel(const Int4DecodeParams P) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane & 15;
    const int tg = lane >> 4;

    int b = blockIdx.x;
    const int ci = b < P.nblk_full ? 0 : 1;
    if (ci) b -= P.nblk_full;
    const Int4SegDev C = i4_select(P.cls[0], P.cls[1], ci != 0);
    const int splits = ci ? P.splits[1] : P.splits[0];
    const int kvh = b / splits;
    const int split = b - kvh * splits;
    const int qh0 = C.q_head_offset + kvh * P.group + blockIdx.y * GT;

    const int L = C.len;
    const int units = (L + 63) >> 6;
    const int uq = units / splits, ur = units - uq * splits;
    const int u0 = split * uq + min(split, ur);
    const int un = uq + (split < ur ? 1 : 0);
    const int c0 = u0 << 6;
    const int c1 = min((u0 + un) << 6, L);
    const int per_wave = (((c1 - c0 + 3) >> 2) + 15) & ~15;
    const int w0 = c0 + wave * per_wave;
    const int w1 = min(w0 + per_wave, c1);

    const uint8_t *kq = C.kq + (int64_t)kvh * C.hs * 64 + sub * 4;
    const uint8_t *vq = C.vq + (int64_t)kvh * C.hs * 64 + sub * 4;
    const __half *ksz = C.ksz + (int64_t)kvh * C.hs * 2;
    const __half *vsz = C.vsz + (int64_t)kvh * C.hs * 2;

    float qf[GT][8];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        load8_as_f32<false>(P.q + (int64_t)(qh0 + g) * P.q_head_stride + sub * 8, qf[g]);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[g][e] *= P.scale_log2e;
    }
    float m[GT], l[GT], acc[GT][8];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        m[g] = kNegSentinelI4;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }

    for (int t = w0; t < w1; t += 16) {
        uint32_t kw[4], ks[4], vw[4], vs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int tok = t + 4 * u + tg;
            tok = tok < w1 ? tok : w1 - 1;
            const int64_t r = (int64_t)tok * C.ts;
            kw[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(kq + r * 64));
            vw[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(vq + r * 64));
            ks[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(ksz + r * 2));
            vs[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(vsz + r * 2));
        }
        float s[4][GT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float kf[8];
            dequant8_f32(kw[u], ks[u], kf);
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                float d = qf[g][0] * kf[0];
#pragma unroll
                for (int e = 1; e < 8; ++e) d = fmaf(qf[g][e], kf[e], d);
                s[u][g] = row16_allreduce_sum(d);
            }
        }
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) valid[u] = (t + 4 * u + tg) < w1;
        float p[4][GT];
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            float mn = m[g];
#pragma unroll
            for (int u = 0; u < 4; ++u) mn = fmaxf(mn, valid[u] ? s[u][g] : kNegSentinelI4);
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
            dequant8_f32(vw[u], vs[u], vf);
#pragma unroll
            for (int g = 0; g < GT; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p[u][g], vf[e], acc[g][e]);
        }
    }

    // ---- combine token groups, then waves (as duo_decode_split_kernel) ---------------------------
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        float mm = m[g];
        mm = fmaxf(mm, __shfl_xor(mm, 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        const float sc = fast_exp2(m[g] - mm);
        float ll = l[g] * sc;
        ll += __shfl_xor(ll, 16);
        ll += __shfl_xor(ll, 32);
        m[g] = mm;
        l[g] = ll;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = acc[g][e] * sc;
            a += __shfl_xor(a, 16);
            a += __shfl_xor(a, 32);
            acc[g][e] = a;
        }
    }
    __shared__ float s_ml[4][GT][2];
    __shared__ float s_acc[4][GT][DUO_HEAD_DIM];
    if (tg == 0) {
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            if (sub == 0) {
                s_ml[wave][g][0] = m[g];
                s_ml[wave][g][1] = l[g];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s_acc[wave][g][sub * 8 + e] = acc[g][e];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < GT * DUO_HEAD_DIM; idx += 256) {
        const int g = idx >> 7;
        const int d = idx & 127;
        float M = s_ml[0][g][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) M = fmaxf(M, s_ml[w][g][0]);
        float Lsum = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = fast_exp2(s_ml[w][g][0] - M);
            Lsum = fmaf(s_ml[w][g][1], sc, Lsum);
            o = fmaf(s_acc[w][g][d], sc, o);
        }
        const int qh = qh0 + g;
        if (splits == 1) {
            P.out[(int64_t)qh * P.out_head_stride + d] = __float2half(o / Lsum);
        } else {
            const int64_t slot = (int64_t)qh * P.max_splits + split;
            P.ws_acc[slot * DUO_HEAD_DIM + d] = o;
            if (d == 0) {
                P.ws_ml[slot * 2 + 0] = M;
                P.ws_ml[slot * 2 + 1] = Lsum;
            }
        }
    }
}

struct Int4MergeParams {
    const float *ws_ml, *ws_acc;
    __half *out;
    int64_t out_head_stride;
    int32_t max_splits;
    int32_t qh_begin[2], qh_end[2], splits[2];
};

__global__ __launch_bounds__(256) void duo_int4_decode_merge_kernel(const Int4MergeParams P) {
    int qh = blockIdx.x, splits;
    {
        const int n0 = P.splits[0] > 1 ? P.qh_end[0] - P.qh_begin[0] : 0;
        if (qh < n0) { qh += P.qh_begin[0]; splits = P.splits[0]; }
        else { qh = qh - n0 + P.qh_begin[1]; splits = P.splits[1]; }
    }
    const int sl = threadIdx.x >> 5, dq = threadIdx.x & 31;
    const float *ml = P.ws_ml + (int64_t)qh * P.max_splits * 2;
    const float *ac = P.ws_acc + (int64_t)qh * P.max_splits * DUO_HEAD_DIM + dq * 4;
    __shared__ float red[4];
    __shared__ float slm[8][32];
    __shared__ f32x4 so[8][32];
    float M = kNegSentinelI4;
    for (int s = threadIdx.x; s < splits; s += 256) M = fmaxf(M, ml[s * 2]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = M;
    __syncthreads();
    M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float Lsum = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int s = sl; s < splits; s += 8) {
        const float wu = fast_exp2(ml[s * 2] - M);
        Lsum = fmaf(ml[s * 2 + 1], wu, Lsum);
        o = o + *reinterpret_cast<const f32x4 *>(ac + (int64_t)s * DUO_HEAD_DIM) * wu;
    }
    slm[sl][dq] = Lsum;
    so[sl][dq] = o;
    __syncthreads();
    if (sl == 0) {
        float LL = 0.f;
        f32x4 oo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) { LL += slm[i][dq]; oo = oo + so[i][dq]; }
        const float inv = 1.f / LL;
        __half h4[4] = {__float2half(oo.x * inv), __float2half(oo.y * inv), __float2half(oo.z * inv),
                        __float2half(oo.w * inv)};
        *reinterpret_cast<u32x2 *>(P.out + (int64_t)qh * P.out_head_stride + dq * 4) = *reinterpret_cast<const u32x2 *>(h4);
    }
}

}  // namespace

extern "C" int duo_int4_quantize(const void *src, int32_t src_is_bf16, int64_t src_token_stride,
                                 int64_t src_head_stride, void *q_pool, void *sz_pool,
                                 int64_t pool_token_stride_rows, int64_t pool_head_stride_rows,
                                 int32_t n_heads, int32_t n_tokens, int32_t dst_row0, int32_t head_dim,
                                 void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (n_heads <= 0 || n_tokens <= 0) return 0;
    if (!src || !q_pool || !sz_pool || dst_row0 < 0 || ((src_token_stride | src_head_stride) & 7)) return DUO_EINVAL;
    QuantParams P{src, src_token_stride, src_head_stride, (uint8_t *)q_pool, (__half *)sz_pool,
                  pool_token_stride_rows, pool_head_stride_rows, n_tokens, n_heads, dst_row0};
    const int64_t rows = (int64_t)n_tokens * n_heads;
    dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    if (src_is_bf16) hipLaunchKernelGGL(duo_int4_quantize_kernel<true>, grid, block, 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(duo_int4_quantize_kernel<false>, grid, block, 0, (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_int4_dequantize_f16(const void *q_pool, const void *sz_pool, int64_t pool_token_stride_rows,
                                       int64_t pool_head_stride_rows, void *out, int32_t n_heads,
                                       int32_t n_tokens, int32_t head_dim, void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (n_heads <= 0 || n_tokens <= 0) return 0;
    if (!q_pool || !sz_pool || !out) return DUO_EINVAL;
    DequantParams P{(const uint8_t *)q_pool, (const __half *)sz_pool, pool_token_stride_rows,
                    pool_head_stride_rows, (__half *)out, n_tokens, n_heads};
    const int64_t rows = (int64_t)n_tokens * n_heads;
    hipLaunchKernelGGL(duo_int4_dequantize_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0,
                       (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_int4_stream_compress(void *kq, void *ksz, void *vq, void *vsz, int64_t pool_token_stride_rows,
                                        int64_t pool_head_stride_rows, int32_t n_heads, int32_t len,
                                        int32_t sink, int32_t recent, int32_t *new_len, void *stream) {
    if (len < 0 || sink < 0 || recent < 0) return DUO_EINVAL;
    const int W = sink + recent;
    if (new_len) *new_len = len <= W ? len : W;
    if (len <= W || n_heads <= 0) return 0;
    if (!kq || !ksz || !vq || !vsz) return DUO_EINVAL;
    Int4CompressParams P{(uint8_t *)kq, (uint8_t *)vq, (__half *)ksz, (__half *)vsz, pool_token_stride_rows,
                         pool_head_stride_rows, n_heads, len, sink, recent};
    hipLaunchKernelGGL(duo_int4_compress_kernel, dim3(2 * n_heads), dim3(256), 0, (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

static void i4_choose_splits(int n_kv_heads, int L, int max_splits, int budget, int &splits) {
    if (n_kv_heads <= 0 || L <= 0) { splits = 0; return; }
    const int units = (L + 63) / 64;
    int s = budget / n_kv_heads;
    s = std::min(s, std::max(1, units / 4));
    splits = std::max(1, std::min(s, std::min(units, max_splits)));
}

extern "C" int duo_attn_decode_int4_f16(const void *q, int64_t q_head_stride, void *out, int64_t out_head_stride,
                                        int32_t group, const duo_int4_pool *full, const duo_int4_pool *stream_cls,
                                        float scale, int32_t head_dim, void *workspace, int64_t workspace_bytes,
                                        void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (!q || !out || group <= 0) return DUO_EINVAL;
    Int4DecodeParams P;
    P.q = (const __half *)q; P.q_head_stride = q_head_stride;
    P.out = (__half *)out; P.out_head_stride = out_head_stride;
    const duo_int4_pool *src[2] = {full, stream_cls};
    int n_q_heads = 0;
    for (int c = 0; c < 2; ++c) {
        Int4SegDev &S = P.cls[c];
        S = Int4SegDev{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
        if (!src[c] || src[c]->n_kv_heads <= 0) continue;
        const duo_int4_pool &p = *src[c];
        if (!p.k_q || !p.v_q || !p.k_sz || !p.v_sz || p.len <= 0) return DUO_EINVAL;
        S = Int4SegDev{(const uint8_t *)p.k_q, (const uint8_t *)p.v_q, (const __half *)p.k_sz, (const __half *)p.v_sz,
                       p.token_stride_rows, p.head_stride_rows, p.len, p.n_kv_heads, p.q_head_offset};
        n_q_heads += p.n_kv_heads * group;
    }
    if (n_q_heads <= 0) return 0;
    P.group = group;
    P.scale_log2e = scale * 1.4426950408889634f;
    const int64_t per_split = (int64_t)n_q_heads * (DUO_HEAD_DIM + 2) * (int64_t)sizeof(float);
    const int max_splits = workspace ? (int)std::min<int64_t>(workspace_bytes / per_split, 1024) : 0;
    const int ms = max_splits > 0 ? max_splits : 1;
    i4_choose_splits(P.cls[1].n_kv_heads, P.cls[1].len, ms, P.cls[0].n_kv_heads > 0 ? P.cls[1].n_kv_heads : 512, P.splits[1]);
    i4_choose_splits(P.cls[0].n_kv_heads, P.cls[0].len, ms,
                     std::max(512 - P.cls[1].n_kv_heads * P.splits[1], P.cls[0].n_kv_heads), P.splits[0]);
    const int need = std::max(P.splits[0], P.splits[1]);
    if (need > 1 && max_splits < need) return DUO_EWORKSPC;
    P.max_splits = need > 1 ? need : 1;
    P.ws_ml = (float *)workspace;
    P.ws_acc = P.ws_ml ? P.ws_ml + (int64_t)n_q_heads * P.max_splits * 2 : nullptr;
    P.nblk_full = P.cls[0].n_kv_heads * P.splits[0];
    const int nblk = P.nblk_full + P.cls[1].n_kv_heads * P.splits[1];
    const int gt = (group % 4 == 0) ? 4 : (group % 2 == 0) ? 2 : 1;
    dim3 grid(nblk, group / gt), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (gt == 4) hipLaunchKernelGGL(duo_int4_decode_split_kernel<4>, grid, block, 0, st, P);
    else if (gt == 2) hipLaunchKernelGGL(duo_int4_decode_split_kernel<2>, grid, block, 0, st, P);
    else hipLaunchKernelGGL(duo_int4_decode_split_kernel<1>, grid, block, 0, st, P);
    DUO_HIP_CHECK_LAUNCH();
    Int4MergeParams M;
    M.ws_ml = P.ws_ml; M.ws_acc = P.ws_acc; M.out = P.out; M.out_head_stride = out_head_stride;
    M.max_splits = P.max_splits;
    int n_merge = 0;
    for (int c = 0; c < 2; ++c) {
        M.qh_begin[c] = P.cls[c].q_head_offset;
        M.qh_end[c] = P.cls[c].q_head_offset + P.cls[c].n_kv_heads * group;
        M.splits[c] = P.splits[c];
        if (P.splits[c] > 1) n_merge += P.cls[c].n_kv_heads * group;
    }
    if (n_merge > 0) {
        hipLaunchKernelGGL(duo_int4_decode_merge_kernel, dim3(n_merge), dim3(256), 0, st, M);
        DUO_HIP_CHECK_LAUNCH();
    }
    return 0;
}
