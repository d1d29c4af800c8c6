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
#include <type_traits>
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
    int64_t s_bs, q_bs;         // batched launch (grid.y = batch row): elements / pool rows between batch rows
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
    load8_as_f32<BF16>((const char *)P.src + ((int64_t)blockIdx.y * P.s_bs + t * P.s_ts + (int64_t)h * P.s_hs + sub * 8) * 2, x);
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
        const int64_t drow = (int64_t)blockIdx.y * P.q_bs + (P.dst_row0 + t) * P.q_ts + (int64_t)h * P.q_hs;
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
__device__ __forceinline__ uint32_t pk_fma_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// w: 4 packed bytes = elements e0..e7 (byte b: e(2b) high nibble, e(2b+1) low nibble); s2 / z2 = the row's
// scale / zero broadcast to both halves.  o2[b] = half2(dequant(e(2b)), dequant(e(2b+1))).
// fused == false: hadd(hmul(q, s), z), two roundings — the source as written, what its `-ffp-contract=off` build
// computes (tests/golden/int4_ref.npz `nocontract`); fused == true: fma(q, s, z), one rounding — what a compiler that
// contracts the pair emits (the `default` build of the same source here; nvcc under the reference's --use_fast_math
// may well do the same).  ~48 % of the values differ between the two by one fp16 ulp.
__device__ __forceinline__ void dequant8_pk(uint32_t w, uint32_t s2, uint32_t z2, uint32_t (&o2)[4], bool fused = false) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t byte = (w >> (8 * b)) & 0xffu;
        const uint32_t n2 = (byte >> 4) | ((byte & 0xfu) << 16) | 0x64006400u;   // half2(1024+hi, 1024+lo)
        const uint32_t q2 = pk_add_f16(n2, 0xE400E400u);                         // - 1024: exact
        o2[b] = fused ? pk_fma_f16(q2, s2, z2) : pk_add_f16(pk_mul_f16(q2, s2), z2);
    }
}
__device__ __forceinline__ void dequant8(uint32_t w, __half s, __half z, __half (&o)[8], bool fused = false) {
    const uint32_t sb = __half_as_ushort(s), zb = __half_as_ushort(z);
    uint32_t o2[4];
    dequant8_pk(w, sb | (sb << 16), zb | (zb << 16), o2, fused);
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
    int32_t fused;              // dequantisation form: 0 = mul then add (two roundings), 1 = fma (one)
    int64_t q_bs, o_bs;         // batched launch (grid.y = batch row): pool rows / output elements between batch rows
};

__global__ __launch_bounds__(256) void duo_int4_dequantize_kernel(const DequantParams P) {
    const int sub = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= (int64_t)P.n_tokens * P.n_heads) return;
    const int h = (int)(row % P.n_heads);
    const int64_t t = row / P.n_heads;
    const int64_t srow = (int64_t)blockIdx.y * P.q_bs + t * P.q_ts + (int64_t)h * P.q_hs;
    const uint32_t w = reinterpret_cast<const uint32_t *>(P.q + srow * 64)[sub];
    __half o[8];
    dequant8(w, P.sz[srow * 2], P.sz[srow * 2 + 1], o, P.fused != 0);
    *reinterpret_cast<u32x4 *>(P.out + (int64_t)blockIdx.y * P.o_bs + row * 128 + sub * 8) = *reinterpret_cast<const u32x4 *>(o);
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
    int64_t q_bs;               // batched launch (grid.y = batch row): pool rows between batch rows
};

__global__ __launch_bounds__(256) void duo_int4_compress_kernel(const Int4CompressParams P) {
    const int h = blockIdx.x >> 1;
    const bool is_v = blockIdx.x & 1;
    uint8_t *q = (is_v ? P.vq : P.kq) + ((int64_t)h * P.q_hs + (int64_t)blockIdx.y * P.q_bs) * 64;
    __half *sz = (is_v ? P.vsz : P.ksz) + ((int64_t)h * P.q_hs + (int64_t)blockIdx.y * P.q_bs) * 2;
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
    int64_t bs;                // batched launch: pool rows between batch rows
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
    uint32_t flags;            // debug: bit 5 = loads only (memory-side ceiling of the access pattern)
    int32_t fused;             // dequantisation form (see dequant8_pk)
    int64_t q_bs, out_bs;      // batched launch (grid.z = batch row): q / out elements between batch rows
    int64_t ws_row_floats;     // ... and floats between the rows' partial areas
};

__device__ __forceinline__ Int4SegDev i4_select(const Int4SegDev &a, const Int4SegDev &b, bool pb) {
    Int4SegDev r;
    r.kq = pb ? b.kq : a.kq; r.vq = pb ? b.vq : a.vq;
    r.ksz = pb ? b.ksz : a.ksz; r.vsz = pb ? b.vsz : a.vsz;
    r.ts = pb ? b.ts : a.ts; r.hs = pb ? b.hs : a.hs;
    r.len = pb ? b.len : a.len;
    r.n_kv_heads = pb ? b.n_kv_heads : a.n_kv_heads;
    r.q_head_offset = pb ? b.q_head_offset : a.q_head_offset;
    r.bs = pb ? b.bs : a.bs;
    return r;
}

__device__ __forceinline__ void dequant8_f32(uint32_t w, uint32_t sz2, float (&f)[8], bool fused) {
    const uint32_t sb = sz2 & 0xffffu, zb = sz2 >> 16;    // (scale, zero) pair as stored
    uint32_t o2[4];
    dequant8_pk(w, sb | (sb << 16), zb | (zb << 16), o2, fused);
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

    const int64_t head_row = (int64_t)kvh * C.hs + (int64_t)blockIdx.z * C.bs;     // (grid.z = batch row)
    const uint8_t *kq = C.kq + head_row * 64 + sub * 4;
    const uint8_t *vq = C.vq + head_row * 64 + sub * 4;
    const __half *ksz = C.ksz + head_row * 2;
    const __half *vsz = C.vsz + head_row * 2;

    float qf[GT][8];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        load8_as_f32<false>(P.q + (int64_t)blockIdx.z * P.q_bs + (int64_t)(qh0 + g) * P.q_head_stride + sub * 8, qf[g]);
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
            dequant8_f32(kw[u], ks[u], kf, P.fused != 0);
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
            dequant8_f32(vw[u], vs[u], vf, P.fused != 0);
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
            P.out[(int64_t)blockIdx.z * P.out_bs + (int64_t)qh * P.out_head_stride + d] = __float2half(o / Lsum);
        } else {
            const int64_t slot = (int64_t)qh * P.max_splits + split;
            P.ws_acc[(int64_t)blockIdx.z * P.ws_row_floats + slot * DUO_HEAD_DIM + d] = o;
            if (d == 0) {
                P.ws_ml[(int64_t)blockIdx.z * P.ws_row_floats + slot * 2 + 0] = M;
                P.ws_ml[(int64_t)blockIdx.z * P.ws_row_floats + slot * 2 + 1] = Lsum;
            }
        }
    }
}

// ----------------------------------------------------------------------------- fused decode, MFMA form
// The scalar-FMA kernel above is VALU-bound: at 136 B per (token, kv head) the HBM roofline asks for
// ~38 K rows/us, and 1024 fp32 FMAs per row (4 q heads x 128 dims x {QK, PV}) alone are the whole
// chip's VALU rate at that pace.  Here the dot products go to the matrix cores and the VALU only
// dequantises (packed fp16, bit-exact with the reference: 17 instructions per 8 values):
//   S^T[16 keys x 16 q] = K^[16 keys x 32 dims] . Q^T        v_mfma_f32_16x16x32_f16, 4 k-steps, 2 key halves
//   O[16 q x 16 dims]  += P[16 q x 32 keys] . V^[32 keys x 16 dims]   8 dim blocks
// (q columns beyond the GQA group are zero padding; the MFMA is ~8x faster than needed even so.)
// One wave owns a tile of 32 keys.  K^ fragments come straight from the packed words: lane (r, g) =
// (key r of the half, 16-byte quarter g of the 64-byte row) holds dims 32g..32g+31, and the order of the
// 8 values inside each packed dword is whatever the nibble extraction yields ([1,5,0,4,3,7,2,6]) — Q is
// loaded in the same order, the dot product does not care.  V^ needs keys along the MFMA k axis, i.e. a
// transpose: the dequantised tile goes through a wave-private 8 KiB LDS tile (XOR-swizzled 16-B chunks)
// and comes back with ds_read_b64_tr_b16.  The S accumulator layout (lane = q column, 4 keys per
// 16-lane group) is exactly the A-operand layout of P if k-slot (g, e) means key 4g+e (e<4) or
// 16+4g+e-4: no cross-lane traffic for P.
// Softmax keeps a per-q-head reference maximum that is only raised when some score exceeds it by 2^8
// (wave vote), so the steady state has no cross-lane reduction and no accumulator rescale.
#ifndef DUO_I4_PROBE
#define DUO_I4_PROBE 0      // ablation builds only (wrong results by design): bit 0 no exp, bit 1 no K dequantisation, bit 2 no V, bit 3 no LDS transpose
#endif
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ h2_t as_h2(uint32_t x) { return *reinterpret_cast<const h2_t *>(&x); }
__device__ __forceinline__ uint32_t as_u32(h2_t x) { return *reinterpret_cast<const uint32_t *>(&x); }
__device__ __forceinline__ h2_t h2_splat(float v) { return h2_t{(_Float16)v, (_Float16)v}; }

// Per-row dequantisation constants, both fp16 lanes equal.  The reference value is
//     hadd(hmul(half(n), s), z)                       (quantize_int4.cu:36-39, two roundings)
// A nibble OR-ed into the mantissa of 1024.0 reads as 1024 + n (bits 0-3) or 1024 + 16 n (bits 4-7), and
//     fma(1024 + n,    s,      -1024 s) = round(n s)         one rounding of the exact product: == hmul
//     fma(1024 + 16 n, s / 16, -64 s)   = round(n s)
// PROVIDED -1024 s and s / 16 are exact in fp16: 2^-10 <= s < 64 (or s == 0).  Rows outside that range
// (wave vote per tile) take the three-instruction form subtract-multiply-add, which is always exact.
struct RowConst {
    h2_t s, z, c, s16, c16;
};
__device__ __forceinline__ RowConst row_const(uint32_t sz, bool fast) {
#pragma clang fp contract(off)
    RowConst R;
    R.s = as_h2(__builtin_amdgcn_perm(sz, sz, 0x01000100u));
    R.z = as_h2(__builtin_amdgcn_perm(sz, sz, 0x03020302u));
    if (fast) {
        R.c = R.s * h2_splat(-1024.f);
        R.s16 = R.s * h2_splat(0.0625f);
        R.c16 = R.s * h2_splat(-64.f);
    }
    return R;
}
// all four (scale, zero) words of a lane carry a scale in [2^-10, 64): as fp16 bit patterns
// 0x1400 <= s < 0x5400  <=>  ((s - 0x1400) & 0xC000) == 0, tested two scales at a time.  (A zero scale —
// a constant row — would qualify too but is left to the general form: one test fewer.)
__device__ __forceinline__ bool scales_in_fma_range(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x05040100u);   // (scale(a), scale(b))
    const uint32_t cd = __builtin_amdgcn_perm(d, c, 0x05040100u);
    const uint32_t k = 0x14001400u;
    uint32_t x, y;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(x) : "v"(ab), "v"(k));
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(y) : "v"(cd), "v"(k));
    return ((x | y) & 0xC000C000u) == 0u;
}

// one packed dword (8 nibbles) -> 4 x half2 of reference-exact dequantised values, element order
// [1,5 | 0,4 | 3,7 | 2,6] of the dword's 8 dims.  m0 / m4: nibble masks 0x000f000f / 0x00f000f0 (SGPRs),
// magic: half2(1024, 1024) (VGPR) — see the note at their definition.
// FUSED: the one-rounding form fma(n, s, z) (see dequant8_pk): (1024 + n) - 1024 = n and (1024 + 16 n) - 1024 = 16 n
// are exact, then one fma against s (or s / 16, exact for voted tiles) — the same 13 / 19 instructions per 8 values.
template <bool FAST, bool FUSED>
__device__ __forceinline__ u32x4 dq8(uint32_t w, const RowConst &R, uint32_t m0, uint32_t m4, uint32_t magic) {
#pragma clang fp contract(off)
    u32x4 o;
    const h2_t bias = h2_splat(1024.f);
    if constexpr (FAST) {
        const uint32_t w8 = w >> 8;
        const h2_t a0 = as_h2((w & m0) | magic), a1 = as_h2((w & m4) | magic);
        const h2_t a2 = as_h2((w8 & m0) | magic), a3 = as_h2((w8 & m4) | magic);
        if constexpr (FUSED) {
            o.x = as_u32(__builtin_elementwise_fma(a0 - bias, R.s, R.z));
            o.y = as_u32(__builtin_elementwise_fma(a1 - bias, R.s16, R.z));
            o.z = as_u32(__builtin_elementwise_fma(a2 - bias, R.s, R.z));
            o.w = as_u32(__builtin_elementwise_fma(a3 - bias, R.s16, R.z));
        } else {
            o.x = as_u32(__builtin_elementwise_fma(a0, R.s, R.c) + R.z);
            o.y = as_u32(__builtin_elementwise_fma(a1, R.s16, R.c16) + R.z);
            o.z = as_u32(__builtin_elementwise_fma(a2, R.s, R.c) + R.z);
            o.w = as_u32(__builtin_elementwise_fma(a3, R.s16, R.c16) + R.z);
        }
    } else {
        const h2_t a0 = as_h2((w & m0) | magic), a1 = as_h2(((w >> 4) & m0) | magic);
        const h2_t a2 = as_h2(((w >> 8) & m0) | magic), a3 = as_h2(((w >> 12) & m0) | magic);
        if constexpr (FUSED) {
            o.x = as_u32(__builtin_elementwise_fma(a0 - bias, R.s, R.z));
            o.y = as_u32(__builtin_elementwise_fma(a1 - bias, R.s, R.z));
            o.z = as_u32(__builtin_elementwise_fma(a2 - bias, R.s, R.z));
            o.w = as_u32(__builtin_elementwise_fma(a3 - bias, R.s, R.z));
        } else {
            o.x = as_u32((a0 - bias) * R.s + R.z);
            o.y = as_u32((a1 - bias) * R.s + R.z);
            o.z = as_u32((a2 - bias) * R.s + R.z);
            o.w = as_u32((a3 - bias) * R.s + R.z);
        }
    }
    return o;
}
__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f16x8_t as_f16x8(const u32x4 &w) { return *reinterpret_cast<const f16x8_t *>(&w); }

#define DUO_I4_TR_READ(dst, addr, off) \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

struct I4Tile {
    u32x4 kw[2], vw[2];
    uint32_t ks[2], vs[2];
};

typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

// MINW: waves per SIMD the register budget is cut for; MODE: 0 = every tile in the always-exact shifted form, 1 = voted
// (tiles whose scales allow it take the masked form); FUSED: fma(n, s, z) instead of hadd(hmul(n, s), z) (see dq8).
template <int MINW, int MODE, bool FUSED>
__global__ __launch_bounds__(256, MINW) void duo_int4_decode_mfma_kernel(const Int4DecodeParams P) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15;    // key row inside a 16-key half / q column / dim column
    const int g = lane >> 4;    // 16-byte quarter of the packed row / k-slot group

    int b = blockIdx.x;
    const int ci = b < P.nblk_full ? 0 : 1;
    if (ci) b -= P.nblk_full;
    const Int4SegDev C = i4_select(P.cls[0], P.cls[1], ci != 0);
    const int splits = ci ? P.splits[1] : P.splits[0];
    const int kvh = b / splits;
    const int split = b - kvh * splits;
    const int qh0 = C.q_head_offset + kvh * P.group;

    const int L = C.len;
    const int units = (L + 63) >> 6;
    const int uq = units / splits, ur = units - uq * splits;
    const int u0 = split * uq + min(split, ur);
    const int un = uq + (split < ur ? 1 : 0);
    const int c0 = u0 << 6;
    const int c1 = min((u0 + un) << 6, L);
    const int per_wave = (((c1 - c0 + 3) >> 2) + 31) & ~31;   // quarter of the chunk, whole 32-key tiles
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + wave * per_wave);
    const int w1 = __builtin_amdgcn_readfirstlane(min(w0 + per_wave, c1));

    // uniform (SGPR) row bases of this kv head + per-lane 32-bit byte offsets inside a 32-key tile
    const int64_t head_row = (int64_t)kvh * C.hs + (int64_t)blockIdx.z * C.bs;     // (grid.z = batch row)
    const uint8_t *kq = C.kq + head_row * 64;
    const uint8_t *vq = C.vq + head_row * 64;
    const uint8_t *ksz = reinterpret_cast<const uint8_t *>(C.ksz) + head_row * 4;
    const uint8_t *vsz = reinterpret_cast<const uint8_t *>(C.vsz) + head_row * 4;
    const int64_t ts = C.ts;
    const uint32_t row_b = (uint32_t)ts * 64u, sz_b = (uint32_t)ts * 4u;   // bytes per token step
    const uint32_t qoff0 = (uint32_t)r * row_b + g * 16, qoff1 = qoff0 + 16u * row_b;
    const uint32_t soff0 = (uint32_t)r * sz_b, soff1 = soff0 + 16u * sz_b;

    // Q^T fragments: column r = q head qh0 + r (zero beyond the group), dims 32g + 8kb + [1,5,0,4,3,7,2,6]
    f16x8_t qB[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        u32x4 w = {0u, 0u, 0u, 0u};
        if (r < P.group)
            w = *reinterpret_cast<const u32x4 *>(P.q + (int64_t)blockIdx.z * P.q_bs + (int64_t)(qh0 + r) * P.q_head_stride + 32 * g + 8 * kb);
        // w = (d0 d1)(d2 d3)(d4 d5)(d6 d7), low half first
        u32x4 o;
        o.x = (w.x >> 16) | (w.z & 0xffff0000u);          // d1, d5
        o.y = (w.x & 0xffffu) | (w.z << 16);              // d0, d4
        o.z = (w.y >> 16) | (w.w & 0xffff0000u);          // d3, d7
        o.w = (w.y & 0xffffu) | (w.w << 16);              // d2, d6
        qB[kb] = as_f16x8(o);
    }
    const u32x4 ones_w = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    const f16x8_t ones = as_f16x8(ones_w);

    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 8192];
    // four-wave form: the Q^T fragments (16 registers) live in LDS and are re-read per tile — they are only needed while
    // S is formed, and 128 registers do not hold them through the V half of the tile body as well
    // (one copy per workgroup: the four waves hold the same fragments — same kv head, same q heads, same lane layout)
    __shared__ __attribute__((aligned(16))) uint8_t lds_q[MINW == 4 ? 4096 : 16];
    const uint32_t qaddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)lds_q + lane * 16;
    if constexpr (MINW == 4) {
        if (wave == 0) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) *(lds_u32x4 *)(uintptr_t)(qaddr + kb * 1024) = *reinterpret_cast<const u32x4 *>(&qB[kb]);
        }
        __syncthreads();
    }
    // V^ tile of this wave: [32 keys][16 chunks of 16 B], chunk c of key k stored at slot c ^ (k & 15)
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)lds + wave * 8192;
    uint32_t wa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[j] = lbase + r * 256 + (((4 * g + j) ^ r) << 4);
    // transpose read: lane i of a 16-lane group supplies the 8-byte piece (row i>>2, cols 4(i&3)..+3) of a
    // 4 x 16 block and receives column i;  rows 4g..4g+3 (and +16), dim block nb -> address ^ (nb << 5)
    const int kr = 4 * g + (r >> 2);
    const uint32_t ra = lbase + kr * 256 + ((((r >> 1) & 1) ^ (kr & 15)) << 4) + 8 * (r & 1);

    // opaque to the optimiser (masks in SGPRs, the exponent word in a VGPR) so that (w & mask) | magic is
    // ONE v_and_or_b32: as literals the pair needs two instructions (one literal per VOP3 on gfx9)
    uint32_t m0, m4, magic;
    asm volatile("s_mov_b32 %0, 0x000f000f" : "=s"(m0));
    asm volatile("s_mov_b32 %0, 0x00f000f0" : "=s"(m4));
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));

    f32x4 O[8], Lacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) O[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_ref = kNegSentinelI4;   // per q column, log2 domain (scores x scale x log2 e)
    const float c_ = P.scale_log2e;

    // A full tile: uniform row bases (SGPR pairs) + the four constant per-lane offsets.  (The empty asm
    // re-materialises each 32-bit offset in the basic block of the loads so that they select the
    // sgpr-base + vgpr32-offset addressing form; in place, so no temporary register is involved.)
    uint32_t qo0 = qoff0, qo1 = qoff1, so0 = soff0, so1 = soff1;
    typedef __attribute__((address_space(1))) const uint8_t gbyte_t;
    typedef __attribute__((address_space(1))) const u32x4 gu32x4_t;
    typedef __attribute__((address_space(1))) const uint32_t gu32_t;
    auto ubase = [&](const uint8_t *p) __attribute__((always_inline)) {
        const uint64_t a = (uint64_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return (gbyte_t *)(((uint64_t)hi << 32) | lo);
    };
    auto load_k_at = [&](int t, I4Tile &T, uint32_t &q0, uint32_t &q1, uint32_t &s0, uint32_t &s1) __attribute__((always_inline)) {
        gbyte_t *kq_t = ubase(kq + (int64_t)t * ts * 64), *ks_t = ubase(ksz + (int64_t)t * ts * 4);
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(s0), "+v"(s1));
        T.kw[0] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(kq_t + q0));
        T.kw[1] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(kq_t + q1));
        T.ks[0] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(ks_t + s0));
        T.ks[1] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(ks_t + s1));
    };
    auto load_v_at = [&](int t, I4Tile &T, uint32_t &q0, uint32_t &q1, uint32_t &s0, uint32_t &s1) __attribute__((always_inline)) {
        gbyte_t *vq_t = ubase(vq + (int64_t)t * ts * 64), *vs_t = ubase(vsz + (int64_t)t * ts * 4);
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(s0), "+v"(s1));
        T.vw[0] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vq_t + q0));
        T.vw[1] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vq_t + q1));
        T.vs[0] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(vs_t + s0));
        T.vs[1] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(vs_t + s1));
    };
    auto load_at = [&](int t, I4Tile &T, uint32_t &q0, uint32_t &q1, uint32_t &s0, uint32_t &s1) __attribute__((always_inline)) {
        gbyte_t *kq_t = ubase(kq + (int64_t)t * ts * 64), *vq_t = ubase(vq + (int64_t)t * ts * 64);
        gbyte_t *ks_t = ubase(ksz + (int64_t)t * ts * 4), *vs_t = ubase(vsz + (int64_t)t * ts * 4);
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(s0), "+v"(s1));
        T.kw[0] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(kq_t + q0));
        T.kw[1] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(kq_t + q1));
        T.vw[0] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vq_t + q0));
        T.vw[1] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vq_t + q1));
        T.ks[0] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(ks_t + s0));
        T.ks[1] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(ks_t + s1));
        T.vs[0] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(vs_t + s0));
        T.vs[1] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(vs_t + s1));
    };
    auto load_full = [&](int t, I4Tile &T) __attribute__((always_inline)) { load_at(t, T, qo0, qo1, so0, so1); };
    // the partial last tile: rows past the range re-read the last one
    auto load_tail = [&](int t, I4Tile &T) __attribute__((always_inline)) {
        const int last = w1 - 1 - t;
        const int r0 = min(r, last), r1 = min(16 + r, last);
        uint32_t q0 = (uint32_t)r0 * row_b + g * 16, q1 = (uint32_t)r1 * row_b + g * 16;
        uint32_t s0 = (uint32_t)r0 * sz_b, s1 = (uint32_t)r1 * sz_b;
        load_at(t, T, q0, q1, s0, s1);
    };
    // SPLIT form (four waves per SIMD): the K half and the V half of a tile are fetched separately into ONE tile buffer —
    // tile t + 32 if it lies in the range (rows past the range re-read its last row), else the range's last tile again
    // (never consumed), so the number of loads in flight is the same on every path
    auto load_half = [&](int tn, I4Tile &T, auto k_tag) __attribute__((always_inline)) {
        constexpr bool KH = decltype(k_tag)::value;
        const int t = tn < w1 ? tn : w0 + (((w1 - 1 - w0) >> 5) << 5);
        if (t + 32 <= w1) {
            if constexpr (KH) load_k_at(t, T, qo0, qo1, so0, so1);
            else load_v_at(t, T, qo0, qo1, so0, so1);
        } else {
            const int last = w1 - 1 - t;
            const int r0 = min(r, last), r1 = min(16 + r, last);
            uint32_t q0 = (uint32_t)r0 * row_b + g * 16, q1 = (uint32_t)r1 * row_b + g * 16;
            uint32_t s0 = (uint32_t)r0 * sz_b, s1 = (uint32_t)r1 * sz_b;
            if constexpr (KH) load_k_at(t, T, q0, q1, s0, s1);
            else load_v_at(t, T, q0, q1, s0, s1);
        }
    };

    // K half of the tile body: S^T = K^ . Q^T.  FAST: the exact-fma dequantisation (every K row scale of the tile in range —
    // the caller has voted); otherwise subtract-multiply-add.
    auto process_k = [&](const I4Tile &T, f32x4 (&S)[2], auto fast_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        f16x8_t qk[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if constexpr (MINW == 4) {      // (volatile: re-read per tile, never hoisted back into registers)
                const u32x4 w = *(volatile lds_u32x4 *)(uintptr_t)(qaddr + kb * 1024);
                qk[kb] = as_f16x8(w);
            } else {
                qk[kb] = qB[kb];
            }
        }
        // ---- S^T = K^ . Q^T ------------------------------------------------------------------
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            S[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            const RowConst R = row_const(T.ks[h], FAST);
            const uint32_t kw[4] = {T.kw[h].x, T.kw[h].y, T.kw[h].z, T.kw[h].w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#if DUO_I4_PROBE & 2     // ablation build (tools/debug/int4_ablation.sh): K words to the MFMA without dequantisation
                const u32x4 raw = {kw[kb] & 0x3bff3bffu, (kw[kb] >> 1) & 0x3bff3bffu, R.s[0] != R.z[1] ? kw[kb] & 0x33ff33ffu : 0u, 0u};
                S[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(raw), qk[kb], S[h], 0, 0, 0);
#else
                S[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(dq8<FAST, FUSED>(kw[kb], R, m0, m4, magic)), qk[kb],
                                                              S[h], 0, 0, 0);
#endif
            }
        }
    };
    // V half: V^ -> LDS, softmax of S against the reference maximum, O += P . V^.  TAIL: partial tile, keys past the range masked.
    auto v_to_lds = [&](const I4Tile &T, auto fast_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        // ---- V^ -> LDS (issued early: the writes drain while the softmax runs) -----------------
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const RowConst R = row_const(T.vs[h], FAST);
            const uint32_t vw[4] = {T.vw[h].x, T.vw[h].y, T.vw[h].z, T.vw[h].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#if DUO_I4_PROBE & 8
                if (R.s[0] != R.s[0]) m_ref += 1.f;     // (keeps the scale loads alive)
#elif DUO_I4_PROBE & 4     // ablation build: V words to LDS without dequantisation
                *(lds_u32x4 *)(uintptr_t)(wa[j] + h * 4096) = u32x4{vw[j] & 0x3bff3bffu, (vw[j] >> 1) & 0x3bff3bffu, R.s[0] != R.z[1] ? vw[j] & 0x33ff33ffu : 0u, 0u};
#else
                *(lds_u32x4 *)(uintptr_t)(wa[j] + h * 4096) = dq8<FAST, FUSED>(vw[j], R, m0, m4, magic);
#endif
            }
        }
    };
    // tail: std::true_type / std::false_type (a compile-time choice), or a bool decided per tile (four-wave form: ONE copy of
    // this body in the loop — with one copy per case the accumulators are shuffled between registers where the cases meet)
    auto softmax_pv = [&](const I4Tile &T, int t, const f32x4 (&S)[2], auto tail) __attribute__((always_inline)) {
        (void)T;
        bool is_tail;
        if constexpr (std::is_same<decltype(tail), bool>::value) is_tail = tail;
        else is_tail = decltype(tail)::value;
        // ---- softmax against the reference maximum ---------------------------------------------
        float sv[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) sv[4 * h + e] = S[h][e];
        if (is_tail) {   // keys past the range score -inf
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t + 16 * h + 4 * g + e >= w1) sv[4 * h + e] = kNegSentinelI4;
        }
        const float mx = c_ * fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])),
                                    fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        if (__any(mx > m_ref + 8.f)) {
            // raise the reference: true running maximum of every q column, accumulators rescaled
            float tm = fmaxf(mx, __shfl_xor(mx, 16));
            tm = fmaxf(tm, __shfl_xor(tm, 32));
            const float m_new = fmaxf(m_ref, tm);
            const float alpha = fast_exp2(m_ref - m_new);
            m_ref = m_new;
            // O rows are q heads 4g + e: fetch their factors from the lanes that own those columns
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __shfl(alpha, 4 * g + e);
                Lacc[e] *= a;
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) O[nb][e] *= a;
            }
        }
        u32x4 pw;
        {
            float p[8];
#pragma unroll
#if DUO_I4_PROBE & 1     // ablation build: no transcendental (a bounded stand-in for the probability)
            for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_fmed3f(fmaf(sv[i], c_, -m_ref), 0.f, 1.f);
#else
            for (int i = 0; i < 8; ++i) p[i] = fast_exp2(fmaf(sv[i], c_, -m_ref));
#endif
            pw.x = cvt_pk_f16(p[0], p[1]);
            pw.y = cvt_pk_f16(p[2], p[3]);
            pw.z = cvt_pk_f16(p[4], p[5]);
            pw.w = cvt_pk_f16(p[6], p[7]);
        }
        const f16x8_t pA = as_f16x8(pw);
        // row sums of the (fp16-rounded) probabilities: one more MFMA against a block of ones
        Lacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pA, ones, Lacc, 0, 0, 0);

        // ---- O += P . V^ : transpose reads two dim blocks at a time, one batch ahead of the MFMAs ------
        u32x2 va[4], vb[4];
        __builtin_amdgcn_sched_barrier(0);
#if DUO_I4_PROBE & 8     // ablation build: no LDS round trip, the MFMAs of P . V^ take the packed words as they are
        {
            const u32x4 g0 = {T.vw[0].x & 0x3bff3bffu, T.vw[0].y & 0x3bff3bffu, T.vw[0].z & 0x3bff3bffu, T.vw[0].w & 0x3bff3bffu};
            const u32x4 g1 = {T.vw[1].x & 0x3bff3bffu, T.vw[1].y & 0x3bff3bffu, T.vw[1].z & 0x3bff3bffu, T.vw[1].w & 0x3bff3bffu};
#pragma unroll
            for (int nb = 0; nb < 8; ++nb)
                O[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pA, as_f16x8((nb & 1) ? g1 : g0), O[nb], 0, 0, 0);
            (void)va; (void)vb;
            return;
        }
#endif
#define DUO_I4_TR_BATCH(buf, nb0)                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                              \
        uint32_t a_ = ra ^ (uint32_t)(((nb0) + i_) << 5);                           \
        if constexpr (MINW == 4) /* recomputed at every use: eight hoisted copies would be eight registers too many */ \
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a_) : "v"(ra), "v"((uint32_t)(((nb0) + i_) << 5)));           \
        DUO_I4_TR_READ(buf[2 * i_], a_, 0);                                         \
        DUO_I4_TR_READ(buf[2 * i_ + 1], a_, 4096);                                  \
    }
#define DUO_I4_PV(buf, nb0)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                   \
        const u32x4 w_ = {buf[2 * i_].x, buf[2 * i_].y, buf[2 * i_ + 1].x, buf[2 * i_ + 1].y};           \
        O[(nb0) + i_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pA, as_f16x8(w_), O[(nb0) + i_], 0, 0, 0); \
    }
#define DUO_I4_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
        DUO_I4_TR_BATCH(va, 0);
        DUO_I4_TR_BATCH(vb, 2);
        DUO_I4_WAIT(4);
        DUO_I4_PV(va, 0);
        __builtin_amdgcn_sched_barrier(0);
        DUO_I4_TR_BATCH(va, 4);
        DUO_I4_WAIT(4);
        DUO_I4_PV(vb, 2);
        __builtin_amdgcn_sched_barrier(0);
        DUO_I4_TR_BATCH(vb, 6);
        DUO_I4_WAIT(4);
        DUO_I4_PV(va, 4);
        DUO_I4_WAIT(0);
        DUO_I4_PV(vb, 6);
#undef DUO_I4_TR_BATCH
#undef DUO_I4_PV
#undef DUO_I4_WAIT
    };
    // Whole tile (two- and three-wave forms: both halves of a tile arrive together and share one vote).
    auto process = [&](const I4Tile &T, int t, auto fast_tag, auto tail_tag) __attribute__((always_inline)) {
        if (P.flags & 32u) {   // debug: consume the loads, skip the arithmetic
            m_ref += __uint_as_float((T.kw[0].x ^ T.kw[1].y ^ T.vw[0].z ^ T.vw[1].w ^ T.ks[0] ^ T.ks[1] ^ T.vs[0] ^ T.vs[1]) & 1u);
            return;
        }
        f32x4 S[2];
        process_k(T, S, fast_tag);
        v_to_lds(T, fast_tag);
        softmax_pv(T, t, S, tail_tag);
    };
    // FOUR waves per SIMD (MINW == 4: 128 registers per wave): ONE tile buffer, refilled half by half — the K half of tile
    // t + 32 is requested as soon as S(t) has consumed K(t), the V half as soon as P.V(t) has consumed V(t) — so four loads
    // are in flight at every use and every wait is vmcnt(4).  A wave has half a tile of fetch lead instead of a whole one;
    // the fourth wave of the SIMD covers the difference (round 6: the three-wave form keeps two tile buffers = 144 VGPRs;
    // the 16 registers above 128 were exactly one half-tile).  The scale vote is taken per half (K scales before S, V
    // scales before the LDS write), so a tile whose rows fail it takes the exact form for that half only — no reload.
    if constexpr (MINW == 4) {
        if (w0 < w1) {
            using T_ = std::true_type;
            using F_ = std::false_type;
            I4Tile X;
            load_half(w0, X, T_{});
            load_half(w0, X, F_{});
            if (P.flags & 32u) {   // debug: consume the loads, skip the arithmetic (a loop of its own: a second path through
                                   // the main loop would make the tile registers meet through copies, i.e. through vmcnt(0))
                for (int t = w0; t < w1; t += 32) {
                    m_ref += __uint_as_float((X.kw[0].x ^ X.kw[1].y ^ X.ks[0] ^ X.ks[1]) & 1u);
                    __builtin_amdgcn_sched_barrier(0);
                    load_half(t + 32, X, T_{});
                    __builtin_amdgcn_sched_barrier(0);
                    m_ref += __uint_as_float((X.vw[0].z ^ X.vw[1].w ^ X.vs[0] ^ X.vs[1]) & 1u);
                    __builtin_amdgcn_sched_barrier(0);
                    load_half(t + 32, X, F_{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int t = w0; t < w1; t += 32) {
                    const bool tail = t + 32 > w1;
                    f32x4 S[2];
                    if (MODE == 1 && __all(scales_in_fma_range(X.ks[0], X.ks[1], X.ks[0], X.ks[1]))) process_k(X, S, T_{});
                    else process_k(X, S, F_{});
                    __builtin_amdgcn_sched_barrier(0);
                    load_half(t + 32, X, T_{});
                    __builtin_amdgcn_sched_barrier(0);
                    if (MODE == 1 && __all(scales_in_fma_range(X.vs[0], X.vs[1], X.vs[0], X.vs[1]))) v_to_lds(X, T_{});
                    else v_to_lds(X, F_{});
                    softmax_pv(X, t, S, tail);
                    __builtin_amdgcn_sched_barrier(0);
                    load_half(t + 32, X, F_{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    } else {
    // Main loop over the full tiles: straight-line code with a fixed number of loads in flight (the next
    // tile's 8 loads are issued, unconditionally, before the current tile is touched; after the last full
    // tile the prefetch re-reads it) so every wait is vmcnt(8), never vmcnt(0).  It runs the FAST body and
    // is left at a tile whose rows fail the scale vote: that one tile (re-loaded) goes through the
    // always-exact general form, then the fast loop resumes.  The partial last tile is general too.
    if (w0 < w1) {
        using T_ = std::true_type;
        using F_ = std::false_type;
        const int t_full_end = w0 + (((w1 - w0) >> 5) << 5);   // end of the full tiles
        int t = w0;
        while (t < w1) {
            if (MODE != 0 && t < t_full_end) {
                const int t_last = t_full_end - 32;
                I4Tile A, B;
                load_full(t, A);
                for (;;) {
                    load_full(min(t + 32, t_last), B);
                    __builtin_amdgcn_sched_barrier(0);
                    if (MODE == 1 && !__all(scales_in_fma_range(A.ks[0], A.ks[1], A.vs[0], A.vs[1]))) break;
                    process(A, t, T_{}, F_{});
                    t += 32;
                    if (t > t_last) break;
                    load_full(min(t + 32, t_last), A);
                    __builtin_amdgcn_sched_barrier(0);
                    if (MODE == 1 && !__all(scales_in_fma_range(B.ks[0], B.ks[1], B.vs[0], B.vs[1]))) break;
                    process(B, t, T_{}, F_{});
                    t += 32;
                    if (t > t_last) break;
                }
                if (t >= w1) break;
            }
            // one tile in the general form: the one that failed the vote, or the partial last one
            I4Tile X;
            if (t + 32 <= w1) {
                load_full(t, X);
                process(X, t, F_{}, F_{});
            } else {
                load_tail(t, X);
                process(X, t, F_{}, T_{});
            }
            t += 32;
        }
    }
    }

    // ---- combine the 4 waves through LDS ------------------------------------------------------
    __syncthreads();   // every wave is done with its V^ tile: the LDS is reused for the partials
    float *s_acc = reinterpret_cast<float *>(lds);                 // [4 waves][16 q][128 dims]  = 32 KiB
    __shared__ float s_ml[4][16][2];
    if (g == 0) s_ml[wave][r][0] = m_ref;
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s_ml[wave][4 * g + e][1] = Lacc[e];
    }
    // O[nb][e] = (q row 4g + e, LDS column 16 nb + r) ; LDS column c <-> dim 8 (c >> 3) + perm[c & 7]
    {
        const int permv = (0x62734051u >> (4 * (r & 7))) & 15;   // [1,5,0,4,3,7,2,6][r & 7]
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const int dim = 16 * nb + 8 * (r >> 3) + permv;
#pragma unroll
            for (int e = 0; e < 4; ++e) s_acc[(wave * 16 + 4 * g + e) * DUO_HEAD_DIM + dim] = O[nb][e];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < P.group * DUO_HEAD_DIM; idx += 256) {
        const int q = idx >> 7;
        const int d = idx & 127;
        float M = s_ml[0][q][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) M = fmaxf(M, s_ml[w][q][0]);
        float Lsum = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = fast_exp2(s_ml[w][q][0] - M);
            Lsum = fmaf(s_ml[w][q][1], sc, Lsum);
            o = fmaf(s_acc[(w * 16 + q) * DUO_HEAD_DIM + d], sc, o);
        }
        const int qh = qh0 + q;
        if (splits == 1) {
            P.out[(int64_t)blockIdx.z * P.out_bs + (int64_t)qh * P.out_head_stride + d] = __float2half(o / Lsum);
        } else {
            const int64_t slot = (int64_t)qh * P.max_splits + split;
            P.ws_acc[(int64_t)blockIdx.z * P.ws_row_floats + slot * DUO_HEAD_DIM + d] = o;
            if (d == 0) {
                P.ws_ml[(int64_t)blockIdx.z * P.ws_row_floats + slot * 2 + 0] = M;
                P.ws_ml[(int64_t)blockIdx.z * P.ws_row_floats + slot * 2 + 1] = Lsum;
            }
        }
    }
}

// ----------------------------------------------------------------------------- fused decode, FOLDED form
// duo_int4_decode_mfma_kernel spends most of its issue slots turning nibbles into the fp16 values the reference's
// dequantiser would have written (13 VALU instructions per 8 values, 61 % of the issue slots at 5 TB/s): it is the one
// HBM-streaming kernel of this library that the VALU co-limits.  The attention does not need those values — only their sums:
//
//   score(k, q)  = sum_d q_d (n_kd s_k + z_k)     = s_k (N . Q^T)[k, q] + z_k Qsum_q          Qsum_q = sum_d q_d
//   out(q, d)    = sum_k p_k (n_kd s'_k + z'_k)   = ((P s') . N')[q, d] + sum_k p_k z'_k
//
// so the matrix cores multiply the RAW nibbles and the per-row scale / zero are applied to the 16 x 16 score tile (8 values
// per lane) and to P (8 values per lane) instead of to the 2 x 32 x 128 elements of the tile.  A nibble masked out of a packed
// word IS an fp16 number: bits 0-3 of a half read as the denormal n 2^-24, bits 4-7 as n 2^-20, and v_mfma_f32_16x16x32_f16
// keeps fp16 denormals exactly (tools/probes/mfma_denorm_probe.hip, profiles/r4_mfma_denormal_probe.txt) — one v_and_b32
// per two values and no conversion at all.  The 2^-24 / 2^-20 split is per DIM position: on the K side it is folded into Q
// (the dims under the low-nibble mask are pre-scaled by 16 — exact), on the V side into the output columns (even dims x 2^-4,
// once, in the epilogue).  (scale, zero) pairs are fetched in the score tile's own layout (lane (r, g) needs the four keys
// 4g..4g+3 of each half: 16 contiguous bytes when the pool is head-major, which is what this form requires).
//
// Arithmetic of a folded tile: exact products of the exact nibbles with fp16 q (resp. fp16-rounded p s'), fp32 accumulation,
// scale / zero applied in fp32 — the attention over n s + z WITHOUT the dequantiser's fp16 roundings.  The reference's values
// are those roundings' (hadd(hmul(n, s), z), or the fma): they differ from n s + z by up to an fp16 ulp OF THE VALUE, which is
// harmless for rows of ordinary magnitude (2^-11 relative: the size of the fp16 rounding of P the reference's flash-attn
// performs anyway) and is NOT for rows with huge entries — a key row of magnitude ~10^3 carries rounding errors of ~0.5 per
// element in the reference, which move its score by tenths of a nat, and the reference's output follows those errors.  So a
// tile is folded only when every row in it is tame — K and V scales below 1 (row range below 15), zero points inside (-8, 8),
// the largest V scale not below 2^-12 (P' = p s' stays a normal fp16 number) — by a wave vote over values the tile needs
// anyway; any other tile (and any NaN) takes the EXACT body: the reference's dequantised values in the form P.fused selects,
// element by element, exactly like duo_int4_decode_mfma_kernel's general path, accumulated into the same state.
// OPT-IN (fused == 2, + 1 for the fma form in the exact tiles: fused == 3; DuoAttentionStaticINT4KVCache(folded_decode=True)):
// measured against the oracle's attention over the reference's dequantised values (profiles/r4_int4_fold.md), the folded
// tiles' outputs sit at 0.3 of the bar that budgets ONE fp16 ulp per dequantised value (2^-10 sum_k p_k |v_k|), but at 1.6-2.5x
// the strict decode bar the dequantising kernel meets (0.3-0.65 of it) when the reference is the two-rounding form, and at
// 0.6-1.2x when it is the fma form: the deviation IS the reference's own rounding of n s (+ z), which no arithmetic on sums
// can reproduce.  The default decode therefore stays the dequantising kernel; this one is for callers who accept "the
// reference up to its own value rounding" for 9-15 % more tokens per second.
// Per folded 32-key tile and lane: 16 v_and + 4 shifts to build the operands of each of K and V, 16 + 16 conversions of (s, z),
// ~70 fp32 operations for the two fix-ups, the vote and the softmax, 8 v_exp, 16 MFMAs — about 70 % of the issue slots of the
// dequantising form.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void duo_int4_decode_fold_kernel(const Int4DecodeParams P) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15;    // key row inside a 16-key half / q column / dim column
    const int g = lane >> 4;    // 16-byte quarter of the packed row / k-slot group

    int b = blockIdx.x;
    const int ci = b < P.nblk_full ? 0 : 1;
    if (ci) b -= P.nblk_full;
    const Int4SegDev C = i4_select(P.cls[0], P.cls[1], ci != 0);
    const int splits = ci ? P.splits[1] : P.splits[0];
    const int kvh = b / splits;
    const int split = b - kvh * splits;
    const int qh0 = C.q_head_offset + kvh * P.group;

    const int L = C.len;
    const int units = (L + 63) >> 6;
    const int uq = units / splits, ur = units - uq * splits;
    const int u0 = split * uq + min(split, ur);
    const int un = uq + (split < ur ? 1 : 0);
    const int c0 = u0 << 6;
    const int c1 = min((u0 + un) << 6, L);
    const int per_wave = (((c1 - c0 + 3) >> 2) + 31) & ~31;   // quarter of the chunk, whole 32-key tiles
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + wave * per_wave);
    const int w1 = __builtin_amdgcn_readfirstlane(min(w0 + per_wave, c1));

    // uniform row bases of this kv head (token stride == 1 row: the launcher guarantees it) + per-lane byte offsets
    const int64_t head_row = (int64_t)kvh * C.hs + (int64_t)blockIdx.z * C.bs;
    const uint8_t *kq = C.kq + head_row * 64;
    const uint8_t *vq = C.vq + head_row * 64;
    const uint8_t *ksz = reinterpret_cast<const uint8_t *>(C.ksz) + head_row * 4;
    const uint8_t *vsz = reinterpret_cast<const uint8_t *>(C.vsz) + head_row * 4;
    const uint32_t qoff0 = (uint32_t)r * 64u + g * 16, qoff1 = qoff0 + 16u * 64u;
    const uint32_t soff0 = (uint32_t)g * 16u, soff1 = soff0 + 64u;     // (s, z) of keys 4g..4g+3 / 16+4g..16+4g+3

    // Q^T fragments as in duo_int4_decode_mfma_kernel: column r = q head qh0 + r (zero beyond the group), dims 32g + 8kb +
    // [1,5 | 0,4 | 3,7 | 2,6]; in qB the pairs that meet LOW-nibble values (n 2^-24: dwords .x and .z) are pre-scaled by 16 so
    // that every product carries the factor 2^-20 (qX: the plain fragments, for exact tiles).  Qsum = sum of the row's 128
    // values (fp32), times scale*log2(e).
    f16x8_t qB[4], qX[4];
    float qs = 0.f;
    const h2_t sixteen = h2_splat(16.f);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        u32x4 w = {0u, 0u, 0u, 0u};
        if (r < P.group)
            w = *reinterpret_cast<const u32x4 *>(P.q + (int64_t)blockIdx.z * P.q_bs + (int64_t)(qh0 + r) * P.q_head_stride + 32 * g + 8 * kb);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 t = __half22float2(*reinterpret_cast<const __half2 *>(&ww[i]));
            qs += t.x + t.y;
        }
        u32x4 o;
        o.x = (w.x >> 16) | (w.z & 0xffff0000u);          // d1, d5   (low nibbles)
        o.y = (w.x & 0xffffu) | (w.z << 16);              // d0, d4   (high nibbles)
        o.z = (w.y >> 16) | (w.w & 0xffff0000u);          // d3, d7
        o.w = (w.y & 0xffffu) | (w.w << 16);              // d2, d6
        qX[kb] = as_f16x8(o);
        o.x = as_u32(as_h2(o.x) * sixteen);
        o.z = as_u32(as_h2(o.z) * sixteen);
        qB[kb] = as_f16x8(o);
    }
    qs += __shfl_xor(qs, 16);
    qs += __shfl_xor(qs, 32);
    const float c_ = P.scale_log2e;
    const float cA = c_ * 1048576.f;      // scale * log2(e) * 2^20: undoes the operands' common factor
    const float Bq = c_ * qs;

    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 8192];
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)lds + wave * 8192;
    uint32_t wa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[j] = lbase + r * 256 + (((4 * g + j) ^ r) << 4);
    const int kr = 4 * g + (r >> 2);
    const uint32_t ra = lbase + kr * 256 + ((((r >> 1) & 1) ^ (kr & 15)) << 4) + 8 * (r & 1);

    uint32_t m0, m4, magic;
    asm volatile("s_mov_b32 %0, 0x000f000f" : "=s"(m0));
    asm volatile("s_mov_b32 %0, 0x00f000f0" : "=s"(m4));
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));

    // accumulators in the FOLDED scale: O = sum (p s') n 2^-24 (odd-position dims) / 2^-20 (even-position dims); an exact
    // tile's true-scale contribution is brought to it by 1 / colscale
    f32x4 O[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) O[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float colscale = (r & 2) ? 1048576.f : 16777216.f;      // this lane's output columns: 2^20 (high-nibble dims) / 2^24
    const float colinv = 1.f / colscale;
    float Lp = 0.f, Zp = 0.f;             // this lane's share (its 8 keys per tile) of sum p and sum p z', q column r
    float m_ref = kNegSentinelI4;

    struct FTile {
        u32x4 kw[2], vw[2], ks[2], vs[2];
    };
    uint32_t qo0 = qoff0, qo1 = qoff1, so0 = soff0, so1 = soff1;
    typedef __attribute__((address_space(1))) const uint8_t gbyte_t;
    typedef __attribute__((address_space(1))) const u32x4 gu32x4_t;
    typedef __attribute__((address_space(1))) const uint32_t gu32_t;
    auto ubase = [&](const uint8_t *p) __attribute__((always_inline)) {
        const uint64_t a = (uint64_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return (gbyte_t *)(((uint64_t)hi << 32) | lo);
    };
    auto load_full = [&](int t, FTile &T) __attribute__((always_inline)) {
        gbyte_t *kq_t = ubase(kq + (int64_t)t * 64), *vq_t = ubase(vq + (int64_t)t * 64);
        gbyte_t *ks_t = ubase(ksz + (int64_t)t * 4), *vs_t = ubase(vsz + (int64_t)t * 4);
        asm volatile("" : "+v"(qo0), "+v"(qo1), "+v"(so0), "+v"(so1));
        T.kw[0] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(kq_t + qo0));
        T.kw[1] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(kq_t + qo1));
        T.ks[0] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(ks_t + so0));
        T.ks[1] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(ks_t + so1));
        T.vw[0] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vq_t + qo0));
        T.vw[1] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vq_t + qo1));
        T.vs[0] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vs_t + so0));
        T.vs[1] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vs_t + so1));
    };
    // the partial last tile: rows past the range re-read the last one, element by element for the (s, z) words
    auto load_tail = [&](int t, FTile &T) __attribute__((always_inline)) {
        const int last = w1 - 1 - t;
        gbyte_t *kq_t = ubase(kq + (int64_t)t * 64), *vq_t = ubase(vq + (int64_t)t * 64);
        gbyte_t *ks_t = ubase(ksz + (int64_t)t * 4), *vs_t = ubase(vsz + (int64_t)t * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t q_ = (uint32_t)min(16 * h + r, last) * 64u + g * 16;
            T.kw[h] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(kq_t + q_));
            T.vw[h] = __builtin_nontemporal_load(reinterpret_cast<gu32x4_t *>(vq_t + q_));
            uint32_t ke[4], ve[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t s_ = (uint32_t)min(16 * h + 4 * g + e, last) * 4u;
                ke[e] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(ks_t + s_));
                ve[e] = __builtin_nontemporal_load(reinterpret_cast<gu32_t *>(vs_t + s_));
            }
            T.ks[h] = u32x4{ke[0], ke[1], ke[2], ke[3]};
            T.vs[h] = u32x4{ve[0], ve[1], ve[2], ve[3]};
        }
    };
    // 8 nibbles of a packed dword as 4 x half2 MFMA operand words: [e1 e5 | e0 e4 | e3 e7 | e2 e6], low-nibble pairs as
    // n 2^-24, high-nibble pairs as n 2^-20
    auto nib = [&](uint32_t w) __attribute__((always_inline)) -> u32x4 {
        const uint32_t w8 = w >> 8;
        return u32x4{w & m0, w & m4, w8 & m0, w8 & m4};
    };
    auto lo_f = [](uint32_t x) __attribute__((always_inline)) { return __half2float(__ushort_as_half((unsigned short)(x & 0xffffu))); };
    auto hi_f = [](uint32_t x) __attribute__((always_inline)) { return __half2float(__ushort_as_half((unsigned short)(x >> 16))); };

    auto process = [&](const FTile &T, int t, auto tail_tag) __attribute__((always_inline)) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if (P.flags & 32u) {   // debug: consume the loads, skip the arithmetic
            m_ref += __uint_as_float((T.kw[0].x ^ T.kw[1].y ^ T.vw[0].z ^ T.vw[1].w ^ T.ks[0].x ^ T.ks[1].y ^ T.vs[0].z ^ T.vs[1].w) & 1u);
            return;
        }
        // ---- (scale, zero) of this lane's 8 keys, and the vote: is every row of the tile tame? -----------------
        float ksf[8], kzf[8], vsf[8], vzf[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t ks[4] = {T.ks[h].x, T.ks[h].y, T.ks[h].z, T.ks[h].w};
            const uint32_t vs[4] = {T.vs[h].x, T.vs[h].y, T.vs[h].z, T.vs[h].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ksf[4 * h + e] = lo_f(ks[e]);
                kzf[4 * h + e] = hi_f(ks[e]);
                vsf[4 * h + e] = lo_f(vs[e]);
                vzf[4 * h + e] = hi_f(vs[e]);
            }
        }
        const float vsm = fmaxf(fmaxf(fmaxf(vsf[0], vsf[1]), fmaxf(vsf[2], vsf[3])), fmaxf(fmaxf(vsf[4], vsf[5]), fmaxf(vsf[6], vsf[7])));
        const float sm = fmaxf(vsm, fmaxf(fmaxf(fmaxf(ksf[0], ksf[1]), fmaxf(ksf[2], ksf[3])), fmaxf(fmaxf(ksf[4], ksf[5]), fmaxf(ksf[6], ksf[7]))));
        float zm = fmaxf(fmaxf(fmaxf(fabsf(kzf[0]), fabsf(kzf[1])), fmaxf(fabsf(kzf[2]), fabsf(kzf[3]))),
                         fmaxf(fmaxf(fabsf(kzf[4]), fabsf(kzf[5])), fmaxf(fabsf(kzf[6]), fabsf(kzf[7]))));
        zm = fmaxf(zm, fmaxf(fmaxf(fmaxf(fabsf(vzf[0]), fabsf(vzf[1])), fmaxf(fabsf(vzf[2]), fabsf(vzf[3]))),
                             fmaxf(fmaxf(fabsf(vzf[4]), fabsf(vzf[5])), fmaxf(fabsf(vzf[6]), fabsf(vzf[7])))));
        // (comparisons written so that a NaN fails them)
        const bool exact = !__all(sm < 1.0f && zm < 8.0f) || !__any(vsm >= 0.000244140625f);

        float sv[8];
        if (!exact) {
            // ---- raw S^T = N . Q'^T, then s_k raw + z_k Qsum (log2 domain) ----------------------------
            f32x4 S[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                S[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                const uint32_t kw[4] = {T.kw[h].x, T.kw[h].y, T.kw[h].z, T.kw[h].w};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
                    S[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(nib(kw[kb])), qB[kb], S[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t vw[4] = {T.vw[h].x, T.vw[h].y, T.vw[h].z, T.vw[h].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) *(lds_u32x4 *)(uintptr_t)(wa[j] + h * 4096) = nib(vw[j]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[4 * h + e] = fmaf(S[h][e], ksf[4 * h + e] * cA, kzf[4 * h + e] * Bq);
        } else {
            // ---- EXACT tile: the reference's dequantised values, element by element (rare) ---------------
            // (scale, zero) in the ROW layout dq8 wants (lane (r, g): rows r and 16 + r) — re-read, clamped into the range
            const int last = w1 - 1 - t;
            uint32_t krow[2], vrow[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t row = (int64_t)t + min(16 * h + r, last);
                krow[h] = *reinterpret_cast<const uint32_t *>(ksz + row * 4);
                vrow[h] = *reinterpret_cast<const uint32_t *>(vsz + row * 4);
            }
            f32x4 S[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                S[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                const RowConst R = row_const(krow[h], false);
                const uint32_t kw[4] = {T.kw[h].x, T.kw[h].y, T.kw[h].z, T.kw[h].w};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const u32x4 kd = P.fused ? dq8<false, true>(kw[kb], R, m0, m4, magic) : dq8<false, false>(kw[kb], R, m0, m4, magic);
                    S[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(kd), qX[kb], S[h], 0, 0, 0);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const RowConst R = row_const(vrow[h], false);
                const uint32_t vw[4] = {T.vw[h].x, T.vw[h].y, T.vw[h].z, T.vw[h].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(lds_u32x4 *)(uintptr_t)(wa[j] + h * 4096) =
                        P.fused ? dq8<false, true>(vw[j], R, m0, m4, magic) : dq8<false, false>(vw[j], R, m0, m4, magic);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[4 * h + e] = S[h][e] * c_;
        }
        if constexpr (TAIL) {   // keys past the range score -inf
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t + 16 * h + 4 * g + e >= w1) sv[4 * h + e] = kNegSentinelI4;
        }
        const float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        if (__any(mx > m_ref + 8.f)) {
            // raise the reference: true running maximum of every q column, accumulators rescaled
            float tm = fmaxf(mx, __shfl_xor(mx, 16));
            tm = fmaxf(tm, __shfl_xor(tm, 32));
            const float m_new = fmaxf(m_ref, tm);
            const float alpha = fast_exp2(m_ref - m_new);
            m_ref = m_new;
            Lp *= alpha;            // (this lane's q column is r: its own factor)
            Zp *= alpha;
            // O rows are q heads 4g + e: fetch their factors from the lanes that own those columns
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __shfl(alpha, 4 * g + e);
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) O[nb][e] *= a;
            }
        }
        // ---- p; the A operand of P.V: P' = p s' (folded: the V rows' scales ride on P) or P itself (exact) ---------
        // (carrying P' as two fp16 operands, hi + lo, was built and measured: the error against the reference's values did
        //  not move — 0.32 of the bar either way — because it is not P' that differs from the reference but the VALUES: the
        //  reference's hmul rounds n s to fp16, up to 2e-3 absolute for |n s| in [4, 8), and this form does not; the second
        //  operand cost 6 % of the step, so it is gone.  profiles/r4_int4_fold.md)
        u32x4 pw;
        {
            float ps[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float p = fast_exp2(sv[i] - m_ref);
                Lp += p;
                if (!exact) {
                    Zp = fmaf(p, vzf[i], Zp);
                    ps[i] = p * vsf[i];
                } else {
                    ps[i] = p;
                }
            }
            pw.x = cvt_pk_f16(ps[0], ps[1]);
            pw.y = cvt_pk_f16(ps[2], ps[3]);
            pw.z = cvt_pk_f16(ps[4], ps[5]);
            pw.w = cvt_pk_f16(ps[6], ps[7]);
        }
        const f16x8_t pA = as_f16x8(pw);
        if (exact) {                // the accumulators take this tile in the true scale ...
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) O[nb] = O[nb] * colscale;
        }
        // ---- O += P' . N' : transpose reads two dim blocks at a time, one batch ahead of the MFMAs -----
        u32x2 va[4], vb[4];
        __builtin_amdgcn_sched_barrier(0);
#define DUO_I4_TR_BATCH(buf, nb0)                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                              \
        uint32_t a_ = ra ^ (uint32_t)(((nb0) + i_) << 5);                           \
        if constexpr (MINW == 4) /* recomputed at every use: eight hoisted copies would be eight registers too many */ \
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a_) : "v"(ra), "v"((uint32_t)(((nb0) + i_) << 5)));           \
        DUO_I4_TR_READ(buf[2 * i_], a_, 0);                                         \
        DUO_I4_TR_READ(buf[2 * i_ + 1], a_, 4096);                                  \
    }
#define DUO_I4_PV(buf, nb0)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                   \
        const u32x4 w_ = {buf[2 * i_].x, buf[2 * i_].y, buf[2 * i_ + 1].x, buf[2 * i_ + 1].y};           \
        O[(nb0) + i_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pA, as_f16x8(w_), O[(nb0) + i_], 0, 0, 0); \
    }
#define DUO_I4_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
        DUO_I4_TR_BATCH(va, 0);
        DUO_I4_TR_BATCH(vb, 2);
        DUO_I4_WAIT(4);
        DUO_I4_PV(va, 0);
        __builtin_amdgcn_sched_barrier(0);
        DUO_I4_TR_BATCH(va, 4);
        DUO_I4_WAIT(4);
        DUO_I4_PV(vb, 2);
        __builtin_amdgcn_sched_barrier(0);
        DUO_I4_TR_BATCH(vb, 6);
        DUO_I4_WAIT(4);
        DUO_I4_PV(va, 4);
        DUO_I4_WAIT(0);
        DUO_I4_PV(vb, 6);
#undef DUO_I4_TR_BATCH
#undef DUO_I4_PV
#undef DUO_I4_WAIT
        if (exact) {                // ... and go back to the folded one (exact powers of two both ways)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) O[nb] = O[nb] * colinv;
        }
    };
    // Main loop: as duo_int4_decode_mfma_kernel — the next tile's 8 loads are issued, unconditionally, before the current
    // tile is touched (behind the last full tile the prefetch re-reads it), so the waits stay counted
    if (w0 < w1) {
        using T_ = std::true_type;
        using F_ = std::false_type;
        const int t_full_end = w0 + (((w1 - w0) >> 5) << 5);
        int t = w0;
        if (t < t_full_end) {
            const int t_last = t_full_end - 32;
            FTile A, B;
            load_full(t, A);
            for (;;) {
                load_full(min(t + 32, t_last), B);
                __builtin_amdgcn_sched_barrier(0);
                process(A, t, F_{});
                t += 32;
                if (t > t_last) break;
                load_full(min(t + 32, t_last), A);
                __builtin_amdgcn_sched_barrier(0);
                process(B, t, F_{});
                t += 32;
                if (t > t_last) break;
            }
        }
        if (t < w1) {
            FTile X;
            load_tail(t, X);
            process(X, t, T_{});
        }
    }

    // ---- combine the 4 waves through LDS ------------------------------------------------------
    Lp += __shfl_xor(Lp, 16);
    Lp += __shfl_xor(Lp, 32);
    Zp += __shfl_xor(Zp, 16);
    Zp += __shfl_xor(Zp, 32);
    __syncthreads();   // every wave is done with its V tile: the LDS is reused for the partials
    float *s_acc = reinterpret_cast<float *>(lds);                 // [4 waves][16 q][128 dims]  = 32 KiB
    __shared__ float s_ml[4][16][3];
    if (g == 0) {
        s_ml[wave][r][0] = m_ref;
        s_ml[wave][r][1] = Lp;
        s_ml[wave][r][2] = Zp;
    }
    // O[nb][e] = (q row 4g + e, LDS column 16 nb + r); LDS column c <-> dim 8 (c >> 3) + perm[c & 7]; columns whose c & 7 is
    // 2, 3, 6 or 7 hold high-nibble dims (values x 16): undone here together with the operands' 2^-24
    {
        const int permv = (0x62734051u >> (4 * (r & 7))) & 15;   // [1,5,0,4,3,7,2,6][r & 7]
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const int dim = 16 * nb + 8 * (r >> 3) + permv;
#pragma unroll
            for (int e = 0; e < 4; ++e) s_acc[(wave * 16 + 4 * g + e) * DUO_HEAD_DIM + dim] = O[nb][e] * colscale;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < P.group * DUO_HEAD_DIM; idx += 256) {
        const int q = idx >> 7;
        const int d = idx & 127;
        float M = s_ml[0][q][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) M = fmaxf(M, s_ml[w][q][0]);
        float Lsum = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = fast_exp2(s_ml[w][q][0] - M);
            Lsum = fmaf(s_ml[w][q][1], sc, Lsum);
            o = fmaf(s_acc[(w * 16 + q) * DUO_HEAD_DIM + d] + s_ml[w][q][2], sc, o);     // + sum p z': the same for every dim
        }
        const int qh = qh0 + q;
        if (splits == 1) {
            P.out[(int64_t)blockIdx.z * P.out_bs + (int64_t)qh * P.out_head_stride + d] = __float2half(o / Lsum);
        } else {
            const int64_t slot = (int64_t)qh * P.max_splits + split;
            P.ws_acc[(int64_t)blockIdx.z * P.ws_row_floats + slot * DUO_HEAD_DIM + d] = o;
            if (d == 0) {
                P.ws_ml[(int64_t)blockIdx.z * P.ws_row_floats + slot * 2 + 0] = M;
                P.ws_ml[(int64_t)blockIdx.z * P.ws_row_floats + slot * 2 + 1] = Lsum;
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
    int64_t ws_row_floats, out_bs;      // batched launch (grid.z = batch row)
};

__global__ __launch_bounds__(256) void duo_int4_decode_merge_kernel(const Int4MergeParams P) {
    int qh = blockIdx.x, splits;
    {
        const int n0 = P.splits[0] > 1 ? P.qh_end[0] - P.qh_begin[0] : 0;
        if (qh < n0) { qh += P.qh_begin[0]; splits = P.splits[0]; }
        else { qh = qh - n0 + P.qh_begin[1]; splits = P.splits[1]; }
    }
    // grid.y = 4: each workgroup merges a 32-dim quarter of the head — 8 threads x 4 dims across, 32 split
    // lanes deep, so even ~200 splits are 6 dependent steps instead of 24
    const int sl = threadIdx.x >> 3, dq = threadIdx.x & 7;
    const int d0 = blockIdx.y * 32 + dq * 4;
    const float *ml = P.ws_ml + (int64_t)blockIdx.z * P.ws_row_floats + (int64_t)qh * P.max_splits * 2;
    const float *ac = P.ws_acc + (int64_t)blockIdx.z * P.ws_row_floats + (int64_t)qh * P.max_splits * DUO_HEAD_DIM + d0;
    __shared__ float sm[32];
    __shared__ float slm[32][8];
    __shared__ f32x4 so[32][8];
    // single pass (as the bf16 merge, duo_decode_merge_task): every split lane folds its own splits against its own
    // running maximum, 4 splits per step with all 8 loads in flight; the 32 lanes are combined through LDS with their maxima
    float m = kNegSentinelI4, Lsum = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int s = sl; s < splits; s += 128) {
        float wm[4], wl[4];
        f32x4 a[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = s + 32 * u < splits;
            const int si = ok[u] ? s + 32 * u : s;
            wm[u] = ml[si * 2];
            wl[u] = ml[si * 2 + 1];
            a[u] = *reinterpret_cast<const f32x4 *>(ac + (int64_t)si * DUO_HEAD_DIM);
        }
        float mx = m;
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, ok[u] ? wm[u] : kNegSentinelI4);
        const float f = fast_exp2(m - mx);
        Lsum *= f;
        o = o * f;
        m = mx;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float wu = ok[u] ? fast_exp2(wm[u] - m) : 0.f;
            Lsum = fmaf(wl[u], wu, Lsum);
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
            const float f = fast_exp2(sm[i] - M);      // lanes without a split: exp2(-huge) = 0
            LL = fmaf(slm[i][dq], f, LL);
            oo = oo + so[i][dq] * f;
        }
        const float inv = 1.f / LL;
        __half h4[4] = {__float2half(oo.x * inv), __float2half(oo.y * inv), __float2half(oo.z * inv),
                        __float2half(oo.w * inv)};
        *reinterpret_cast<u32x2 *>(P.out + (int64_t)blockIdx.z * P.out_bs + (int64_t)qh * P.out_head_stride + d0) = *reinterpret_cast<const u32x2 *>(h4);
    }
}

}  // namespace

// The `_batched` forms (ABI v2): the batch row is grid.y (grid.z for the decode) of the same kernels; the un-batched entry
// points are the n_batch = 1 case.
extern "C" int duo_int4_quantize_batched(const void *src, int32_t src_is_bf16, int64_t src_batch_stride,
                                         int64_t src_token_stride, int64_t src_head_stride, void *q_pool, void *sz_pool,
                                         int64_t pool_batch_stride_rows, int64_t pool_token_stride_rows,
                                         int64_t pool_head_stride_rows, int32_t n_batch, int32_t n_heads,
                                         int32_t n_tokens, int32_t dst_row0, int32_t head_dim, void *stream);
extern "C" int duo_int4_quantize(const void *src, int32_t src_is_bf16, int64_t src_token_stride,
                                 int64_t src_head_stride, void *q_pool, void *sz_pool,
                                 int64_t pool_token_stride_rows, int64_t pool_head_stride_rows,
                                 int32_t n_heads, int32_t n_tokens, int32_t dst_row0, int32_t head_dim,
                                 void *stream) {
    return duo_int4_quantize_batched(src, src_is_bf16, 0, src_token_stride, src_head_stride, q_pool, sz_pool, 0,
                                     pool_token_stride_rows, pool_head_stride_rows, 1, n_heads, n_tokens, dst_row0,
                                     head_dim, stream);
}
extern "C" int duo_int4_quantize_batched(const void *src, int32_t src_is_bf16, int64_t src_batch_stride,
                                         int64_t src_token_stride, int64_t src_head_stride, void *q_pool, void *sz_pool,
                                         int64_t pool_batch_stride_rows, int64_t pool_token_stride_rows,
                                         int64_t pool_head_stride_rows, int32_t n_batch, int32_t n_heads,
                                         int32_t n_tokens, int32_t dst_row0, int32_t head_dim, void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (n_heads <= 0 || n_tokens <= 0 || n_batch <= 0) return 0;
    if (!src || !q_pool || !sz_pool || dst_row0 < 0 || n_batch > 65535 ||
        ((src_token_stride | src_head_stride | src_batch_stride) & 7))
        return DUO_EINVAL;
    QuantParams P{src, src_token_stride, src_head_stride, (uint8_t *)q_pool, (__half *)sz_pool,
                  pool_token_stride_rows, pool_head_stride_rows, n_tokens, n_heads, dst_row0,
                  src_batch_stride, pool_batch_stride_rows};
    const int64_t rows = (int64_t)n_tokens * n_heads;
    dim3 grid((unsigned)((rows + 15) / 16), (unsigned)n_batch), block(256);
    if (src_is_bf16) hipLaunchKernelGGL(duo_int4_quantize_kernel<true>, grid, block, 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(duo_int4_quantize_kernel<false>, grid, block, 0, (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_int4_dequantize_batched_f16(const void *q_pool, const void *sz_pool, int64_t pool_batch_stride_rows,
                                               int64_t pool_token_stride_rows, int64_t pool_head_stride_rows, void *out,
                                               int64_t out_batch_stride, int32_t n_batch, int32_t n_heads,
                                               int32_t n_tokens, int32_t head_dim, int32_t fused, void *stream);
extern "C" int duo_int4_dequantize_f16(const void *q_pool, const void *sz_pool, int64_t pool_token_stride_rows,
                                       int64_t pool_head_stride_rows, void *out, int32_t n_heads,
                                       int32_t n_tokens, int32_t head_dim, int32_t fused, void *stream) {
    return duo_int4_dequantize_batched_f16(q_pool, sz_pool, 0, pool_token_stride_rows, pool_head_stride_rows, out, 0, 1,
                                           n_heads, n_tokens, head_dim, fused, stream);
}
// out: [B][n_tokens, n_heads, 128] fp16, batch rows out_batch_stride elements apart
extern "C" int duo_int4_dequantize_batched_f16(const void *q_pool, const void *sz_pool, int64_t pool_batch_stride_rows,
                                               int64_t pool_token_stride_rows, int64_t pool_head_stride_rows, void *out,
                                               int64_t out_batch_stride, int32_t n_batch, int32_t n_heads,
                                               int32_t n_tokens, int32_t head_dim, int32_t fused, void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (n_heads <= 0 || n_tokens <= 0 || n_batch <= 0) return 0;
    if (!q_pool || !sz_pool || !out || n_batch > 65535 || (out_batch_stride & 7)) return DUO_EINVAL;
    DequantParams P{(const uint8_t *)q_pool, (const __half *)sz_pool, pool_token_stride_rows,
                    pool_head_stride_rows, (__half *)out, n_tokens, n_heads, fused != 0,
                    pool_batch_stride_rows, out_batch_stride};
    const int64_t rows = (int64_t)n_tokens * n_heads;
    hipLaunchKernelGGL(duo_int4_dequantize_kernel, dim3((unsigned)((rows + 15) / 16), (unsigned)n_batch), dim3(256), 0,
                       (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_int4_stream_compress_batched(void *kq, void *ksz, void *vq, void *vsz, int64_t pool_batch_stride_rows,
                                                int64_t pool_token_stride_rows, int64_t pool_head_stride_rows,
                                                int32_t n_batch, int32_t n_heads, int32_t len, int32_t sink,
                                                int32_t recent, int32_t *new_len, void *stream);
extern "C" int duo_int4_stream_compress(void *kq, void *ksz, void *vq, void *vsz, int64_t pool_token_stride_rows,
                                        int64_t pool_head_stride_rows, int32_t n_heads, int32_t len,
                                        int32_t sink, int32_t recent, int32_t *new_len, void *stream) {
    return duo_int4_stream_compress_batched(kq, ksz, vq, vsz, 0, pool_token_stride_rows, pool_head_stride_rows, 1, n_heads,
                                            len, sink, recent, new_len, stream);
}
extern "C" int duo_int4_stream_compress_batched(void *kq, void *ksz, void *vq, void *vsz, int64_t pool_batch_stride_rows,
                                                int64_t pool_token_stride_rows, int64_t pool_head_stride_rows,
                                                int32_t n_batch, int32_t n_heads, int32_t len, int32_t sink,
                                                int32_t recent, int32_t *new_len, void *stream) {
    if (len < 0 || sink < 0 || recent < 0 || n_batch < 0 || n_batch > 65535) return DUO_EINVAL;
    const int W = sink + recent;
    if (new_len) *new_len = len <= W ? len : W;
    if (len <= W || n_heads <= 0 || n_batch == 0) return 0;
    if (!kq || !ksz || !vq || !vsz) return DUO_EINVAL;
    Int4CompressParams P{(uint8_t *)kq, (uint8_t *)vq, (__half *)ksz, (__half *)vsz, pool_token_stride_rows,
                         pool_head_stride_rows, n_heads, len, sink, recent, pool_batch_stride_rows};
    hipLaunchKernelGGL(duo_int4_compress_kernel, dim3(2 * n_heads, n_batch), dim3(256), 0, (hipStream_t)stream, P);
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

extern "C" int duo_attn_decode_int4_batched_f16(const void *q, int64_t q_batch_stride, int64_t q_head_stride, void *out,
                                                int64_t out_batch_stride, int64_t out_head_stride, int32_t n_batch,
                                                int32_t group, const duo_int4_pool *full,
                                                const duo_int4_pool *stream_cls, float scale, int32_t head_dim,
                                                int32_t fused, void *workspace, int64_t workspace_bytes, void *stream);
extern "C" int duo_attn_decode_int4_f16(const void *q, int64_t q_head_stride, void *out, int64_t out_head_stride,
                                        int32_t group, const duo_int4_pool *full, const duo_int4_pool *stream_cls,
                                        float scale, int32_t head_dim, int32_t fused, void *workspace,
                                        int64_t workspace_bytes, void *stream) {
    return duo_attn_decode_int4_batched_f16(q, 0, q_head_stride, out, 0, out_head_stride, 1, group, full, stream_cls, scale,
                                            head_dim, fused, workspace, workspace_bytes, stream);
}
// q / out [B, n_q_heads, 128] fp16; the pools' batch rows are duo_int4_pool::batch_stride_rows apart; grid.z = batch row
extern "C" int duo_attn_decode_int4_batched_f16(const void *q, int64_t q_batch_stride, int64_t q_head_stride, void *out,
                                                int64_t out_batch_stride, int64_t out_head_stride, int32_t n_batch,
                                                int32_t group, const duo_int4_pool *full,
                                                const duo_int4_pool *stream_cls, float scale, int32_t head_dim,
                                                int32_t fused, void *workspace, int64_t workspace_bytes, void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (!q || !out || group <= 0 || n_batch < 0 || n_batch > 65535) return DUO_EINVAL;
    if (n_batch == 0) return 0;
    if (n_batch > 1 && ((q_batch_stride & 7) || (out_batch_stride & 3))) return DUO_EINVAL;
    if (fused < 0 || fused > 3) return DUO_EINVAL;
    Int4DecodeParams P;
    P.fused = fused & 1;
    P.q_bs = q_batch_stride;
    P.out_bs = out_batch_stride;
    P.q = (const __half *)q; P.q_head_stride = q_head_stride;
    P.out = (__half *)out; P.out_head_stride = out_head_stride;
    const duo_int4_pool *src[2] = {full, stream_cls};
    int n_q_heads = 0;
    for (int c = 0; c < 2; ++c) {
        Int4SegDev &S = P.cls[c];
        S = Int4SegDev{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
        if (!src[c] || src[c]->n_kv_heads <= 0) continue;
        const duo_int4_pool &p = *src[c];
        if (!p.k_q || !p.v_q || !p.k_sz || !p.v_sz || p.len <= 0) return DUO_EINVAL;
        S = Int4SegDev{(const uint8_t *)p.k_q, (const uint8_t *)p.v_q, (const __half *)p.k_sz, (const __half *)p.v_sz,
                       p.token_stride_rows, p.head_stride_rows, p.len, p.n_kv_heads, p.q_head_offset, p.batch_stride_rows};
        n_q_heads += p.n_kv_heads * group;
    }
    if (n_q_heads <= 0) return 0;
    P.group = group;
    P.flags = duo_get_debug_flags();
    P.scale_log2e = scale * 1.4426950408889634f;
    const int64_t per_split = (int64_t)n_q_heads * (DUO_HEAD_DIM + 2) * (int64_t)sizeof(float) * n_batch;   // partials per batch row
    const int max_splits = workspace ? (int)std::min<int64_t>(workspace_bytes / per_split, 1024) : 0;
    const int ms = max_splits > 0 ? max_splits : 1;
    // matrix-core form for GQA groups up to 16 q heads per kv head; debug flag bit 4 (and wider groups)
    // select the scalar-FMA kernel
    const bool mfma = group <= 16 && !(duo_get_debug_flags() & 16u);
    // waves per SIMD (= workgroups per CU) the matrix-core kernels are built and launched for.  The dequantising kernel
    // needs 144 VGPRs, so THREE fit (512 / 3 = 170) without a spill: round 4, same box, kernel alone at 1 M context
    // 116 -> 106 us, the 32-layer step at 3.3 M 11.49 -> 11.25 ms (round 3 ran two; four spill).  The folded kernel holds two
    // sets of Q fragments and the tile's (scale, zero) in fp32 — 228 VGPRs — and stays at two.  DUO_INT4_DECODE_WAVES overrides.
    static const int occ_env = [] {
        const char *e = getenv("DUO_INT4_DECODE_WAVES");
        const int x = e ? atoi(e) : 0;
        return (x >= 2 && x <= 4) ? x : 0;
    }();
    bool folded = mfma && fused >= 2 && !(duo_get_debug_flags() & 2048u);
    for (int c = 0; c < 2; ++c)
        if (P.cls[c].n_kv_heads > 0 && P.cls[c].ts != 1) folded = false;       // the folded form needs head-major pools
    const int occ = occ_env ? occ_env : (folded ? 2 : 3);
    // one resident round: 256 CUs x (workgroups per CU = waves per SIMD of the kernel in use)
    const int target = std::max(1, 256 * (mfma ? occ : 2) / n_batch);      // one resident round over all batch rows
    i4_choose_splits(P.cls[1].n_kv_heads, P.cls[1].len, ms, P.cls[0].n_kv_heads > 0 ? P.cls[1].n_kv_heads : target, P.splits[1]);
    i4_choose_splits(P.cls[0].n_kv_heads, P.cls[0].len, ms,
                     std::max(target - P.cls[1].n_kv_heads * P.splits[1], P.cls[0].n_kv_heads), P.splits[0]);
    const int need = std::max(P.splits[0], P.splits[1]);
    if (need > 1 && max_splits < need) return DUO_EWORKSPC;
    P.max_splits = need > 1 ? need : 1;
    P.ws_ml = (float *)workspace;
    P.ws_acc = P.ws_ml ? P.ws_ml + (int64_t)n_q_heads * P.max_splits * 2 : nullptr;
    P.ws_row_floats = (int64_t)n_q_heads * P.max_splits * (DUO_HEAD_DIM + 2);
    P.nblk_full = P.cls[0].n_kv_heads * P.splits[0];
    const int nblk = P.nblk_full + P.cls[1].n_kv_heads * P.splits[1];
    hipStream_t st = (hipStream_t)stream;
    if (mfma) {
        static const int mode = [] {
            const char *e = getenv("DUO_INT4_DECODE_MODE");    // 0: always the 3-instruction form; 1: voted fma form (default)
            const int x = e ? atoi(e) : 1;
            return x == 0 ? 0 : 1;
        }();
        dim3 grid(nblk, 1, n_batch), block(256);
        // fused >= 2: the folded form (no per-element dequantisation; `folded` above) — pools in another layout, and debug
        // bit 11, take the dequantising kernel in the (fused & 1) form
        if (folded) {
            if (occ == 2) hipLaunchKernelGGL(duo_int4_decode_fold_kernel<2>, grid, block, 0, st, P);
            else if (occ == 4) hipLaunchKernelGGL(duo_int4_decode_fold_kernel<4>, grid, block, 0, st, P);
            else hipLaunchKernelGGL(duo_int4_decode_fold_kernel<3>, grid, block, 0, st, P);
        } else {
#define DUO_I4_LAUNCH(W_)                                                                                          \
    do {                                                                                                           \
        if (P.fused) {                                                                                             \
            if (mode == 0) hipLaunchKernelGGL((duo_int4_decode_mfma_kernel<W_, 0, true>), grid, block, 0, st, P);  \
            else hipLaunchKernelGGL((duo_int4_decode_mfma_kernel<W_, 1, true>), grid, block, 0, st, P);            \
        } else {                                                                                                   \
            if (mode == 0) hipLaunchKernelGGL((duo_int4_decode_mfma_kernel<W_, 0, false>), grid, block, 0, st, P); \
            else hipLaunchKernelGGL((duo_int4_decode_mfma_kernel<W_, 1, false>), grid, block, 0, st, P);           \
        }                                                                                                          \
    } while (0)
        if (occ == 2) DUO_I4_LAUNCH(2);
        else if (occ == 4) DUO_I4_LAUNCH(4);
        else DUO_I4_LAUNCH(3);
#undef DUO_I4_LAUNCH
        }
    } else {
        const int gt = (group % 4 == 0) ? 4 : (group % 2 == 0) ? 2 : 1;
        dim3 grid(nblk, group / gt, n_batch), block(256);
        if (gt == 4) hipLaunchKernelGGL(duo_int4_decode_split_kernel<4>, grid, block, 0, st, P);
        else if (gt == 2) hipLaunchKernelGGL(duo_int4_decode_split_kernel<2>, grid, block, 0, st, P);
        else hipLaunchKernelGGL(duo_int4_decode_split_kernel<1>, grid, block, 0, st, P);
    }
    DUO_HIP_CHECK_LAUNCH();
    Int4MergeParams M;
    M.ws_ml = P.ws_ml; M.ws_acc = P.ws_acc; M.out = P.out; M.out_head_stride = out_head_stride;
    M.max_splits = P.max_splits;
    M.ws_row_floats = P.ws_row_floats;
    M.out_bs = out_batch_stride;
    int n_merge = 0;
    for (int c = 0; c < 2; ++c) {
        M.qh_begin[c] = P.cls[c].q_head_offset;
        M.qh_end[c] = P.cls[c].q_head_offset + P.cls[c].n_kv_heads * group;
        M.splits[c] = P.splits[c];
        if (P.splits[c] > 1) n_merge += P.cls[c].n_kv_heads * group;
    }
    // debug flag bit 1: leave the partials unmerged (a HIP-event pair then brackets the split kernel alone)
    if (n_merge > 0 && !(duo_get_debug_flags() & 2u)) {
        hipLaunchKernelGGL(duo_int4_decode_merge_kernel, dim3(n_merge, 4, n_batch), dim3(256), 0, st, M);
        DUO_HIP_CHECK_LAUNCH();
    }
    return 0;
}
