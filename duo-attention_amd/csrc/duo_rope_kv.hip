// duo_rope_kv.hip — the small HBM-bound kernels either side of the attention
// kernels (gfx950): RoPE in place, KV append into a pool, the streaming-pool
// sink+recent compaction, and RMSNorm.  All of them move 16 B per lane.
#include <math.h>
#include <algorithm>
#include "duo_kv_ops.h"

namespace {

// ---------------------------------------------------------------------------
// RoPE, rotate-half convention (flashinfer interleave=False; reference call
// site duo_attn/patch/flashinfer_utils.py:48-56):
//   x'[i]      = x[i]      * cos(a_i) - x[i+64] * sin(a_i)
//   x'[i+64]   = x[i+64]   * cos(a_i) + x[i]    * sin(a_i),   i in [0,64)
//   a_i = float(pos) * inv_freq[i],  inv_freq[i] = theta^(-2i/128) / rope_scale
// inv_freq is computed on the host in double and rounded once to fp32, so the
// fp32 angle is a single IEEE multiply and is reproducible bit-for-bit by the
// oracle (at position 1e6 one fp32 ulp of the angle is already 0.06 rad).
// ---------------------------------------------------------------------------
struct RopeParams {
    bf16_t *q;
    int64_t q_ts, q_hs;
    int32_t n_q_heads;
    bf16_t *k;
    int64_t k_ts, k_hs;
    int32_t n_kv_heads;
    int32_t n_tokens;
    int64_t pos0;
    float inv_freq[64];
    int64_t q_bs, k_bs;      // batched launch (grid.y = batch row, all rows at the same first position)
};

// 64 threads per token: c = tid&7 picks dims [8c,8c+8) and [64+8c,64+8c+8),
// hs = (tid>>3)&7 strides over the heads 8 at a time.  4 tokens per block.
// 16-bit element <-> fp32, by element type (bf16: the static dual-cache path; fp16: the INT4-KV path's model)
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ void unpack8x(const u32x4 &w, float (&f)[8]) {
    if constexpr (F16) {
        const f16x8_t h = __builtin_bit_cast(f16x8_t, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
    } else {
        unpack8f(w, f);
    }
}
template <bool F16>
__device__ __forceinline__ u32x4 pack8x(const float (&f)[8]) {
    if constexpr (F16) {
        f16x8_t h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)f[e];     // round to nearest even
        return __builtin_bit_cast(u32x4, h);
    } else {
        return pack8f(f);
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void duo_rope_kernel(const RopeParams P) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= P.n_tokens) return;
    bf16_t *const qb = P.q + (int64_t)blockIdx.y * P.q_bs, *const kb = P.k + (int64_t)blockIdx.y * P.k_bs;
    const int c = threadIdx.x & 7;
    const int hs = (threadIdx.x >> 3) & 7;
    const float pos = (float)(P.pos0 + tok);
    float cs[8], sn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = pos * P.inv_freq[c * 8 + e];
        sincos_rev(a, sn[e], cs[e]);
    }
    const int n_heads = P.n_q_heads + P.n_kv_heads;
    for (int h = hs; h < n_heads; h += 8) {
        bf16_t *row = h < P.n_q_heads
                          ? qb + (int64_t)tok * P.q_ts + (int64_t)h * P.q_hs
                          : kb + (int64_t)tok * P.k_ts + (int64_t)(h - P.n_q_heads) * P.k_hs;
        u32x4 *plo = reinterpret_cast<u32x4 *>(row + c * 8);
        u32x4 *phi = reinterpret_cast<u32x4 *>(row + 64 + c * 8);
        float lo[8], hi[8], olo[8], ohi[8];
        unpack8x<F16>(*plo, lo);
        unpack8x<F16>(*phi, hi);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            olo[e] = lo[e] * cs[e] - hi[e] * sn[e];
            ohi[e] = hi[e] * cs[e] + lo[e] * sn[e];
        }
        *plo = pack8x<F16>(olo);
        *phi = pack8x<F16>(ohi);
    }
}

// ---------------------------------------------------------------------------
// KV append: pool[dst_row0 + t, h, :] = src[t, h, :] for K and V
// (DuoAttentionStaticKVCache.put_full_kv, static_kv_cache.py:109-125)
// ---------------------------------------------------------------------------
struct AppendParams {
    const bf16_t *ks, *vs;
    int64_t s_ts, s_hs;
    bf16_t *kp, *vp;
    int64_t p_ts, p_hs;
    int32_t n_heads, n_tokens, dst_row0;
    int64_t s_bs, p_bs;      // batched launch (grid.y = batch row)
};

__global__ __launch_bounds__(256) void duo_kv_append_kernel(const AppendParams P) {
    const int64_t total = (int64_t)P.n_tokens * P.n_heads * 16;   // 16-B chunks per (token, head)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i & 15);
        const int64_t th = i >> 4;
        const int h = (int)(th % P.n_heads);
        const int64_t t = th / P.n_heads;
        const int64_t so = t * P.s_ts + (int64_t)h * P.s_hs + ch * 8;
        const int64_t po = (P.dst_row0 + t) * P.p_ts + (int64_t)h * P.p_hs + ch * 8;
        const int64_t sb = (int64_t)blockIdx.y * P.s_bs, pb = (int64_t)blockIdx.y * P.p_bs;
        *reinterpret_cast<u32x4 *>(P.kp + pb + po) = *reinterpret_cast<const u32x4 *>(P.ks + sb + so);
        *reinterpret_cast<u32x4 *>(P.vp + pb + po) = *reinterpret_cast<const u32x4 *>(P.vs + sb + so);
    }
}

__global__ __launch_bounds__(256) void duo_stream_compress_kernel(const CompressParams P) {
    duo_stream_compress_block(P, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------
// RMSNorm (flashinfer.norm.rmsnorm semantics, flashinfer_utils.py:9-16):
// fp32 throughout, one rounding to bf16.  One workgroup per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void duo_rmsnorm_kernel(const bf16_t *x, const bf16_t *w, bf16_t *y,
                                                         int hidden, float eps) {
    const int64_t row = blockIdx.x;
    const bf16_t *xr = x + row * hidden;
    bf16_t *yr = y + row * hidden;
    const int nchunk = hidden >> 3;
    float ss = 0.f;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        float f[8];
        unpack8f(*reinterpret_cast<const u32x4 *>(xr + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float rs = rsqrtf(tot / (float)hidden + eps);
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        float f[8], g[8];
        unpack8f(*reinterpret_cast<const u32x4 *>(xr + c * 8), f);
        unpack8f(*reinterpret_cast<const u32x4 *>(w + c * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * rs * g[e];
        *reinterpret_cast<u32x4 *>(yr + c * 8) = pack8f(f);
    }
}

// SiLU(gate) * up of the SwiGLU MLP on whole chunks (HF LlamaMLP: act_fn(gate_proj(x)) * up_proj(x)) in one pass:
// 2 reads + 1 write per element instead of the 3 reads + 2 writes of a SiLU kernel followed by a multiply.  silu(g) is
// rounded to bf16 before the product, as the module sequence does.  (The q_len == 1 form lives in duo_linear.hip as the
// down_proj prologue.)
__global__ __launch_bounds__(256) void duo_silu_mul_kernel(const bf16_t *g, const bf16_t *u, bf16_t *y, int64_t g_rs,
                                                          int64_t u_rs, int64_t y_rs, int32_t cols8, int64_t n_chunks) {
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * 256) {
        const int64_t row = c / cols8;
        const int col = (int)(c - row * cols8) * 8;
        float f[8], w[8];
        unpack8f(__builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(g + row * g_rs + col)), f);
        unpack8f(__builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(u + row * u_rs + col)), w);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sv = f[e] / (1.f + expf(-f[e]));
            f[e] = __uint_as_float(f32_to_bf16_bits(sv) << 16) * w[e];
        }
        *reinterpret_cast<u32x4 *>(y + row * y_rs + col) = pack8f(f);
    }
}

}  // namespace

template <bool F16>
static int rope_impl(void *q, int64_t q_token_stride, int64_t q_head_stride, int32_t n_q_heads, void *k,
                     int64_t k_token_stride, int64_t k_head_stride, int32_t n_kv_heads, int32_t n_tokens,
                     int64_t pos0, float rope_scale, float rope_theta, int32_t head_dim, void *stream,
                     int32_t n_batch = 1, int64_t q_batch_stride = 0, int64_t k_batch_stride = 0) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (n_tokens <= 0 || n_batch <= 0) return 0;
    if (n_batch > 65535) return DUO_EINVAL;
    if ((n_q_heads > 0 && !q) || (n_kv_heads > 0 && !k) || rope_scale <= 0.f || rope_theta <= 0.f)
        return DUO_EINVAL;
    if (((q_token_stride | q_head_stride | k_token_stride | k_head_stride) & 7) != 0) return DUO_EINVAL;
    RopeParams P;
    P.q = (bf16_t *)q; P.q_ts = q_token_stride; P.q_hs = q_head_stride; P.n_q_heads = n_q_heads;
    P.k = (bf16_t *)k; P.k_ts = k_token_stride; P.k_hs = k_head_stride; P.n_kv_heads = n_kv_heads;
    P.n_tokens = n_tokens;
    P.pos0 = pos0;
    P.q_bs = q_batch_stride;
    P.k_bs = k_batch_stride;
    for (int i = 0; i < 64; ++i)
        P.inv_freq[i] = (float)(pow((double)rope_theta, -2.0 * i / 128.0) / (double)rope_scale);
    hipLaunchKernelGGL(duo_rope_kernel<F16>, dim3((n_tokens + 3) / 4, n_batch), dim3(256), 0, (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_rope_inplace_bf16(void *q, int64_t q_token_stride, int64_t q_head_stride,
                                     int32_t n_q_heads, void *k, int64_t k_token_stride,
                                     int64_t k_head_stride, int32_t n_kv_heads, int32_t n_tokens,
                                     int64_t pos0, float rope_scale, float rope_theta,
                                     int32_t head_dim, void *stream) {
    return rope_impl<false>(q, q_token_stride, q_head_stride, n_q_heads, k, k_token_stride, k_head_stride, n_kv_heads,
                            n_tokens, pos0, rope_scale, rope_theta, head_dim, stream);
}

// Batched RoPE: q [B, S, Hq, 128], k [B, S, Hkv, 128]; row b starts at position pos0[b] (the reference hands
// position_ids[:, 0] to flashinfer, llama.py:350-352).  Rows that share their first position — every reference harness —
// go out as ONE launch (grid.y = batch row); differing positions (left-padded batches) fall back to a launch per row.
template <bool F16>
static int rope_batched(void *q, int64_t q_bs, int64_t q_ts, int64_t q_hs, int32_t n_q_heads, void *k, int64_t k_bs,
                        int64_t k_ts, int64_t k_hs, int32_t n_kv_heads, int32_t n_batch, int32_t n_tokens,
                        const int64_t *pos0, float rope_scale, float rope_theta, int32_t head_dim, void *stream) {
    if (n_batch <= 0) return 0;
    if (!pos0 || ((q_bs | k_bs) & 7) != 0) return DUO_EINVAL;
    bool same = true;
    for (int b = 1; b < n_batch; ++b) same = same && pos0[b] == pos0[0];
    if (same)
        return rope_impl<F16>(q, q_ts, q_hs, n_q_heads, k, k_ts, k_hs, n_kv_heads, n_tokens, pos0[0], rope_scale, rope_theta,
                              head_dim, stream, n_batch, q_bs, k_bs);
    for (int b = 0; b < n_batch; ++b) {
        const int rc = rope_impl<F16>(q ? (bf16_t *)q + b * q_bs : nullptr, q_ts, q_hs, n_q_heads,
                                      k ? (bf16_t *)k + b * k_bs : nullptr, k_ts, k_hs, n_kv_heads, n_tokens, pos0[b],
                                      rope_scale, rope_theta, head_dim, stream);
        if (rc) return rc;
    }
    return 0;
}
extern "C" int duo_rope_inplace_batched_bf16(void *q, int64_t q_batch_stride, int64_t q_token_stride,
                                             int64_t q_head_stride, int32_t n_q_heads, void *k, int64_t k_batch_stride,
                                             int64_t k_token_stride, int64_t k_head_stride, int32_t n_kv_heads,
                                             int32_t n_batch, int32_t n_tokens, const int64_t *pos0, float rope_scale,
                                             float rope_theta, int32_t head_dim, void *stream) {
    return rope_batched<false>(q, q_batch_stride, q_token_stride, q_head_stride, n_q_heads, k, k_batch_stride,
                               k_token_stride, k_head_stride, n_kv_heads, n_batch, n_tokens, pos0, rope_scale, rope_theta,
                               head_dim, stream);
}
extern "C" int duo_rope_inplace_batched_f16(void *q, int64_t q_batch_stride, int64_t q_token_stride,
                                            int64_t q_head_stride, int32_t n_q_heads, void *k, int64_t k_batch_stride,
                                            int64_t k_token_stride, int64_t k_head_stride, int32_t n_kv_heads,
                                            int32_t n_batch, int32_t n_tokens, const int64_t *pos0, float rope_scale,
                                            float rope_theta, int32_t head_dim, void *stream) {
    return rope_batched<true>(q, q_batch_stride, q_token_stride, q_head_stride, n_q_heads, k, k_batch_stride,
                              k_token_stride, k_head_stride, n_kv_heads, n_batch, n_tokens, pos0, rope_scale, rope_theta,
                              head_dim, stream);
}

// fp16 twin: apply_rope_inplace of the INT4-KV path's fp16 model (demo/w8a8kv4_llama.py:207-215)
extern "C" int duo_rope_inplace_f16(void *q, int64_t q_token_stride, int64_t q_head_stride,
                                    int32_t n_q_heads, void *k, int64_t k_token_stride,
                                    int64_t k_head_stride, int32_t n_kv_heads, int32_t n_tokens,
                                    int64_t pos0, float rope_scale, float rope_theta,
                                    int32_t head_dim, void *stream) {
    return rope_impl<true>(q, q_token_stride, q_head_stride, n_q_heads, k, k_token_stride, k_head_stride, n_kv_heads,
                           n_tokens, pos0, rope_scale, rope_theta, head_dim, stream);
}

static int kv_append_impl(const void *k_src, const void *v_src, int64_t src_batch_stride, int64_t src_token_stride,
                          int64_t src_head_stride, void *k_pool, void *v_pool, int64_t pool_batch_stride,
                          int64_t pool_token_stride, int64_t pool_head_stride, int32_t n_batch, int32_t n_heads,
                          int32_t n_tokens, int32_t dst_row0, int32_t head_dim, void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (n_heads <= 0 || n_tokens <= 0 || n_batch <= 0) return 0;
    if (!k_src || !v_src || !k_pool || !v_pool || dst_row0 < 0 || n_batch > 65535) return DUO_EINVAL;
    if (((src_token_stride | src_head_stride | pool_token_stride | pool_head_stride | src_batch_stride |
          pool_batch_stride) & 7) != 0)
        return DUO_EINVAL;
    AppendParams P{(const bf16_t *)k_src, (const bf16_t *)v_src, src_token_stride, src_head_stride,
                   (bf16_t *)k_pool, (bf16_t *)v_pool, pool_token_stride, pool_head_stride,
                   n_heads, n_tokens, dst_row0, src_batch_stride, pool_batch_stride};
    const int64_t total = (int64_t)n_tokens * n_heads * 16;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(duo_kv_append_kernel, dim3(grid, n_batch), dim3(256), 0, (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}
extern "C" int duo_kv_append_bf16(const void *k_src, const void *v_src, int64_t src_token_stride,
                                  int64_t src_head_stride, void *k_pool, void *v_pool,
                                  int64_t pool_token_stride, int64_t pool_head_stride,
                                  int32_t n_heads, int32_t n_tokens, int32_t dst_row0,
                                  int32_t head_dim, void *stream) {
    return kv_append_impl(k_src, v_src, 0, src_token_stride, src_head_stride, k_pool, v_pool, 0, pool_token_stride,
                          pool_head_stride, 1, n_heads, n_tokens, dst_row0, head_dim, stream);
}
// batched: src [B, S, h, 128], pools [B, T, h, 128] (any strides): one launch, grid.y = batch row
extern "C" int duo_kv_append_batched_bf16(const void *k_src, const void *v_src, int64_t src_batch_stride,
                                          int64_t src_token_stride, int64_t src_head_stride, void *k_pool, void *v_pool,
                                          int64_t pool_batch_stride, int64_t pool_token_stride, int64_t pool_head_stride,
                                          int32_t n_batch, int32_t n_heads, int32_t n_tokens, int32_t dst_row0,
                                          int32_t head_dim, void *stream) {
    return kv_append_impl(k_src, v_src, src_batch_stride, src_token_stride, src_head_stride, k_pool, v_pool,
                          pool_batch_stride, pool_token_stride, pool_head_stride, n_batch, n_heads, n_tokens, dst_row0,
                          head_dim, stream);
}

static int stream_compress_impl(void *k_pool, void *v_pool, int64_t pool_batch_stride, int64_t pool_token_stride,
                                int64_t pool_head_stride, const void *k_new, const void *v_new,
                                int64_t new_batch_stride, int64_t new_token_stride, int64_t new_head_stride,
                                int32_t n_batch, int32_t n_heads, int32_t cur_len, int32_t n_new, int32_t sink,
                                int32_t recent, int32_t head_dim, int32_t *new_len, void *stream);
extern "C" int duo_stream_compress_bf16(void *k_pool, void *v_pool, int64_t pool_token_stride,
                                        int64_t pool_head_stride, const void *k_new, const void *v_new,
                                        int64_t new_token_stride, int64_t new_head_stride,
                                        int32_t n_heads, int32_t cur_len, int32_t n_new, int32_t sink,
                                        int32_t recent, int32_t head_dim, int32_t *new_len, void *stream) {
    return stream_compress_impl(k_pool, v_pool, 0, pool_token_stride, pool_head_stride, k_new, v_new, 0, new_token_stride,
                                new_head_stride, 1, n_heads, cur_len, n_new, sink, recent, head_dim, new_len, stream);
}
// batched: pools [B, W, h, 128], new rows [B, S, h, 128]; every row at the same length (static_kv_cache.py:127-167)
extern "C" int duo_stream_compress_batched_bf16(void *k_pool, void *v_pool, int64_t pool_batch_stride,
                                                int64_t pool_token_stride, int64_t pool_head_stride, const void *k_new,
                                                const void *v_new, int64_t new_batch_stride, int64_t new_token_stride,
                                                int64_t new_head_stride, int32_t n_batch, int32_t n_heads,
                                                int32_t cur_len, int32_t n_new, int32_t sink, int32_t recent,
                                                int32_t head_dim, int32_t *new_len, void *stream) {
    return stream_compress_impl(k_pool, v_pool, pool_batch_stride, pool_token_stride, pool_head_stride, k_new, v_new,
                                new_batch_stride, new_token_stride, new_head_stride, n_batch, n_heads, cur_len, n_new, sink,
                                recent, head_dim, new_len, stream);
}
static int stream_compress_impl(void *k_pool, void *v_pool, int64_t pool_batch_stride, int64_t pool_token_stride,
                                int64_t pool_head_stride, const void *k_new, const void *v_new,
                                int64_t new_batch_stride, int64_t new_token_stride, int64_t new_head_stride,
                                int32_t n_batch, int32_t n_heads, int32_t cur_len, int32_t n_new, int32_t sink,
                                int32_t recent, int32_t head_dim, int32_t *new_len, void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (cur_len < 0 || n_new < 0 || sink < 0 || recent < 0 || cur_len > sink + recent) return DUO_EINVAL;
    const int T = cur_len + n_new, W = sink + recent;
    if (new_len) *new_len = T <= W ? T : W;
    if (n_heads <= 0 || n_new == 0 || n_batch <= 0) return 0;
    if (!k_pool || !v_pool || !k_new || !v_new || n_batch > 65535) return DUO_EINVAL;
    if (((pool_token_stride | pool_head_stride | new_token_stride | new_head_stride | pool_batch_stride |
          new_batch_stride) & 7) != 0)
        return DUO_EINVAL;
    CompressParams P{};
    P.kp = (bf16_t *)k_pool; P.vp = (bf16_t *)v_pool; P.p_ts = pool_token_stride; P.p_hs = pool_head_stride;
    P.kn = (const bf16_t *)k_new; P.vn = (const bf16_t *)v_new; P.n_ts = new_token_stride; P.n_hs = new_head_stride;
    P.n_heads = n_heads; P.cur = cur_len; P.n_new = n_new; P.sink = sink; P.recent = recent;
    P.p_bs = pool_batch_stride; P.n_bs = new_batch_stride;
    hipLaunchKernelGGL(duo_stream_compress_kernel, dim3(n_heads * 2, n_batch), dim3(256), 0, (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_rmsnorm_bf16(const void *x, const void *w, void *y, int64_t n_rows, int32_t hidden,
                                float eps, void *stream) {
    if (n_rows <= 0) return 0;
    if (!x || !w || !y || hidden <= 0 || (hidden & 7)) return DUO_EINVAL;
    hipLaunchKernelGGL(duo_rmsnorm_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t *)x, (const bf16_t *)w, (bf16_t *)y, hidden, eps);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_silu_mul_bf16(const void *gate, int64_t gate_row_stride, const void *up, int64_t up_row_stride, void *y,
                                 int64_t y_row_stride, int64_t n_rows, int32_t n_cols, void *stream) {
    if (n_rows <= 0 || n_cols == 0) return 0;
    if (!gate || !up || !y || n_cols < 0 || (n_cols & 7) || ((gate_row_stride | up_row_stride | y_row_stride) & 7) ||
        (((uintptr_t)gate | (uintptr_t)up | (uintptr_t)y) & 15))
        return DUO_EINVAL;
    const int64_t n_chunks = n_rows * (n_cols >> 3);
    const int64_t blocks = (n_chunks + 255) / 256;
    hipLaunchKernelGGL(duo_silu_mul_kernel, dim3((unsigned)(blocks < 256 * 64 ? blocks : 256 * 64)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t *)gate, (const bf16_t *)up, (bf16_t *)y, gate_row_stride,
                       up_row_stride, y_row_stride, n_cols >> 3, n_chunks);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_abi_version(void) { return DUO_ABI_VERSION; }
extern "C" const char *duo_target_arch(void) { return "gfx950"; }
extern "C" const char *duo_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case DUO_EINVAL: return "DUO_EINVAL: bad pointer, size or stride";
        case DUO_EHEADDIM: return "DUO_EHEADDIM: head_dim must be 128";
        case DUO_EGROUP: return "DUO_EGROUP: unsupported GQA group size";
        case DUO_EWORKSPC: return "DUO_EWORKSPC: decode workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown duo error";
    }
}
