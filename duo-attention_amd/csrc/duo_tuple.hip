// The tuple-cache path's own small kernels (gfx950): duo_tuple_decode_prep_bf16 (the decode step's data movement),
// duo_rope_hf_inplace_bf16 (HF rotary on prefill chunks), duo_rmsnorm_hf_bf16 (HF's two-rounding RMSNorm).
//
// Reference: llama_duo_attention_forward_one_way_reordered, duo_attn/patch/llama.py:146-306, at q_len == 1.  Between the
// projections and the two flash_attn_func calls the reference issues ~20 small torch kernels per layer and token: the HF
// rotary (6 elementwise ops, :177-184), four torch.cat of cache ++ new row (:202-223), the sink/recent truncation (two
// clone + copy_ pairs, :273-290) and the two K-on-V stacks (:292-301) — and the retrieval cat re-writes the WHOLE
// context.  Here it is one launch:
//
//   blocks [0, nrot)        one 16-lane row per q head / k head / v head (256-B rows, 16 B per lane):
//                             q, k   HF rotary in place, in torch's bf16 arithmetic (three roundings per element);
//                             k, v   of a retrieval head: the rotated k row and the v row land at row full_len of the arena;
//                             k, v   of a streaming head: they land at the row the truncation gives the new token in
//                                    the NEW streaming cache
//   blocks [nrot, ...)      (streaming head, K|V) x 64-row slabs: the old rows the truncation keeps, copied to their new
//                           row (out of place — a tuple handed out earlier is never rewritten)
//
// HBM-bound byte movement, a few hundred KB per launch: the point is the launch count, not the bandwidth.
#include "duo_common.h"

namespace {

struct TuplePrepParams {
    bf16_t *q;
    int64_t q_hs;
    int32_t n_q, n_kv, n_full;
    bf16_t *k;
    const bf16_t *v;
    int64_t kv_hs;
    const bf16_t *cos_row, *sin_row;
    bf16_t *full_k, *full_v;
    int64_t full_ts, full_hs;
    int32_t full_len;
    const bf16_t *sk_src, *sv_src;
    int64_t src_ts, src_hs;
    bf16_t *sk_dst, *sv_dst;
    int64_t dst_ts, dst_hs;
    int32_t str_len, sink, recent, out_len;
    int32_t nrot;       // blocks of the rotate / place part
    int32_t slabs;      // 64-row slabs per (streaming head, K|V)
};

__device__ __forceinline__ float bf16r(float x) { return __uint_as_float(f32_to_bf16_bits(x) << 16); }

// source row (index into old ++ new) of destination row r of the truncated streaming cache (llama.py:273-290)
__device__ __forceinline__ int trunc_src(int r, int T, int sink, int recent) {
    return (T <= sink + recent || r < sink) ? r : T - recent + (r - sink);
}

__global__ __launch_bounds__(256) void duo_tuple_decode_prep_kernel(const TuplePrepParams P) {
    const int sub = threadIdx.x & 15;            // 16-byte slice of the 256-byte row
    const int T = P.str_len + 1;
    if ((int)blockIdx.x < P.nrot) {
        const int row = blockIdx.x * 16 + (threadIdx.x >> 4);      // q heads, then k heads, then v heads
        if (row >= P.n_q + 2 * P.n_kv) return;
        const bool is_q = row < P.n_q, is_k = !is_q && row < P.n_q + P.n_kv;
        const int h = is_q ? row : is_k ? row - P.n_q : row - P.n_q - P.n_kv;
        u32x4 outw;
        if (is_q || is_k) {
            bf16_t *x = is_q ? P.q + (int64_t)h * P.q_hs : P.k + (int64_t)h * P.kv_hs;
            const u32x4 own = *reinterpret_cast<const u32x4 *>(x + sub * 8);
            const u32x4 par = *reinterpret_cast<const u32x4 *>(x + (sub ^ 8) * 8);     // rotate_half partner: the other half
            const u32x4 cw = *reinterpret_cast<const u32x4 *>(P.cos_row + sub * 8);
            const u32x4 sw = *reinterpret_cast<const u32x4 *>(P.sin_row + sub * 8);
            const uint32_t ow[4] = {own.x, own.y, own.z, own.w}, pw[4] = {par.x, par.y, par.z, par.w};
            const uint32_t c4[4] = {cw.x, cw.y, cw.z, cw.w}, s4[4] = {sw.x, sw.y, sw.z, sw.w};
            const float sgn = sub < 8 ? -1.f : 1.f;          // rotate_half(x) = cat(-x2, x1)
            uint32_t r4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // torch: (x * cos) -> bf16, (rotate_half(x) * sin) -> bf16, sum -> bf16  (negation is exact)
                const float lo = bf16r(bf16r(bf16_lo(ow[e]) * bf16_lo(c4[e])) + bf16r(sgn * bf16_lo(pw[e]) * bf16_lo(s4[e])));
                const float hi = bf16r(bf16r(bf16_hi(ow[e]) * bf16_hi(c4[e])) + bf16r(sgn * bf16_hi(pw[e]) * bf16_hi(s4[e])));
                r4[e] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
            }
            outw.x = r4[0]; outw.y = r4[1]; outw.z = r4[2]; outw.w = r4[3];
            // in place: every lane of the row has its two slices in registers before any lane stores (one wave, one store
            // instruction behind the loads' wait)
            *reinterpret_cast<u32x4 *>(x + sub * 8) = outw;
            if (is_q) return;
        } else {
            outw = *reinterpret_cast<const u32x4 *>(P.v + (int64_t)h * P.kv_hs + sub * 8);
        }
        // the new row of kv head h joins its cache
        if (h < P.n_full) {
            bf16_t *dst = (is_k ? P.full_k : P.full_v) + (int64_t)P.full_len * P.full_ts + (int64_t)h * P.full_hs;
            *reinterpret_cast<u32x4 *>(dst + sub * 8) = outw;
        } else if (P.out_len > 0) {
            // destination row of the new token (source index T - 1): the last row unless sink >= T (it then keeps its place)
            const int W = P.sink + P.recent;
            const int r = T <= W ? T - 1 : (P.recent > 0 ? W - 1 : -1);
            if (r >= 0) {
                bf16_t *dst = (is_k ? P.sk_dst : P.sv_dst) + (int64_t)r * P.dst_ts + (int64_t)(h - P.n_full) * P.dst_hs;
                *reinterpret_cast<u32x4 *>(dst + sub * 8) = outw;
            }
        }
        return;
    }
    // ---- the old streaming rows that survive, 64 destination rows per block ----
    const int b = blockIdx.x - P.nrot;
    const int slab = b % P.slabs, hv = b / P.slabs;          // hv = 2 * head + (0: K, 1: V)
    const int head = hv >> 1;
    const bf16_t *src = (hv & 1) ? P.sv_src : P.sk_src;
    bf16_t *dst = (hv & 1) ? P.sv_dst : P.sk_dst;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = slab * 64 + it * 16 + (threadIdx.x >> 4);
        if (r >= P.out_len) continue;
        const int xi = trunc_src(r, T, P.sink, P.recent);
        if (xi >= P.str_len) continue;                       // the new token's row: written by the rotate part
        const u32x4 w = __builtin_nontemporal_load(
            reinterpret_cast<const u32x4 *>(src + (int64_t)xi * P.src_ts + (int64_t)head * P.src_hs + sub * 8));
        *reinterpret_cast<u32x4 *>(dst + (int64_t)r * P.dst_ts + (int64_t)head * P.dst_hs + sub * 8) = w;
    }
}

// ---- HF rotary on whole chunks (the tuple forward's prefill calls, llama.py:177-184): every (token, head) row of q and k in
//      place, cos / sin rows [n_tokens, 128] bf16 as model.rotary_emb hands them over; torch's bf16 arithmetic as above.
//      64 threads per token: c = tid & 7 owns dims [8c, 8c+8) and [64+8c, 64+8c+8), hs = (tid >> 3) & 7 strides over the heads.
struct RopeHfParams {
    bf16_t *q, *k;
    int64_t q_ts, q_hs, k_ts, k_hs;
    const bf16_t *cos_rows, *sin_rows;
    int64_t cs_ts;
    int32_t n_q, n_kv, n_tokens;
};

__global__ __launch_bounds__(256) void duo_rope_hf_kernel(const RopeHfParams P) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= P.n_tokens) return;
    const int c = threadIdx.x & 7, hs = (threadIdx.x >> 3) & 7;
    const bf16_t *cr = P.cos_rows + (int64_t)tok * P.cs_ts, *sr = P.sin_rows + (int64_t)tok * P.cs_ts;
    const u32x4 c_lo = *reinterpret_cast<const u32x4 *>(cr + c * 8), c_hi = *reinterpret_cast<const u32x4 *>(cr + 64 + c * 8);
    const u32x4 s_lo = *reinterpret_cast<const u32x4 *>(sr + c * 8), s_hi = *reinterpret_cast<const u32x4 *>(sr + 64 + c * 8);
    const uint32_t cl[4] = {c_lo.x, c_lo.y, c_lo.z, c_lo.w}, ch[4] = {c_hi.x, c_hi.y, c_hi.z, c_hi.w};
    const uint32_t sl[4] = {s_lo.x, s_lo.y, s_lo.z, s_lo.w}, sh[4] = {s_hi.x, s_hi.y, s_hi.z, s_hi.w};
    auto rot = [](float x, float cosv, float partner, float sinv) {      // bf16(bf16(x cos) + bf16(partner sin))
        return bf16r(bf16r(x * cosv) + bf16r(partner * sinv));
    };
    const int n_heads = P.n_q + P.n_kv;
    for (int h = hs; h < n_heads; h += 8) {
        bf16_t *row = h < P.n_q ? P.q + (int64_t)tok * P.q_ts + (int64_t)h * P.q_hs
                                : P.k + (int64_t)tok * P.k_ts + (int64_t)(h - P.n_q) * P.k_hs;
        u32x4 *plo = reinterpret_cast<u32x4 *>(row + c * 8), *phi = reinterpret_cast<u32x4 *>(row + 64 + c * 8);
        const u32x4 lo = *plo, hi = *phi;
        const uint32_t lw[4] = {lo.x, lo.y, lo.z, lo.w}, hw[4] = {hi.x, hi.y, hi.z, hi.w};
        uint32_t ol[4], oh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // first half: x1 cos - x2 sin (rotate_half = cat(-x2, x1)); second half: x2 cos + x1 sin
            const float a0 = rot(bf16_lo(lw[e]), bf16_lo(cl[e]), -bf16_lo(hw[e]), bf16_lo(sl[e]));
            const float a1 = rot(bf16_hi(lw[e]), bf16_hi(cl[e]), -bf16_hi(hw[e]), bf16_hi(sl[e]));
            const float b0 = rot(bf16_lo(hw[e]), bf16_lo(ch[e]), bf16_lo(lw[e]), bf16_lo(sh[e]));
            const float b1 = rot(bf16_hi(hw[e]), bf16_hi(ch[e]), bf16_hi(lw[e]), bf16_hi(sh[e]));
            ol[e] = (__float_as_uint(a0) >> 16) | (__float_as_uint(a1) & 0xffff0000u);
            oh[e] = (__float_as_uint(b0) >> 16) | (__float_as_uint(b1) & 0xffff0000u);
        }
        *plo = u32x4{ol[0], ol[1], ol[2], ol[3]};
        *phi = u32x4{oh[0], oh[1], oh[2], oh[3]};
    }
}

// ---- HuggingFace's LlamaRMSNorm / MistralRMSNorm.forward on [rows, hidden] bf16 (the tuple path keeps HF's norm modules):
//      y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))) — the normalised activations are rounded to bf16 BEFORE the weight
//      multiply (duo_rmsnorm_kernel is flashinfer's one-rounding form).  One workgroup per row, fp32 statistics.
__global__ __launch_bounds__(256) void duo_rmsnorm_hf_kernel(const bf16_t *x, const bf16_t *w, bf16_t *y, int hidden, float eps) {
    const int64_t row = blockIdx.x;
    const bf16_t *xr = x + row * hidden;
    bf16_t *yr = y + row * hidden;
    const int nchunk = hidden >> 3;
    float ss = 0.f;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(xr + c * 8);
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ss = fmaf(bf16_lo(vw[e]), bf16_lo(vw[e]), ss);
            ss = fmaf(bf16_hi(vw[e]), bf16_hi(vw[e]), ss);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)hidden + eps);
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(xr + c * 8), g = *reinterpret_cast<const u32x4 *>(w + c * 8);
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, gw[4] = {g.x, g.y, g.z, g.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bf16r(bf16r(bf16_lo(vw[e]) * rs) * bf16_lo(gw[e]));
            const float b = bf16r(bf16r(bf16_hi(vw[e]) * rs) * bf16_hi(gw[e]));
            o[e] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
        }
        *reinterpret_cast<u32x4 *>(yr + c * 8) = u32x4{o[0], o[1], o[2], o[3]};
    }
}

}  // namespace

extern "C" int duo_rope_hf_inplace_bf16(void *q, int64_t q_token_stride, int64_t q_head_stride, int32_t n_q_heads, void *k,
                                        int64_t k_token_stride, int64_t k_head_stride, int32_t n_kv_heads, int32_t n_tokens,
                                        const void *cos_rows, const void *sin_rows, int64_t cos_sin_token_stride,
                                        int32_t head_dim, void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (n_tokens <= 0 || n_q_heads + n_kv_heads <= 0) return 0;
    if ((n_q_heads > 0 && !q) || (n_kv_heads > 0 && !k) || !cos_rows || !sin_rows || n_q_heads < 0 || n_kv_heads < 0) return DUO_EINVAL;
    if (((q_token_stride | q_head_stride | k_token_stride | k_head_stride | cos_sin_token_stride) & 7) != 0) return DUO_EINVAL;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos_rows | (uintptr_t)sin_rows) & 15) != 0) return DUO_EINVAL;
    RopeHfParams P{(bf16_t *)q, (bf16_t *)k, q_token_stride, q_head_stride, k_token_stride, k_head_stride,
                   (const bf16_t *)cos_rows, (const bf16_t *)sin_rows, cos_sin_token_stride, n_q_heads, n_kv_heads, n_tokens};
    hipLaunchKernelGGL(duo_rope_hf_kernel, dim3((n_tokens + 3) / 4), dim3(256), 0, (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_rmsnorm_hf_bf16(const void *x, const void *w, void *y, int64_t n_rows, int32_t hidden, float eps,
                                   void *stream) {
    if (n_rows <= 0) return 0;
    if (!x || !w || !y || hidden <= 0 || (hidden & 7) || n_rows > 0x7fffffffll) return DUO_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) != 0) return DUO_EINVAL;
    hipLaunchKernelGGL(duo_rmsnorm_hf_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x,
                       (const bf16_t *)w, (bf16_t *)y, hidden, eps);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int duo_tuple_decode_prep_bf16(const duo_tuple_decode_args *a, int32_t *new_stream_len, void *stream) {
    if (!a) return DUO_EINVAL;
    if (a->head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    const int nq = a->n_q_heads, nkv = a->n_kv_heads, nf = a->n_full, ns = nkv - nf;
    if (!a->q || !a->k || !a->v || !a->cos_row || !a->sin_row || nq <= 0 || nkv <= 0 || nf < 0 || ns < 0) return DUO_EINVAL;
    if (a->sink < 0 || a->recent < 0 || a->str_len < 0) return DUO_EINVAL;
    if (nf > 0 && (!a->full_k || !a->full_v || a->full_len < 0 || a->full_len + 1 > a->full_capacity)) return DUO_EINVAL;
    const int W = a->sink + a->recent, T = a->str_len + 1;
    const int out_len = T <= W ? T : W;
    if (new_stream_len) *new_stream_len = out_len;
    if (ns > 0 && out_len > 0 && (!a->str_k_dst || !a->str_v_dst)) return DUO_EINVAL;
    if (ns > 0 && a->str_len > 0 && (!a->str_k_src || !a->str_v_src)) return DUO_EINVAL;
    if (ns > 0 && a->str_len > 0 && (a->str_k_src == a->str_k_dst || a->str_v_src == a->str_v_dst)) return DUO_EINVAL;   // out of place
    if (((a->q_head_stride | a->kv_head_stride | a->full_token_stride | a->full_head_stride | a->src_token_stride |
          a->src_head_stride | a->dst_token_stride | a->dst_head_stride) & 7) != 0)
        return DUO_EINVAL;
    if ((((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->cos_row | (uintptr_t)a->sin_row |
          (uintptr_t)a->full_k | (uintptr_t)a->full_v | (uintptr_t)a->str_k_src | (uintptr_t)a->str_v_src |
          (uintptr_t)a->str_k_dst | (uintptr_t)a->str_v_dst) & 15) != 0)
        return DUO_EINVAL;
    TuplePrepParams P;
    P.q = (bf16_t *)a->q; P.q_hs = a->q_head_stride; P.n_q = nq; P.n_kv = nkv; P.n_full = nf;
    P.k = (bf16_t *)a->k; P.v = (const bf16_t *)a->v; P.kv_hs = a->kv_head_stride;
    P.cos_row = (const bf16_t *)a->cos_row; P.sin_row = (const bf16_t *)a->sin_row;
    P.full_k = (bf16_t *)a->full_k; P.full_v = (bf16_t *)a->full_v; P.full_ts = a->full_token_stride; P.full_hs = a->full_head_stride;
    P.full_len = a->full_len;
    P.sk_src = (const bf16_t *)a->str_k_src; P.sv_src = (const bf16_t *)a->str_v_src; P.src_ts = a->src_token_stride; P.src_hs = a->src_head_stride;
    P.sk_dst = (bf16_t *)a->str_k_dst; P.sv_dst = (bf16_t *)a->str_v_dst; P.dst_ts = a->dst_token_stride; P.dst_hs = a->dst_head_stride;
    P.str_len = a->str_len; P.sink = a->sink; P.recent = a->recent; P.out_len = ns > 0 ? out_len : 0;
    P.nrot = (nq + 2 * nkv + 15) / 16;
    P.slabs = (P.out_len + 63) / 64;
    const int ncopy = (ns > 0 && a->str_len > 0) ? 2 * ns * P.slabs : 0;
    if (P.slabs == 0) P.slabs = 1;
    hipLaunchKernelGGL(duo_tuple_decode_prep_kernel, dim3(P.nrot + ncopy), dim3(256), 0, (hipStream_t)stream, P);
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}
