// Token-row linear layers of the decode step (gfx950).
//
// At q_len == 1 the projections either side of the attention op — q/k/v_proj and o_proj inside the reference's static
// forward (duo_attn/patch/llama.py:332-340, :430-432) and the MLP + norms of its decoder layer
// (duo_attn/patch/static_kv_cache.py:482-537) — are matrix-VECTOR products: every weight byte is read once per token,
// 33-117 MB per call, nothing to reuse.  They are HBM-bound byte streaming, not GEMM work, and the library GEMM kernels
// that serve them at M = 1 run at 0.6-4.9 TB/s (profiles/r3_model_level_kernels.md).  This file streams the weight rows
// once with every wave of the chip holding loads in flight, keeps the token rows in LDS, and folds the element-wise
// neighbours of each product into it:
//
//   prologue (while the first weight loads are in flight; the token rows land in LDS as packed bf16)
//     PRO_NONE   x as given
//     PRO_NORM   x <- RMSNorm(x; norm_weight, eps)             (flashinfer.norm.rmsnorm, one rounding to bf16)
//     PRO_SILU   x <- silu(x) * x2                              (LlamaMLP: act_fn(gate_proj(h)) * up_proj(h))
//   product      y[b, n] = sum_k W[n, k] * x[b, k] + bias[n]   fp32 accumulate, one rounding to bf16; up to three weight
//                blocks whose outputs are concatenated (q | k | v, gate | up)
//   epilogue     y <- y + residual                              (the decoder layer's residual adds)
//
// Every value the unfused module sequence materialises as a bf16 tensor is rounded to bf16 here at the same point
// (normalised x, silu(g), silu(g) * u, the linear output before the residual add), so the fused step differs from the
// module-by-module one only by the summation order inside a dot product.
//
// Work split: one weight row = one wave (64 lanes x 16 bytes per load instruction = 1 KiB of the row), rows dealt
// round-robin to all waves of the grid; a wave keeps two groups of four loads in flight across row boundaries (the next
// group is issued unconditionally — behind the wave's last group it re-reads that group — so the waits are counted).
// The grid is chosen so that every wave gets the same number of rows (token_linear_grid below).
#include <type_traits>
#include "duo_common.h"
#include "duo_kv_ops.h"

namespace {

constexpr int kLinG = 4;            // weight loads per group (1 KiB each per wave)
constexpr int kLinMaxRows = 4;      // token rows per call (DUO_TOKEN_LINEAR_MAX_ROWS)

struct LinSegDev {
    const bf16_t *w;
    const bf16_t *bias;
    int64_t rs;
    int32_t n;
};

struct TokenLinearParams {
    const bf16_t *x, *x2;
    int64_t x_rs;
    int32_t K, kpad, gpr;           // gpr: groups of kLinG x 512 elements per weight row, EVEN (kpad = gpr * kLinG * 512)
    LinSegDev seg[3];
    int32_t n_total;
    const bf16_t *norm_w;
    float eps;
    const bf16_t *res;
    int64_t res_rs;
    bf16_t *y;
    int64_t y_rs;
};

enum { PRO_NONE = 0, PRO_NORM = 1, PRO_SILU = 2 };

// the group's four loads have landed when at most N newer vector-memory operations are outstanding (loads return in order)
template <int N>
__device__ __forceinline__ void lin_wait(u32x4 (&buf)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]) : "n"(N));
}

__device__ __forceinline__ float wave_sum(float x) {
    x = row16_allreduce_sum(x);                 // every lane of a 16-lane row holds the row's total
    const int xi = __float_as_int(x);
    return (__int_as_float(__builtin_amdgcn_readlane(xi, 0)) + __int_as_float(__builtin_amdgcn_readlane(xi, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(xi, 32)) + __int_as_float(__builtin_amdgcn_readlane(xi, 48)));
}

template <int B, int PRO>
__global__ __launch_bounds__(1024) void duo_token_linear_kernel(const TokenLinearParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t xs[];      // [B][kpad / 2] packed bf16 pairs
    __shared__ float red[kLinMaxRows][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TW = gridDim.x * nw, gw = blockIdx.x * nw + wave;
    const int my_rows = gw < P.n_total ? (P.n_total - gw + TW - 1) / TW : 0;
    const int gpr = P.gpr, K = P.K;
    const int xrow = P.kpad >> 1;                                       // dwords per staged token row

    // the three weight blocks' addressing as plain scalars (indexing P.seg[] with a computed index makes the compiler
    // re-read the kernel-argument segment through a computed address in front of every group)
    uint64_t w0 = (uint64_t)P.seg[0].w, w1 = (uint64_t)P.seg[1].w, w2 = (uint64_t)P.seg[2].w;
    uint64_t rb0 = (uint64_t)P.seg[0].rs * 2u, rb1 = (uint64_t)P.seg[1].rs * 2u, rb2 = (uint64_t)P.seg[2].rs * 2u;   // row strides in bytes
    int n0 = P.seg[0].n, n1 = n0 + P.seg[1].n;
    asm volatile("" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(rb0), "+s"(rb1), "+s"(rb2), "+s"(n0), "+s"(n1));   // (values, not addresses to select between)
    auto row_ptr = [&](int n) __attribute__((always_inline)) -> uint64_t {
        uint64_t w = w2, rb = rb2;
        int nb = n - n1;
        if (n < n1) { w = w1; rb = rb1; nb = n - n0; }
        if (n < n0) { w = w0; rb = rb0; nb = n; }
        return w + (uint64_t)(uint32_t)nb * rb;
    };
    // group g of this wave's i-th row (cursor clamped to the wave's last row: the loads behind the last group are never
    // consumed); per lane 8 elements at k0, clamped into the row — the staged x is zero there
    auto issue = [&](int i, int g, u32x4 (&buf)[kLinG]) __attribute__((always_inline)) {
        const int n = min(gw + max(min(i, my_rows - 1), 0) * TW, P.n_total - 1);     // (a wave without rows reads a valid row, never consumed)
        const uint64_t wr = row_ptr(n);
#pragma unroll
        for (int j = 0; j < kLinG; ++j) {
            const int k0 = ((g * kLinG + j) * 64 + lane) * 8;
            const uint32_t off = (uint32_t)min(k0, K - 8) * 2u;
            // (asm: the stream's loads are outside the compiler's wait bookkeeping on purpose — its loop-carried
            //  analysis put a vmcnt(0) at the loop head, i.e. it let the prefetch drain every second group; the waits
            //  are the explicit counted ones in lin_wait below)
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(buf[j]) : "v"(off), "s"(wr));
        }
    };

    u32x4 bufA[kLinG], bufB[kLinG];
    const int T = my_rows * gpr;
    int ii = 0, ig = 0;
    auto adv = [&](int &i, int &g) __attribute__((always_inline)) { if (++g == gpr) { g = 0; ++i; } };

    // bias and residual of this wave's rows, row i in lane i (raw bf16 bits; used by the epilogue behind the stream):
    // requested first, so nothing waits for them later
    uint32_t e_bias = 0u, e_res[B];
#pragma unroll
    for (int b = 0; b < B; ++b) e_res[b] = 0u;
    if (lane < my_rows) {
        const int n = gw + lane * TW;
        const bf16_t *bias = n < n0 ? P.seg[0].bias : n < n1 ? P.seg[1].bias : P.seg[2].bias;
        const int nb = n < n0 ? n : n < n1 ? n - n0 : n - n1;
        if (bias) e_bias = bias[nb];
        if (P.res) {
#pragma unroll
            for (int b = 0; b < B; ++b) e_res[b] = P.res[(int64_t)b * P.res_rs + n];
        }
    }

    // ---- token rows -> LDS (all threads of the workgroup) ---------------------------------------------------------
    // Order of the requests: token-row chunks (L2 hits), then the wave's first TWO weight groups (for a 4096-feature row
    // that is the whole row), then wait for the token rows only — the staging arithmetic, the RMSNorm reduction and the
    // barrier run under the weight loads' latency.  Loads return in order, so this needs the token-row loads outside the
    // compiler's wait bookkeeping as well; they cover the fast path (at most two chunks per thread and row), the
    // generic path below re-reads what it needs behind the weight loads.
    {
        const int nchunk = K >> 3, npad = P.kpad >> 3;
        constexpr int XI = 2;
        constexpr int XPER = PRO == PRO_SILU ? 2 : 1;
        const bool fast = B * XPER <= 4 && npad <= XI * nthr;
        // one staged chunk: raw x chunk (+ partner / norm weight) -> the packed bf16 the product reads
        auto xform = [&](u32x4 v, u32x4 v2, u32x4 gw8, float rs_b) __attribute__((always_inline)) -> u32x4 {
            if constexpr (PRO == PRO_NORM) {
                float f[8], g[8];
                unpack8f(v, f);
                unpack8f(gw8, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] * rs_b * g[e];
                return pack8f(f);
            } else if constexpr (PRO == PRO_SILU) {
                float f[8], u[8];
                unpack8f(v, f);
                unpack8f(v2, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = f[e] / (1.f + expf(-f[e]));                     // silu in fp32 ...
                    const float sr = __uint_as_float(f32_to_bf16_bits(sv) << 16);    // ... a bf16 tensor in the module
                    f[e] = sr * u[e];
                }
                return pack8f(f);
            } else {
                return v;
            }
        };
        auto sumsq = [&](u32x4 v, float acc_) __attribute__((always_inline)) -> float {
            float f[8];
            unpack8f(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc_ = fmaf(f[e], f[e], acc_);
            return acc_;
        };
        auto reduce_rs = [&](float (&ss)[B], float (&rs)[B]) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) ss[b] += __shfl_xor(ss[b], off);
                if (lane == 0) red[b][wave] = ss[b];
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < B; ++b) {
                float tot = 0.f;
                if (nw == 4) tot = red[b][0] + red[b][1] + red[b][2] + red[b][3];     // (the order of duo_rmsnorm_kernel)
                else for (int w2_ = 0; w2_ < nw; ++w2_) tot += red[b][w2_];
                rs[b] = rsqrtf(tot / (float)K + P.eps);
            }
        };
        float rs[B];
#pragma unroll
        for (int b = 0; b < B; ++b) rs[b] = 1.f;
        // One straight-line sequence for every wave, whatever its path afterwards: token-row requests, weight requests,
        // wait for the former.  The loads are asm (outside the compiler's wait bookkeeping), so nothing may touch their
        // destination registers before the matching wait — in particular no copies at a control-flow merge, hence no
        // control flow here (tests/test_token_linear_isa.py audits the built code for exactly that).
        u32x4 xr[B][XI], x2r[B][XI], nwr[XI];
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const uint32_t voff = (uint32_t)min(tid + it * nthr, nchunk - 1) * 16u;
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const uint64_t xb = (uint64_t)(P.x + (int64_t)b * P.x_rs);
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xr[b][it]) : "v"(voff), "s"(xb));
                if constexpr (PRO == PRO_SILU && B * XPER <= 4) {
                    const uint64_t x2b = (uint64_t)(P.x2 + (int64_t)b * P.x_rs);
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(x2r[b][it]) : "v"(voff), "s"(x2b));
                }
            }
            if constexpr (PRO == PRO_NORM) {
                const uint64_t nb_ = (uint64_t)P.norm_w;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(nwr[it]) : "v"(voff), "s"(nb_));
            }
        }
        issue(ii, ig, bufA); adv(ii, ig);
        issue(ii, ig, bufB); adv(ii, ig);
#pragma unroll
        for (int it = 0; it < XI; ++it) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xr[b][it]) : "n"(2 * kLinG));
                if constexpr (PRO == PRO_SILU && B * XPER <= 4) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x2r[b][it]) : "n"(2 * kLinG));
            }
            if constexpr (PRO == PRO_NORM) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(nwr[it]) : "n"(2 * kLinG));
        }
        if (fast) {
            if constexpr (PRO == PRO_NORM) {
                float ss[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    ss[b] = 0.f;
#pragma unroll
                    for (int it = 0; it < XI; ++it)
                        if (tid + it * nthr < nchunk) ss[b] = sumsq(xr[b][it], ss[b]);
                }
                reduce_rs(ss, rs);
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                u32x4 *dst = reinterpret_cast<u32x4 *>(xs + b * xrow);
#pragma unroll
                for (int it = 0; it < XI; ++it) {
                    const int c = tid + it * nthr;
                    u32x4 v2 = {0u, 0u, 0u, 0u}, g8 = {0u, 0u, 0u, 0u};
                    if constexpr (PRO == PRO_SILU && B * XPER <= 4) v2 = x2r[b][it];
                    if constexpr (PRO == PRO_NORM) g8 = nwr[it];
                    if (c < npad) dst[c] = c < nchunk ? xform(xr[b][it], v2, g8, rs[b]) : u32x4{0u, 0u, 0u, 0u};
                }
            }
        } else {
            if constexpr (PRO == PRO_NORM) {
                float ss[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    ss[b] = 0.f;
                    const bf16_t *xr = P.x + (int64_t)b * P.x_rs;
                    for (int c = tid; c < nchunk; c += nthr) ss[b] = sumsq(*reinterpret_cast<const u32x4 *>(xr + c * 8), ss[b]);
                }
                reduce_rs(ss, rs);
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const bf16_t *xr = P.x + (int64_t)b * P.x_rs;
                u32x4 *dst = reinterpret_cast<u32x4 *>(xs + b * xrow);
                for (int c = tid; c < npad; c += nthr) {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (c < nchunk) {
                        u32x4 v2 = v, g8 = v;
                        if constexpr (PRO == PRO_SILU) v2 = *reinterpret_cast<const u32x4 *>(P.x2 + (int64_t)b * P.x_rs + c * 8);
                        if constexpr (PRO == PRO_NORM) g8 = *reinterpret_cast<const u32x4 *>(P.norm_w + c * 8);
                        v = xform(*reinterpret_cast<const u32x4 *>(xr + c * 8), v2, g8, rs[b]);
                    }
                    dst[c] = v;
                }
            }
        }
    }
    __syncthreads();
    if (my_rows <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- the stream ------------------------------------------------------------------------------------------------
    float acc[B];
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b] = 0.f;

    // Row totals are parked one per lane (row i of this wave in lane i) and the epilogue — bias, rounding, residual add,
    // store — runs once behind the stream: its loads would otherwise sit in the same in-order queue as the weight
    // prefetch, and waiting for them would drain it once per row.
    float keep[B];
#pragma unroll
    for (int b = 0; b < B; ++b) keep[b] = 0.f;
    auto flush = [&]() {
        asm volatile("" : "+v"(e_bias));          // (keeps the conversions — and with them the wait for these loads — here)
#pragma unroll
        for (int b = 0; b < B; ++b) asm volatile("" : "+v"(e_res[b]));
        if (lane < my_rows) {
            const int n = gw + lane * TW;
            const float bv = __uint_as_float(e_bias << 16);
#pragma unroll
            for (int b = 0; b < B; ++b) {
                uint32_t r = f32_to_bf16_bits(keep[b] + bv);
                if (P.res) r = f32_to_bf16_bits(__uint_as_float(e_res[b] << 16) + __uint_as_float(r << 16));
                P.y[(int64_t)b * P.y_rs + n] = (bf16_t)r;
            }
        }
    };
    auto consume = [&](int i, int g, u32x4 (&buf)[kLinG], auto may_end_row) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < kLinG; ++j) {
            const int kk = (g * kLinG + j) * 256 + lane * 4;
            float wf[8];
            unpack8f(buf[j], wf);
#pragma unroll
            for (int b = 0; b < B; ++b) {
                float xf[8];
                unpack8f(*reinterpret_cast<const u32x4 *>(xs + b * xrow + kk), xf);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[b] = fmaf(wf[e], xf[e], acc[b]);
            }
        }
        if (decltype(may_end_row)::value && g == gpr - 1) {       // the row is complete (wave-uniform)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const float t = wave_sum(acc[b]);
                acc[b] = 0.f;
                keep[b] = lane == i ? t : keep[b];
            }
        }
    };

    // (gpr is even — the launcher pads a row to whole PAIRS of groups — so a row always ends in the B half and the loop
    //  body has no exit in the middle: one shape for the compiler, and for the audit in tests/test_token_linear_isa.py)
    int ci = 0, cg = 0;
    for (int t = 0; t < T; t += 2) {
        lin_wait<kLinG>(bufA);
        consume(ci, cg, bufA, std::false_type{}); adv(ci, cg);
        issue(ii, ig, bufA); adv(ii, ig);
        lin_wait<kLinG>(bufB);
        consume(ci, cg, bufB, std::true_type{}); adv(ci, cg);
        issue(ii, ig, bufB); adv(ii, ig);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the group re-read behind the wave's last one)
    flush();          // (the launcher keeps a wave's rows <= 64)
}

// Workgroups and threads per workgroup.  Rows are dealt round-robin to waves, so the launch ends with the wave that has
// one row more than the others: pick the wave count (16, 14, 12, 10 or 8 per CU) that deals the rows most evenly.  With
// 256-thread workgroups every such count is a whole number of workgroups per CU; every workgroup stages the token rows
// itself, so long rows (down_proj: 14336 features, twice that read for the SiLU prologue) use 1024-thread workgroups —
// a quarter of the staging work and L2 traffic — when 16 waves per CU deal the rows evenly enough.
static void token_linear_grid(int n_total, int n_in, int n_rows, size_t lds_bytes, int &blocks, int &threads) {
    auto eff_of = [&](int tw) { const int r = (n_total + tw - 1) / tw; return (double)n_total / ((double)r * tw); };
    static const int cand[5] = {16, 14, 12, 10, 8};               // waves per CU (x 256 CUs)
    double best = -1.0;
    int best_c = 16;
    for (int c : cand) {
        const double e = eff_of(256 * c);
        if (e > best + 1e-9) { best = e; best_c = c; }
    }
    const bool long_rows = (int64_t)n_in * n_rows > 8192;
    if (lds_bytes > 38 * 1024 || (long_rows && eff_of(4096) >= 0.95 * best)) {
        threads = lds_bytes > 78 * 1024 || long_rows ? 1024 : 512;
        blocks = 256 * 16 * 64 / threads;                          // 16 waves per CU
    } else {
        threads = 256;
        blocks = 256 * best_c / 4;
    }
    const int wpb = threads / 64;
    if ((int64_t)blocks * wpb > n_total) blocks = (n_total + wpb - 1) / wpb;     // fewer rows than waves
    while ((int64_t)blocks * wpb * 64 < n_total) blocks *= 2;                    // a wave parks at most 64 row totals
}

template <int B>
static void token_linear_launch(const TokenLinearParams &P, int pro, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    switch (pro) {
    case PRO_NORM: hipLaunchKernelGGL((duo_token_linear_kernel<B, PRO_NORM>), grid, block, lds, s, P); break;
    case PRO_SILU: hipLaunchKernelGGL((duo_token_linear_kernel<B, PRO_SILU>), grid, block, lds, s, P); break;
    default: hipLaunchKernelGGL((duo_token_linear_kernel<B, PRO_NONE>), grid, block, lds, s, P); break;
    }
}

}  // namespace

extern "C" int duo_token_linear_bf16(const duo_token_linear_args *a, void *stream) {
    if (!a || !a->x || !a->y) return DUO_EINVAL;
    if (a->n_rows < 0 || a->n_rows > kLinMaxRows || a->n_in < 8 || (a->n_in & 7)) return DUO_EINVAL;
    if (a->n_rows == 0) return 0;
    if (a->norm_weight && a->x2) return DUO_EINVAL;                 // one prologue at a time
    if ((a->x_row_stride & 7) || ((uintptr_t)a->x & 15) || ((uintptr_t)a->x2 & 15) || ((uintptr_t)a->norm_weight & 15))
        return DUO_EINVAL;
    TokenLinearParams P;
    P.x = (const bf16_t *)a->x; P.x2 = (const bf16_t *)a->x2; P.x_rs = a->x_row_stride;
    P.K = a->n_in;
    P.gpr = 2 * ((a->n_in + 2 * kLinG * 512 - 1) / (2 * kLinG * 512));      // whole pairs of groups (see the stream loop)
    P.kpad = P.gpr * kLinG * 512;
    int64_t n_total = 0;
    for (int s = 0; s < 3; ++s) {
        const duo_linear_seg &g = a->seg[s];
        if (g.n < 0 || (g.n > 0 && (!g.w || (g.row_stride & 7) || ((uintptr_t)g.w & 15) || g.row_stride < a->n_in)))
            return DUO_EINVAL;
        if (g.n == 0 && s + 1 < 3 && a->seg[s + 1].n > 0) return DUO_EINVAL;     // blocks are packed from seg[0]
        P.seg[s].w = (const bf16_t *)g.w; P.seg[s].bias = (const bf16_t *)g.bias; P.seg[s].rs = g.row_stride; P.seg[s].n = g.n;
        n_total += g.n;
    }
    if (n_total <= 0) return 0;
    if (n_total > (1 << 30)) return DUO_EINVAL;
    P.n_total = (int32_t)n_total;
    P.norm_w = (const bf16_t *)a->norm_weight; P.eps = a->norm_eps;
    P.res = (const bf16_t *)a->residual; P.res_rs = a->residual_row_stride;
    P.y = (bf16_t *)a->y; P.y_rs = a->y_row_stride;
    const size_t lds = (size_t)a->n_rows * P.kpad * 2;
    if (lds > 156 * 1024) return DUO_EINVAL;                        // n_rows * n_in beyond one CU's LDS
    int blocks, threads;
    token_linear_grid(P.n_total, a->n_in, a->n_rows, lds, blocks, threads);
    const int pro = a->norm_weight ? PRO_NORM : a->x2 ? PRO_SILU : PRO_NONE;
    hipStream_t s = (hipStream_t)stream;
    if (lds > 64 * 1024) {
        // (above the default dynamic-LDS limit the attribute has to be raised once per instantiation; cheap, idempotent)
#define DUO_LIN_ATTR(Bv, PROv) (void)hipFuncSetAttribute((const void *)duo_token_linear_kernel<Bv, PROv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        switch (a->n_rows) {
        case 1: DUO_LIN_ATTR(1, PRO_NONE); DUO_LIN_ATTR(1, PRO_NORM); DUO_LIN_ATTR(1, PRO_SILU); break;
        case 2: DUO_LIN_ATTR(2, PRO_NONE); DUO_LIN_ATTR(2, PRO_NORM); DUO_LIN_ATTR(2, PRO_SILU); break;
        case 3: DUO_LIN_ATTR(3, PRO_NONE); DUO_LIN_ATTR(3, PRO_NORM); DUO_LIN_ATTR(3, PRO_SILU); break;
        default: DUO_LIN_ATTR(4, PRO_NONE); DUO_LIN_ATTR(4, PRO_NORM); DUO_LIN_ATTR(4, PRO_SILU); break;
        }
#undef DUO_LIN_ATTR
    }
    switch (a->n_rows) {
    case 1: token_linear_launch<1>(P, pro, dim3(blocks), dim3(threads), lds, s); break;
    case 2: token_linear_launch<2>(P, pro, dim3(blocks), dim3(threads), lds, s); break;
    case 3: token_linear_launch<3>(P, pro, dim3(blocks), dim3(threads), lds, s); break;
    default: token_linear_launch<4>(P, pro, dim3(blocks), dim3(threads), lds, s); break;
    }
    return (int)hipGetLastError();
}
