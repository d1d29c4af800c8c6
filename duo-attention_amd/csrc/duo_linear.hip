// Token-row linear layers of the decode step (gfx950).
//
// At q_len == 1 the projections either side of the attention op — q/k/v_proj and o_proj inside the reference's static
// forward (duo_attn/patch/llama.py:332-340, :430-432) and the MLP + norms of its decoder layer
// (duo_attn/patch/static_kv_cache.py:482-537) — are matrix-VECTOR products: every weight byte is read once per token,
// 33-117 MB per call, nothing to reuse.  They are HBM-bound byte streaming, not GEMM work, and the library GEMM kernels
// that serve them at M = 1 run at 0.6-4.9 TB/s (profiles/r3_model_level_kernels.md).  This file streams the weight rows
// once (four streaming waves per CU, two groups of loads in flight each), keeps the token rows in LDS, and folds the
// element-wise neighbours of each product into it:
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
// round-robin to the streaming waves of the grid; a wave keeps two groups of four loads in flight across row boundaries
// (the next group is issued unconditionally — behind the wave's last group it re-reads that group — so the waits are
// counted).  FOUR streaming waves per CU (one workgroup per CU) when that deals the rows evenly: measured faster than 8,
// 12 or 16 on every shape (profiles/r3_token_linear.md); token_linear_grid below picks the count.
#include <stdlib.h>
#include "duo_common.h"
#include "duo_kv_ops.h"

namespace {

#ifndef DUO_LIN_G
#define DUO_LIN_G 4
#endif
constexpr int kLinG = DUO_LIN_G;    // weight loads per group (1 KiB each per wave); two groups in flight per wave
static_assert(kLinG == 4 || kLinG == 8, "issue() emits the loads four at a time");
static_assert(DUO_LIN_G != 4 || kLinG * 512 == DUO_TOKEN_LINEAR_PAD, "the header's LDS padding rule is the default build's");
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
    int32_t K, kpad, gpr;           // gpr: groups of kLinG x 512 elements per weight row (kpad = gpr * kLinG * 512)
    LinSegDev seg[3];
    int32_t n_total;
    int32_t sw;                     // waves per workgroup that take rows (<= blockDim / 64)
    const bf16_t *norm_w;
    float eps;
    int32_t norm_hf;                // DUO_LINEAR_NORM_HF: normalised x rounded to bf16 before the weight multiply (HF *RMSNorm)
    const bf16_t *res;
    int64_t res_rs;
    bf16_t *y;
    int64_t y_rs;
};

enum { PRO_NONE = 0, PRO_NORM = 1, PRO_SILU = 2 };

// the group's four loads have landed when at most N newer vector-memory operations are outstanding (loads return in order)
template <int N>
__device__ __forceinline__ void lin_wait(u32x4 (&buf)[kLinG]) {
#pragma unroll
    for (int j = 0; j < kLinG; j += 4)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(buf[j]), "+v"(buf[j + 1]), "+v"(buf[j + 2]), "+v"(buf[j + 3]) : "n"(N));
}

__device__ __forceinline__ float wave_sum(float x) {
    x = row16_allreduce_sum(x);                 // every lane of a 16-lane row holds the row's total
    const int xi = __float_as_int(x);
    return (__int_as_float(__builtin_amdgcn_readlane(xi, 0)) + __int_as_float(__builtin_amdgcn_readlane(xi, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(xi, 32)) + __int_as_float(__builtin_amdgcn_readlane(xi, 48)));
}

// the token-row chunk in `reg` has landed: the two weight groups issued behind it may still fly — or nothing at all in a
// wave without rows, which skipped them (has_rows == 0)
#define LIN_WAIT_X(reg)                                                                                                  \
    asm volatile("s_waitcnt vmcnt(%1)\n\t"                                                                               \
                 "s_cmp_eq_u32 %2, 0\n\t"                                                                                 \
                 "s_cbranch_scc0 2f\n\t"                                                                                  \
                 "s_waitcnt vmcnt(0)\n"                                                                                   \
                 "2:"                                                                                                     \
                 : "+v"(reg) : "n"(2 * kLinG), "s"(has_rows) : "scc")

template <int B, int PRO>
__global__ __launch_bounds__(1024) void duo_token_linear_kernel(const TokenLinearParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t xs[];      // [B][kpad / 2] packed bf16 pairs
    __shared__ float red[kLinMaxRows][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // rows go to the first P.sw waves of a workgroup only: the stream is fastest with FEW waves per CU (measured: 4 per CU
    // beat 8, 12 and 16 on every shape), while staging long token rows wants many threads — the other waves stage, meet the
    // barrier and leave
    const int sw = P.sw;
    const int TW = gridDim.x * sw, gw = blockIdx.x * sw + wave;
    const int my_rows = __builtin_amdgcn_readfirstlane(wave < sw && gw < P.n_total ? (P.n_total - gw + TW - 1) / TW : 0);
    uint32_t has_rows;           // (wave-uniform; the asm puts it in an SGPR whatever unit the compiler computed my_rows on)
    {
        const uint32_t hv = my_rows > 0 ? 1u : 0u;
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(has_rows) : "v"(hv));
    }
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
        uint32_t off[kLinG];
#pragma unroll
        for (int j = 0; j < kLinG; ++j) off[j] = (uint32_t)min(((g * kLinG + j) * 64 + lane) * 8, K - 8) * 2u;
        // (asm: the stream's loads are outside the compiler's wait bookkeeping on purpose — its loop-carried analysis
        //  put a vmcnt(0) at the loop head, i.e. it let the prefetch drain every second group; the waits are the
        //  explicit counted ones in lin_wait below.  A wave without rows branches over them INSIDE the statement:
        //  to the compiler it is the same straight-line code for every wave.)
#pragma unroll
        for (int j = 0; j < kLinG; j += 4)
            asm volatile("s_cmp_eq_u32 %[hr], 0\n\t"
                         "s_cbranch_scc1 1f\n\t"
                         "global_load_dwordx4 %[d0], %[o0], %[b] nt\n\t"
                         "global_load_dwordx4 %[d1], %[o1], %[b] nt\n\t"
                         "global_load_dwordx4 %[d2], %[o2], %[b] nt\n\t"
                         "global_load_dwordx4 %[d3], %[o3], %[b] nt\n"
                         "1:"
                         : [d0] "=&v"(buf[j]), [d1] "=&v"(buf[j + 1]), [d2] "=&v"(buf[j + 2]), [d3] "=&v"(buf[j + 3])
                         : [o0] "v"(off[j]), [o1] "v"(off[j + 1]), [o2] "v"(off[j + 2]), [o3] "v"(off[j + 3]), [b] "s"(wr), [hr] "s"(has_rows)
                         : "scc");
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
                for (int e = 0; e < 8; ++e) {
                    // flashinfer form: one rounding of x * rs * w.  HuggingFace's LlamaRMSNorm / MistralRMSNorm (the tuple
                    // path's norm): `weight * (x * rsqrt(var + eps)).to(bf16)` — two roundings.  A select, not a branch:
                    // weight loads are in flight here (see the note on control flow below)
                    const float t = f[e] * rs_b;
                    const float tr = __uint_as_float(f32_to_bf16_bits(t) << 16);
                    f[e] = (P.norm_hf ? tr : t) * g[e];
                }
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
                LIN_WAIT_X(xr[b][it]);
                if constexpr (PRO == PRO_SILU && B * XPER <= 4) LIN_WAIT_X(x2r[b][it]);
            }
            if constexpr (PRO == PRO_NORM) LIN_WAIT_X(nwr[it]);
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
    auto consume = [&](int i, int g, u32x4 (&buf)[kLinG]) __attribute__((always_inline)) {
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
        if (g == gpr - 1) {                 // the row is complete (wave-uniform)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const float t = wave_sum(acc[b]);
                acc[b] = 0.f;
                keep[b] = lane == i ? t : keep[b];
            }
        }
    };

    // (the loop body has no exit in the middle — with an odd number of groups the last B half is skipped by a forward
    //  branch over its arithmetic only: one shape for the compiler, and for the audit in tests/test_token_linear_isa.py)
    int ci = 0, cg = 0;
    for (int t = 0; t < T; t += 2) {
        lin_wait<kLinG>(bufA);
        consume(ci, cg, bufA); adv(ci, cg);
        issue(ii, ig, bufA); adv(ii, ig);
        lin_wait<kLinG>(bufB);
        if (t + 1 < T) { consume(ci, cg, bufB); adv(ci, cg); }
        issue(ii, ig, bufB); adv(ii, ig);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the group re-read behind the wave's last one)
    flush();          // (the launcher keeps a wave's rows <= 64)
}

// Workgroups, threads per workgroup and streaming waves per workgroup.  Rows are dealt round-robin to the streaming waves,
// so the launch ends with the wave that has one row more than the others: the count must deal the rows evenly.  Among
// the counts that do, FEWER streaming waves per CU are faster — measured on every shape of a Llama-3-8B layer (device-side
// durations, profiles/r3_token_linear.md): o_proj 12.2 / 10.5 / 8.6 us and gate|up 47 / 46 / 41 us at 16 / 8 / 4 waves per
// CU — so 4 per CU (one workgroup per CU) is preferred and a larger count has to deal the rows at least 2 % better.
// Staging is the opposite: every workgroup stages the token rows itself, and long rows (down_proj: 14336 features, twice
// that read for the SiLU prologue) want many threads — those launches use 1024-thread workgroups of which the first four
// waves stream (29.2 -> 24.7 us against four-wave workgroups).
static void token_linear_grid(int n_total, int n_in, int n_rows, size_t lds_bytes, int &blocks, int &threads, int &sw) {
    auto eff_of = [&](int tw) { const int r = (n_total + tw - 1) / tw; return (double)n_total / ((double)r * tw); };
    static const int cand[6] = {4, 8, 16, 12, 14, 10};            // streaming waves per CU (x 256 CUs), in order of preference
    static const int forced = [] { const char *e = getenv("DUO_LINEAR_WAVES_PER_CU"); return e ? atoi(e) : 0; }();   // (sweeps)
    static const int forced_thr = [] { const char *e = getenv("DUO_LINEAR_THREADS"); return e ? atoi(e) : 0; }();
    double best = -1.0;
    int best_c = 4;
    for (int c : cand) {
        const double e = eff_of(256 * c);
        if (e > best + 0.02) { best = e; best_c = c; }           // (a later candidate has to deal the rows 2 % better)
    }
    if (forced > 0 && forced <= 16) best_c = forced;
    // threads per workgroup: 256 stage a 4096-feature row in two chunks each; long rows (down_proj: 14336 features, twice
    // that read for the SiLU prologue) take 1024
    const bool long_rows = (int64_t)n_in * n_rows > 8192 || lds_bytes > 38 * 1024;
    threads = forced_thr == 256 || forced_thr == 512 || forced_thr == 1024 ? forced_thr : long_rows ? 1024 : 256;
    if (lds_bytes > 78 * 1024) threads = 1024;
    const int wpb = threads / 64;
    const int per_cu = lds_bytes > 78 * 1024 ? 1 : lds_bytes > 38 * 1024 ? 2 : 4;       // workgroups that fit a CU's LDS
    // one workgroup per CU carries best_c streaming waves when it can (best_c <= its waves); else several workgroups
    int wgs_per_cu = 1;
    sw = best_c;
    while (sw > wpb && wgs_per_cu < per_cu) { wgs_per_cu *= 2; sw = best_c / wgs_per_cu; }
    if (sw > wpb) sw = wpb;
    if (sw < 1) sw = 1;
    blocks = 256 * wgs_per_cu;
    if ((int64_t)blocks * sw > n_total) blocks = (n_total + sw - 1) / sw;        // fewer rows than waves
    while ((int64_t)blocks * sw * 64 < n_total) blocks *= 2;                     // a wave parks at most 64 row totals
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
    P.gpr = (a->n_in + kLinG * 512 - 1) / (kLinG * 512);
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
    if (a->flags & ~DUO_LINEAR_NORM_HF) return DUO_EINVAL;         // unknown flag bits
    P.norm_hf = (a->flags & DUO_LINEAR_NORM_HF) ? 1 : 0;
    P.res = (const bf16_t *)a->residual; P.res_rs = a->residual_row_stride;
    P.y = (bf16_t *)a->y; P.y_rs = a->y_row_stride;
    const size_t lds = (size_t)a->n_rows * P.kpad * 2;
    if (lds > 156 * 1024) return DUO_EINVAL;                        // n_rows * n_in beyond one CU's LDS
    int blocks, threads;
    token_linear_grid(P.n_total, a->n_in, a->n_rows, lds, blocks, threads, P.sw);
    const int pro = a->norm_weight ? PRO_NORM : a->x2 ? PRO_SILU : PRO_NONE;
    hipStream_t s = (hipStream_t)stream;
    if (lds > 64 * 1024) {
        // above the default dynamic-LDS limit the attribute has to be raised on the instantiation that is launched
        // (idempotent, host-side only; sized to this launch — the static part of the kernel's LDS comes on top)
        const void *fn = nullptr;
#define DUO_LIN_FN(Bv) (pro == PRO_NORM ? (const void *)duo_token_linear_kernel<Bv, PRO_NORM> : pro == PRO_SILU ? (const void *)duo_token_linear_kernel<Bv, PRO_SILU> : (const void *)duo_token_linear_kernel<Bv, PRO_NONE>)
        switch (a->n_rows) {
        case 1: fn = DUO_LIN_FN(1); break;
        case 2: fn = DUO_LIN_FN(2); break;
        case 3: fn = DUO_LIN_FN(3); break;
        default: fn = DUO_LIN_FN(4); break;
        }
#undef DUO_LIN_FN
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    switch (a->n_rows) {
    case 1: token_linear_launch<1>(P, pro, dim3(blocks), dim3(threads), lds, s); break;
    case 2: token_linear_launch<2>(P, pro, dim3(blocks), dim3(threads), lds, s); break;
    case 3: token_linear_launch<3>(P, pro, dim3(blocks), dim3(threads), lds, s); break;
    default: token_linear_launch<4>(P, pro, dim3(blocks), dim3(threads), lds, s); break;
    }
    return (int)hipGetLastError();
}
