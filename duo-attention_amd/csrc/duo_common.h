// Shared device helpers for the gfx950 DuoAttention kernels.
// gfx950 only: wave = 64 lanes, bf16 MFMA 32x32x16, ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/duo_attn_hip.h"

#define DUO_HEAD_DIM 128
#define DUO_WAVE 64

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B operand
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// round-to-nearest-even fp32 -> bf16 bits (NaN quieted), same as torch's cast
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
}

// DPP cross-lane moves inside a 16-lane row (no LDS traffic)
#define DUO_DPP_QUAD_XOR1 0xB1     // quad_perm [1,0,3,2]
#define DUO_DPP_QUAD_XOR2 0x4E     // quad_perm [2,3,0,1]
#define DUO_DPP_ROW_HALF_MIRROR 0x141
#define DUO_DPP_ROW_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row; every lane ends up with the total
__device__ __forceinline__ float row16_allreduce_sum(float x) {
    x += dpp_mov<DUO_DPP_QUAD_XOR1>(x);
    x += dpp_mov<DUO_DPP_QUAD_XOR2>(x);
    x += dpp_mov<DUO_DPP_ROW_HALF_MIRROR>(x);
    x += dpp_mov<DUO_DPP_ROW_MIRROR>(x);
    return x;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct DuoSegDev {
    const bf16_t *k;
    const bf16_t *v;
    int64_t token_stride;
    int64_t head_stride;
    int32_t len;
    int64_t batch_stride;
};
struct DuoClassDev {
    int32_t n_kv_heads;
    int32_t q_head_offset;
    DuoSegDev a;
    DuoSegDev b;
};

// Field-wise select between two class descriptors living in the kernarg segment.  Indexing a by-value
// kernel argument with a runtime index (P.cls[ci]) makes hipcc copy the struct to scratch and reload
// it with VMEM instructions inside the hot loop; scalar selects keep everything in SGPRs.
__device__ __forceinline__ DuoSegDev duo_select(const DuoSegDev &a, const DuoSegDev &b, bool pick_b) {
    DuoSegDev r;
    r.k = pick_b ? b.k : a.k;
    r.v = pick_b ? b.v : a.v;
    r.token_stride = pick_b ? b.token_stride : a.token_stride;
    r.head_stride = pick_b ? b.head_stride : a.head_stride;
    r.len = pick_b ? b.len : a.len;
    r.batch_stride = pick_b ? b.batch_stride : a.batch_stride;
    return r;
}
__device__ __forceinline__ DuoClassDev duo_select(const DuoClassDev &a, const DuoClassDev &b, bool pick_b) {
    DuoClassDev r;
    r.n_kv_heads = pick_b ? b.n_kv_heads : a.n_kv_heads;
    r.q_head_offset = pick_b ? b.q_head_offset : a.q_head_offset;
    r.a = duo_select(a.a, b.a, pick_b);
    r.b = duo_select(a.b, b.b, pick_b);
    return r;
}

// batched launches: the batch row is a grid dimension; a row's view of a class = the descriptor moved by its strides
__device__ __forceinline__ void duo_class_batch_row(DuoClassDev &C, int row) {
    C.a.k += (int64_t)row * C.a.batch_stride;
    C.a.v += (int64_t)row * C.a.batch_stride;
    C.b.k += (int64_t)row * C.b.batch_stride;
    C.b.v += (int64_t)row * C.b.batch_stride;
}

static inline DuoSegDev duo_seg_dev(const duo_kv_seg &s) {
    DuoSegDev d;
    d.k = (const bf16_t *)s.k;
    d.v = (const bf16_t *)s.v;
    d.token_stride = s.token_stride;
    d.head_stride = s.head_stride;
    d.len = s.len;
    d.batch_stride = s.batch_stride;
    return d;
}
static inline DuoClassDev duo_class_dev(const duo_head_class *c) {
    DuoClassDev d;
    if (c == nullptr) {
        d.n_kv_heads = 0;
        d.q_head_offset = 0;
        d.a = DuoSegDev{nullptr, nullptr, 0, 0, 0, 0};
        d.b = DuoSegDev{nullptr, nullptr, 0, 0, 0, 0};
        return d;
    }
    d.n_kv_heads = c->n_kv_heads;
    d.q_head_offset = c->q_head_offset;
    d.a = duo_seg_dev(c->segA);
    d.b = duo_seg_dev(c->segB);
    return d;
}

#define DUO_HIP_CHECK_LAUNCH()                      \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)
