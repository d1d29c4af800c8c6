// Shared pieces of the two prefill kernels (8 waves x 32 rows, 4 waves x 64 rows): tile geometry,
// parameter block, LDS layouts, LDS-DMA staging helpers.  Internal linkage, included by the .hip files.
#pragma once
#include <type_traits>
#include "duo_common.h"

namespace {

constexpr int QBLK = 256;   // query rows per workgroup
constexpr int KVBLK = 64;   // keys per tile
constexpr int K_TILE_BYTES = KVBLK * DUO_HEAD_DIM * 2;  // 16 KiB
constexpr int V_TILE_BYTES = K_TILE_BYTES;
constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;          // 96 KiB
// each wave issues 4 global_load_lds per tile (2 K pieces + 2 V pieces): the vmcnt(4) below
constexpr float kDeferLog2 = 8.0f;   // deferred-rescale threshold in the exp2 domain

struct PrefillParams {
    const bf16_t *q;
    int64_t q_ts, q_hs;
    bf16_t *out;
    int64_t o_ts, o_hs;
    int32_t S;
    int32_t group;
    int32_t n_qtiles;
    int32_t nblk_full;     // workgroups of class 0 (its key-range pieces and the padding of its last XCD period included)
    DuoClassDev cls[2];
    float scale_log2e;
    uint32_t flags;
    // key-range splits, per head class: each (q tile, q head) of class c is covered by ks[c] workgroups that walk
    // disjoint ranges of its K/V tiles and leave un-normalised partials (O, m, l) in the workspace for
    // duo_prefill_merge_kernel.  1 = no split (the workgroup writes `out` itself).  Chosen per launch by the planner in
    // duo_prefill.hip (a list-scheduling replay of the launch on the chip's 256 CUs).
    int32_t ks[2];
    int32_t pbase[2];      // first partial of class c (partials of one batch row; unused when ks[c] == 1)
    int32_t nparts;        // partials per batch row
    float *ws_o;     // [partial][256 rows][128] fp32
    float *ws_ml;    // [partial][256 rows][2]   (row max in score units, row sum)
    // XCD-aware block order of class c (4-wave kernel; see prefill_map_block): q-tile rows per period and workgroups per
    // XCD per period; xmap_q[c] == 0: plain q-tile-major order
    int32_t xmap_rows[2], xmap_q[2];
    // batched launch: grid.y = batch row; q / out rows of a batch row are q_bs / o_bs elements apart, the segments carry
    // their own batch strides, every row has nparts partials of its own in the workspace
    int64_t q_bs, o_bs;
};

// block -> (class, q tile, kv head, q head of the group, key-range piece).  One place for both prefill kernels and the
// merge kernel's inverse (prefill_item_of).
//
// Plain order (xmap_q == 0): pieces of one (q tile, q head) are adjacent block ids, q heads next (kv head fastest: group
// mates sit n_kv_heads items apart), q tiles heaviest (latest) first.
// XCD-aware order (xmap_q > 0): workgroup b runs on XCD b % 8 (round-robin dispatch; class 0 starts at block 0 and is
// padded to a multiple of 8 blocks, so the same holds for class 1), and each XCD has its own L2, so one K/V stream — a kv
// head, or with key-range splits ONE PIECE of a kv head: a "virtual head" vh = kvh * ks + piece — should be read by as
// few XCDs as possible, for every head count, not only the divisors of 8.  A period = xmap_rows q-tile rows = 8 * xmap_q
// workgroups (the smallest whole number of rows that deals every XCD the same count); inside a period the workgroups are
// laid out VIRTUAL-HEAD-major and XCD x takes the contiguous positions [x * xmap_q, (x + 1) * xmap_q): one or two streams
// per XCD, every XCD the same work per period, periods heaviest (latest q tiles) first.  Workgroups of a padded last
// period (n_qtiles not a multiple of xmap_rows) have no tile: `tile` < 0, and they leave.
struct PrefillItem {
    int ci, tile, kvh, g, split, ks;
    int part;       // partial slot of this workgroup inside one batch row's partials (valid when ks > 1)
    int item;       // canonical index of the (q tile, q head) inside its class: (rank * n_kv_heads + kvh) * group + g
};
__host__ __device__ __forceinline__ PrefillItem prefill_map_block(const PrefillParams &P, int b) {
    PrefillItem I;
    I.ci = b < P.nblk_full ? 0 : 1;
    if (I.ci) b -= P.nblk_full;
    const int nkv = I.ci ? P.cls[1].n_kv_heads : P.cls[0].n_kv_heads;
    const int xq = I.ci ? P.xmap_q[1] : P.xmap_q[0], xrows = I.ci ? P.xmap_rows[1] : P.xmap_rows[0];
    I.ks = I.ci ? P.ks[1] : P.ks[0];
    const int nq_c = nkv * P.group;
    int rank;
    if (xq > 0) {
        const int x = b & 7, r = b >> 3;
        const int per = r / xq, j = r - per * xq;
        const int w = x * xq + j;
        const int per_head = xrows * P.group;
        const int vh = w / per_head;
        const int e = w - vh * per_head;
        const int row = e / P.group;
        I.g = e - row * P.group;
        I.kvh = vh / I.ks;
        I.split = vh - I.kvh * I.ks;
        rank = per * xrows + row;
    } else {
        I.split = b % I.ks;
        b /= I.ks;
        rank = b / nq_c;
        const int p = b - rank * nq_c;
        I.kvh = p % nkv;                   // group mates sit n_kv_heads items apart
        I.g = p / nkv;
    }
    I.tile = rank < P.n_qtiles ? P.n_qtiles - 1 - rank : -1;
    I.item = (rank * nkv + I.kvh) * P.group + I.g;
    I.part = (I.ci ? P.pbase[1] : P.pbase[0]) + I.item * I.ks + I.split;
    return I;
}

typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    f32x2 v = {lo, hi};
    hw_bf16x2 r = __builtin_convertvector(v, hw_bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
    return *reinterpret_cast<uint32_t *>(&r);
}

// 16-bit element type of the kernel: bf16 (the static dual-cache path) or fp16 (dequantised INT4 pools,
// fp16 models).  Same data movement and layouts; only the MFMA opcode and the float -> 16-bit pack differ.
typedef _Float16 f16x8_mfma __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_cvt __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ uint32_t cvt_pk16(float lo, float hi) {
    if constexpr (F16) {
        f32x2 v = {lo, hi};
        f16x2_cvt r = __builtin_convertvector(v, f16x2_cvt);   // v_cvt_pk_f16_f32 (RNE)
        return *reinterpret_cast<uint32_t *>(&r);
    } else {
        return cvt_pk_bf16(lo, hi);
    }
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma32x32x16(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8_mfma *>(&a),
                                                      *reinterpret_cast<const f16x8_mfma *>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// One output row slice of a split item: combine its ks partials — out = sum_s 2^((m_s - M) c) O_s / sum_s 2^((m_s - M) c) l_s.
// row0: index of the row in piece 0's partial (piece s: + s * QBLK rows); jd: which group of 16 dims.
// Every piece's (m, l) is requested up front, the accumulator rows follow four pieces at a time.
template <bool F16>
__device__ __forceinline__ void prefill_merge_row(const PrefillParams &P, int ks, int64_t row0, bf16_t *op, int jd) {
    constexpr int KMAX = 16;
    f32x2 ml[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        const float *mp = P.ws_ml + (row0 + (int64_t)(s < ks ? s : 0) * QBLK) * 2;
        ml[s] = *reinterpret_cast<const f32x2 *>(mp);
    }
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) M = fmaxf(M, s < ks ? ml[s].x : -INFINITY);
    float L = 0.f;
    float w[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        // a piece that saw no key of the row (m = -inf, l = 0) has weight 0; a row no piece saw cannot exist (the row's own
        // key is visible to it)
        w[s] = (s >= ks || ml[s].x == -INFINITY) ? 0.f : fast_exp2((ml[s].x - M) * P.scale_log2e);
        L = fmaf(ml[s].y, w[s], L);
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *src0 = P.ws_o + row0 * DUO_HEAD_DIM + 16 * jd;
#pragma unroll
    for (int s0 = 0; s0 < KMAX; s0 += 4) {
        if (s0 >= ks) break;
        f32x4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u < ks ? s0 + u : s0;       // past the end: re-read a valid piece with weight 0
            const float *src = src0 + (int64_t)s * QBLK * DUO_HEAD_DIM;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[u][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src) + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float wu = s0 + u < ks ? w[s0 + u] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = acc[i] + v[u][i] * wu;
        }
    }
    const float inv = 1.f / L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x2 w2;
        w2.x = cvt_pk16<F16>(acc[i].x * inv, acc[i].y * inv);
        w2.y = cvt_pk16<F16>(acc[i].z * inv, acc[i].w * inv);
        *reinterpret_cast<u32x2 *>(op + 16 * jd + 4 * i) = w2;
    }
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

struct TileSrc {
    const bf16_t *k;
    const bf16_t *v;
    int64_t ts;
    int32_t row0;
    int32_t cnt;   // valid rows in the tile (1..64)
};

__device__ __forceinline__ TileSrc tile_src(const DuoClassDev &C, int kvh, int t, int nA, int S) {
    TileSrc s;
    if (t < nA) {
        s.k = C.a.k + (int64_t)kvh * C.a.head_stride;
        s.v = C.a.v + (int64_t)kvh * C.a.head_stride;
        s.ts = C.a.token_stride;
        s.row0 = t * KVBLK;
        s.cnt = min(KVBLK, C.a.len - s.row0);
    } else {
        s.k = C.b.k + (int64_t)kvh * C.b.head_stride;
        s.v = C.b.v + (int64_t)kvh * C.b.head_stride;
        s.ts = C.b.token_stride;
        s.row0 = (t - nA) * KVBLK;
        s.cnt = min(KVBLK, S - s.row0);
    }
    return s;
}

__device__ __forceinline__ int k_lds_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }

// LDS-DMA staging: HBM/L2 -> LDS without the VGPR round trip and without ds_write.  The LDS
// destination of a wave-instruction is wave-uniform base + lane*16 (one contiguous KiB), so the K
// swizzle and the V block layout are applied on the per-lane SOURCE address:
//   K piece w = rows 4w..4w+3, LDS chunk p of row r holds source chunk p ^ (r & 15);
//   V piece w = key quad w = 8 blocks [4 keys][16 dims]  (byte (key,d): ((key/4)*8 + d/16)*128
//               + (key%4)*32 + (d%16)*2).
// 512 threads: 16 pieces of K and 16 of V per tile, 2 of each per wave (DMA_PER_TILE = 4).
//
// Issued through inline asm: hipcc orders two LDS-DMA writes whose destinations it cannot tell apart
// (runtime ring slot) with an s_waitcnt vmcnt(0) in front of the second one, and drains the DMA
// before LDS reads it cannot disambiguate — either would collapse the ring to depth one.  asm VMEM
// operations are invisible to its waitcnt pass (they can only make its own waits more conservative),
// so the kernel counts them itself: DMA_PER_TILE per wave per tile, waited with s_waitcnt vmcnt(N).
// M0 (the LDS destination base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16(const bf16_t *gsrc, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst_uniform)
        : "memory");
}
// same, address = wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset: the
// per-tile address arithmetic is scalar, the lane offsets are loop invariants -> no VALU per tile
__device__ __forceinline__ void glds16_s(const bf16_t *sbase_uniform, uint32_t voff_bytes, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff_bytes), "s"(sbase_uniform), "s"(lds_dst_uniform)
        : "memory");
}

// per-lane source byte offsets of piece j=0 inside a tile of a segment with token stride ts
struct DmaLane {
    uint32_t kofs, vofs;
};
__device__ __forceinline__ DmaLane dma_lane(int tid, int64_t ts) {
    const int lane = tid & 63;
    const int w0 = tid >> 6;
    const int krow = 4 * w0 + (lane >> 4);
    const int kch = (lane & 15) ^ (krow & 15);
    const int vrow = 4 * w0 + ((lane & 7) >> 1);
    const int vch = ((lane >> 3) << 1) + (lane & 1);
    DmaLane d;
    d.kofs = (uint32_t)(krow * ts + kch * 8) * 2u;
    d.vofs = (uint32_t)(vrow * ts + vch * 8) * 2u;
    return d;
}

// general form (tail tiles: rows past the segment end are clamped to its last row).
// NW = waves per workgroup: wave w stages pieces w, w+NW, ... of the 16 K and 16 V pieces.
template <int NW>
__device__ __forceinline__ void stage_dma_tail(const TileSrc &s, uint32_t stage_lds, int tid) {
    const int lane = tid & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t base = __builtin_amdgcn_readfirstlane(stage_lds);
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) {
        const int piece = w0 + NW * j;                   // 0..15, wave-uniform
        const int krow = 4 * piece + (lane >> 4);
        const int kch = (lane & 15) ^ (krow & 15);
        const int rk = s.row0 + min(krow, s.cnt - 1);
        glds16(s.k + (int64_t)rk * s.ts + kch * 8, base + piece * 1024);
        const int vrow = 4 * piece + ((lane & 7) >> 1);
        const int vch = ((lane >> 3) << 1) + (lane & 1);
        const int rv = s.row0 + min(vrow, s.cnt - 1);
        glds16(s.v + (int64_t)rv * s.ts + vch * 8, base + K_TILE_BYTES + piece * 1024);
    }
}

// full 64-row tile: scalar tile base + loop-invariant lane offsets (4*NW rows between a wave's pieces;
// (krow + 4*NW*j) & 15 == krow & 15 for NW in {4, 8}, so the swizzled lane offset is the same)
template <int NW>
__device__ __forceinline__ void stage_dma_full(const TileSrc &s, const DmaLane &L, uint32_t stage_lds, int tid) {
    const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t base = __builtin_amdgcn_readfirstlane(stage_lds) + w0 * 1024;
    const bf16_t *kb = s.k + (int64_t)s.row0 * s.ts;     // wave-uniform
    const bf16_t *vb = s.v + (int64_t)s.row0 * s.ts;
    const int64_t step = 4 * NW * s.ts;                  // rows between consecutive pieces of a wave
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) {
        glds16_s(kb + j * step, L.kofs, base + j * NW * 1024);
        glds16_s(vb + j * step, L.vofs, base + K_TILE_BYTES + j * NW * 1024);
    }
}

// ds_read_b64_tr_b16 through inline asm: hipcc treats the builtin form as possibly aliasing the
// LDS-DMA in flight and drains it (s_waitcnt vmcnt(0)) before the first read of every tile, which
// would undo the counted-vmcnt pipeline.  asm loads are invisible to the waitcnt pass, so their
// completion is waited for by hand (lgkmcnt) before the MFMAs that consume them.
#define DUO_TR_READ(dst, addr, off) \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

// the 8 transpose reads (4 output dim blocks x 2 key quads) of PV k-step `step` (= 2*bb + s)
#define DUO_TR_STEP(buf, vaddr, ibase, step)                                          \
    do {                                                                              \
        _Pragma("unroll") for (int db_ = 0; db_ < 4; ++db_) {                         \
            DUO_TR_READ(buf[2 * db_], vaddr, (ibase) + (step) * 4096 + db_ * 256);    \
            DUO_TR_READ(buf[2 * db_ + 1], vaddr, (ibase) + (step) * 4096 + db_ * 256 + 2048); \
        }                                                                             \
    } while (0)

__device__ __forceinline__ u32x4 join_u(const u32x2 &a, const u32x2 &b) { return u32x4{a.x, a.y, b.x, b.y}; }
__device__ __forceinline__ bf16x8 join_frag(const u32x2 &a, const u32x2 &b) {
    u32x4 w = {a.x, a.y, b.x, b.y};
    return *reinterpret_cast<bf16x8 *>(&w);
}

}  // namespace

// the round-1 kernel (8 waves x 32 rows), debug / cross-check paths only: duo_prefill_w32_debug.hip
// (`params`: a PrefillParams — the type has internal linkage, both translation units see the same definition from this header)
int duo_prefill_w32_launch(const void *params, bool tr, bool f16, int nblk, int n_batch, int dev, hipStream_t st);
