// duo_prefill.hip — chunked-prefill flash attention for both DuoAttention head
// classes in one launch (gfx950, bf16 MFMA 32x32x16).
//
// Replaces flash_attn_func at duo_attn/patch/llama.py:366-372 (first chunk: all
// heads causal over the chunk) and llama.py:392-421 (later chunks: retrieval
// heads over the whole full-KV pool, streaming heads over
// [sink+recent pool rows ++ the chunk]).  Semantics: keys = segA (all visible)
// ++ segB (the S new rows, causal: query i sees j <= i) — flash-attn's
// bottom-right aligned causal mask; fp32 scores/softmax, P rounded to bf16
// before P.V (as FA2 does), bf16 output.
//
// Structure (one workgroup = 256 query rows of one q head, 8 waves x 32 rows):
//   * "swapped" QK^T: S^T[key][q] = K . Q^T, so each lane owns one query row
//     (lane&31) and the row max / row sum are lane-local plus ONE exchange with
//     lane^32;  Q fragments live in registers for the whole kernel;
//   * O^T[d][q] = V^T . P^T: the softmax scale factors stay lane-local too and
//     P^T feeds the MFMA B operand straight from the score registers (the key
//     order inside a 16-key step is permuted identically on the V^T side);
//   * K tile [64][128] in LDS, 16-B chunks XOR-swizzled by (row & 15) ->
//     conflict-free ds_read_b128 for the A operand;
//   * V tile in LDS as [key/4][dim/16][4][16] blocks read with
//     ds_read_b64_tr_b16 (hardware transpose) -> V^T A operand with no shuffles;
//   * K/V tiles double-buffered in LDS, the next tile's global loads are issued
//     before the MFMA work on the current one and written to LDS after it
//     (register staging; one barrier per tile);
//   * causal tiles beyond a wave's last row are skipped per wave; blocks are
//     ordered heaviest-first, and the q heads that share a kv head are mapped to
//     the same XCD (block id % 8) so K/V tiles are shared through one L2.
#include "duo_common.h"

namespace {

constexpr int QBLK = 256;   // query rows per workgroup
constexpr int KVBLK = 64;   // keys per tile
constexpr int NWAVE = 8;
constexpr int K_TILE_BYTES = KVBLK * DUO_HEAD_DIM * 2;  // 16 KiB
constexpr int V_TILE_BYTES = K_TILE_BYTES;
constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;               // 64 KiB
constexpr float kDeferLog2 = 8.0f;   // deferred-rescale threshold in the exp2 domain

struct PrefillParams {
    const bf16_t *q;
    int64_t q_ts, q_hs;
    bf16_t *out;
    int64_t o_ts, o_hs;
    int32_t S;
    int32_t group;
    int32_t n_qtiles;
    int32_t nblk_full;     // cls[0] q heads * n_qtiles
    DuoClassDev cls[2];
    float scale_log2e;
    uint32_t flags;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    f32x2 v = {lo, hi};
    hw_bf16x2 r = __builtin_convertvector(v, hw_bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
    return *reinterpret_cast<uint32_t *>(&r);
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

// 512 threads fetch one 64x128 K tile and one V tile: 2 x 16 B of each per thread.
// K: thread -> (row = idx/16, 16-B chunk = idx%16): 16 lanes read one 256-B row.
// V: within each wave the lanes are permuted so that 8 consecutive lanes hold one 128-B LDS block
//    ([4 keys][16 dims]) -> conflict-free ds_write_b128; a wave still reads 4 whole rows.
__device__ __forceinline__ void v_stage_coord(int idx, int &row, int &ch) {
    const int l = idx & 63;
    row = ((idx >> 6) << 2) + ((l & 7) >> 1);     // 4 rows per wave-load
    ch = ((l >> 3) << 1) + (l & 1);               // dim block (l>>3), half (l&1)
}
__device__ __forceinline__ void stage_load(const TileSrc &s, int tid, u32x4 (&kr)[2], u32x4 (&vr)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 512 * j;
        const int row = idx >> 4;
        const int ch = idx & 15;
        const int r = s.row0 + min(row, s.cnt - 1);
        kr[j] = *reinterpret_cast<const u32x4 *>(s.k + (int64_t)r * s.ts + ch * 8);
        int vrow, vch;
        v_stage_coord(idx, vrow, vch);
        const int rv = s.row0 + min(vrow, s.cnt - 1);
        vr[j] = *reinterpret_cast<const u32x4 *>(s.v + (int64_t)rv * s.ts + vch * 8);
    }
}

__device__ __forceinline__ int k_lds_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
// V image: [key/4][dim/16][4 keys][16 dims] bf16, 128-B blocks
__device__ __forceinline__ int v_lds_off(int row, int d) {
    return (((row >> 2) * 8 + (d >> 4)) << 7) + ((row & 3) << 5) + ((d & 15) << 1);
}

__device__ __forceinline__ void stage_write(char *stage, int tid, const u32x4 (&kr)[2], const u32x4 (&vr)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 512 * j;
        const int row = idx >> 4;
        const int ch = idx & 15;
        *reinterpret_cast<u32x4 *>(stage + k_lds_off(row, ch)) = kr[j];
        int vrow, vch;
        v_stage_coord(idx, vrow, vch);
        *reinterpret_cast<u32x4 *>(stage + K_TILE_BYTES + v_lds_off(vrow, vch * 8)) = vr[j];
    }
}

template <bool USE_TR>
__device__ __forceinline__ bf16x8 load_vt_frag(const char *vst, int kq, int blk16, int lane15) {
    // keys 4*kq..4*kq+3 and 4*(kq+2)..4*(kq+2)+3 of dim column (blk16*16 + lane15)
    const int b0 = ((kq * 8 + blk16) << 7);
    const int b1 = (((kq + 2) * 8 + blk16) << 7);
    bf16x8 r;
    if constexpr (USE_TR) {
        // each lane of a 16-lane group points at its 8-byte piece of the
        // row-major 4x16 block; the hardware hands lane i column i
        const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(vst + b0 + lane15 * 8));
        const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(vst + b1 + lane15 * 8));
        r[0] = x[0]; r[1] = x[1]; r[2] = x[2]; r[3] = x[3];
        r[4] = y[0]; r[5] = y[1]; r[6] = y[2]; r[7] = y[3];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = *reinterpret_cast<const short *>(vst + b0 + j * 32 + lane15 * 2);
            r[4 + j] = *reinterpret_cast<const short *>(vst + b1 + j * 32 + lane15 * 2);
        }
    }
    return r;
}

template <bool USE_TR>
__global__ __launch_bounds__(512) void duo_prefill_kernel(const PrefillParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int lane15 = lane & 15;

    // ---- block -> (class, q tile, kv head, q head) --------------------------
    int b = blockIdx.x;
    const int ci = b < P.nblk_full ? 0 : 1;
    if (ci) b -= P.nblk_full;
    const DuoClassDev &C = P.cls[ci];
    const int nq_c = C.n_kv_heads * P.group;
    const int tile = P.n_qtiles - 1 - b / nq_c;   // heaviest (latest) tiles first
    const int p = b % nq_c;
    const int kvh = p % C.n_kv_heads;             // group mates sit 8 blocks apart -> same XCD
    const int g = p / C.n_kv_heads;
    const int qh = C.q_head_offset + kvh * P.group + g;

    const int S = P.S;
    const int q0 = tile * QBLK;
    const int wq0 = q0 + wave * 32;               // first query row of this wave
    const int my_q = wq0 + l31;
    const int my_q_ld = min(my_q, S - 1);

    // ---- Q fragments (B operand of the swapped QK^T) --------------------------
    bf16x8 qfrag[8];
    {
        const bf16_t *qp = P.q + (int64_t)my_q_ld * P.q_ts + (int64_t)qh * P.q_hs + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qfrag[kk] = *reinterpret_cast<const bf16x8 *>(qp + kk * 16);
    }

    const int lenA = C.a.len;
    const int nA = (lenA + KVBLK - 1) / KVBLK;
    const int last_q = min(q0 + QBLK - 1, S - 1);
    const int nB = last_q / KVBLK + 1;
    const int nT = nA + nB;

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float mrow = -INFINITY;
    float lsum = 0.f;
    const float c = P.scale_log2e;

    u32x4 kr[2], vr[2];
    {
        const TileSrc s0 = tile_src(C, kvh, 0, nA, S);
        stage_load(s0, tid, kr, vr);
        stage_write(smem, tid, kr, vr);
    }
    __syncthreads();

    for (int t = 0; t < nT; ++t) {
        char *stage = smem + (t & 1) * STAGE_BYTES;
        const bool has_next = t + 1 < nT;
        if (has_next) {
            const TileSrc sn = tile_src(C, kvh, t + 1, nA, S);
            stage_load(sn, tid, kr, vr);
        }

        const bool inB = t >= nA;
        const int key0 = inB ? (t - nA) * KVBLK : t * KVBLK;   // first key of the tile in its segment
        const int cnt = inB ? min(KVBLK, S - key0) : min(KVBLK, lenA - key0);
        // a causal tile that starts after this wave's last row contributes nothing
        const bool skip = inB && key0 > wq0 + 31;

        if (!skip) {
            // ---- S^T = K . Q^T  (two 32-key blocks) ---------------------------
            f32x16 sc[2];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[bb][r] = 0.f;
                const int row = bb * 32 + l31;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(stage + k_lds_off(row, 2 * kk + hi));
                    sc[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfrag[kk], sc[bb], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            // ---- mask ----------------------------------------------------------
            const bool need_mask = inB ? (key0 + KVBLK - 1 > wq0) : (cnt < KVBLK);
            if (need_mask) {
                const int lim = inB ? min(my_q - key0, cnt - 1) : cnt - 1;   // last visible key (tile-local)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = bb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (kl > lim) sc[bb][r] = -INFINITY;
                    }
            }
            // ---- online softmax (lane = one query row; partner lane^32 holds the other keys)
            float t0 = fmaxf(fmaxf(sc[0][0], sc[0][1]), sc[0][2]);
            float t1 = fmaxf(fmaxf(sc[1][0], sc[1][1]), sc[1][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) {
                t0 = fmaxf(fmaxf(t0, sc[0][r]), sc[0][r + 1]);
                t1 = fmaxf(fmaxf(t1, sc[1][r]), sc[1][r + 1]);
            }
            float tmax = fmaxf(fmaxf(t0, t1), fmaxf(sc[0][15], sc[1][15]));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            // Deferred rescale: while no row of the wave grows its max by more than 2^kDeferLog2 the
            // old reference point is kept (P <= 2^kDeferLog2, exact in fp32/bf16 ranges) and the
            // 64-register O rescale is skipped.  First tile: mrow = -inf forces the rescale path.
            if (!__all((tmax - mrow) * c <= kDeferLog2)) {
                const float mnew = fmaxf(mrow, tmax);
                const float alpha = fast_exp2((mrow - mnew) * c);
                lsum *= alpha;
                mrow = mnew;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
            const float mc = mrow * c;
            float psum = 0.f;
            bf16x8 pf[2][2];
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = fast_exp2(fmaf(sc[bb][r], c, -mc));
                    psum += pv[r];
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 w;
                    w.x = cvt_pk_bf16(pv[8 * s + 0], pv[8 * s + 1]);
                    w.y = cvt_pk_bf16(pv[8 * s + 2], pv[8 * s + 3]);
                    w.z = cvt_pk_bf16(pv[8 * s + 4], pv[8 * s + 5]);
                    w.w = cvt_pk_bf16(pv[8 * s + 6], pv[8 * s + 7]);
                    pf[bb][s] = *reinterpret_cast<bf16x8 *>(&w);
                }
            }
            lsum += psum;

            // ---- O^T += V^T . P^T ----------------------------------------------
            const char *vst = stage + K_TILE_BYTES;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const int blk16 = 2 * db + (l31 >> 4);
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int kq = 8 * bb + 4 * s + hi;
                        const bf16x8 vf = load_vt_frag<USE_TR>(vst, kq, blk16, lane15);
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[bb][s], o[db], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_s_setprio(0);
        }

        if (has_next) stage_write(smem + ((t + 1) & 1) * STAGE_BYTES, tid, kr, vr);
        __syncthreads();
    }

    // ---- epilogue: O^T / l -> out[q][qh][d] -----------------------------------
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.f / lsum;
    if (my_q < S) {
        bf16_t *op = P.out + (int64_t)my_q * P.o_ts + (int64_t)qh * P.o_hs;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = 32 * db + 8 * rq + 4 * hi;   // rows (r&3)+8*(r>>2)+4*hi, r = 4rq..4rq+3
                u32x2 w;
                w.x = cvt_pk_bf16(o[db][4 * rq + 0] * inv, o[db][4 * rq + 1] * inv);
                w.y = cvt_pk_bf16(o[db][4 * rq + 2] * inv, o[db][4 * rq + 3] * inv);
                *reinterpret_cast<u32x2 *>(op + d) = w;
            }
    }
}

}  // namespace

static uint32_t g_debug_flags = 0;
extern "C" void duo_set_debug_flags(uint32_t flags) { g_debug_flags = flags; }
extern "C" uint32_t duo_get_debug_flags(void) { return g_debug_flags; }

extern "C" int duo_attn_prefill_bf16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                     void *out, int64_t out_token_stride, int64_t out_head_stride,
                                     int32_t n_tokens, int32_t group, const duo_head_class *full,
                                     const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                     void *stream) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (q == nullptr || out == nullptr || group <= 0 || n_tokens < 0) return DUO_EINVAL;
    if (n_tokens == 0) return 0;
    PrefillParams P;
    P.q = (const bf16_t *)q;
    P.q_ts = q_token_stride;
    P.q_hs = q_head_stride;
    P.out = (bf16_t *)out;
    P.o_ts = out_token_stride;
    P.o_hs = out_head_stride;
    P.S = n_tokens;
    P.group = group;
    P.n_qtiles = (n_tokens + QBLK - 1) / QBLK;
    P.cls[0] = duo_class_dev(full);
    P.cls[1] = duo_class_dev(stream_cls);
    P.scale_log2e = scale * 1.4426950408889634f;
    P.flags = g_debug_flags;
    int nblk = 0;
    for (int c = 0; c < 2; ++c) {
        DuoClassDev &C = P.cls[c];
        if (C.n_kv_heads <= 0) { C.n_kv_heads = 0; continue; }
        if (C.b.len != n_tokens || !C.b.k || !C.b.v) return DUO_EINVAL;   // segB is the chunk itself
        if (C.a.len < 0 || (C.a.len > 0 && (!C.a.k || !C.a.v))) return DUO_EINVAL;
        if ((C.a.token_stride | C.a.head_stride | C.b.token_stride | C.b.head_stride) & 7) {
            // rows must be 16-byte aligned for the dwordx4 tile loads
            if (C.a.len > 0 || ((C.b.token_stride | C.b.head_stride) & 7)) return DUO_EINVAL;
        }
        nblk += C.n_kv_heads * group * P.n_qtiles;
    }
    if ((q_token_stride | q_head_stride) & 7) return DUO_EINVAL;
    if ((out_token_stride | out_head_stride) & 3) return DUO_EINVAL;
    if (nblk == 0) return 0;
    P.nblk_full = P.cls[0].n_kv_heads * group * P.n_qtiles;

    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (g_debug_flags & 1u) {
        e = hipFuncSetAttribute((const void *)duo_prefill_kernel<false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(duo_prefill_kernel<false>, dim3(nblk), dim3(512), LDS_BYTES, st, P);
    } else {
        e = hipFuncSetAttribute((const void *)duo_prefill_kernel<true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(duo_prefill_kernel<true>, dim3(nblk), dim3(512), LDS_BYTES, st, P);
    }
    DUO_HIP_CHECK_LAUNCH();
    return 0;
}
