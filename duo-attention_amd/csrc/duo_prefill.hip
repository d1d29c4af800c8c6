// duo_prefill.hip — chunked-prefill flash attention for both DuoAttention head
// classes in one launch (gfx950, bf16 MFMA 32x32x16): the launcher / C entry points of BOTH prefill kernels, the
// key-range-split merge kernel, and the 8-wave x 32-row kernel.
//
// Which kernel runs: every product launch runs duo_prefill_w64_kernel (4 waves x 64 rows, duo_prefill_w64.h) since
// round 2.  The 8-wave kernel below is kept as ONE clean body for two purposes only: the gather (non-transposed-LDS)
// debug layout behind debug bit 0, and the same-box A/B against the w64 kernel (debug bit 7, DUO_PREFILL_W64=0); the GPU
// tests run every prefill case on both.  (Its measurement / ablation switches of rounds 1-2 were resolved out of this
// file in round 3: the object code did not change by a byte.)
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
//   * deferred rescale: the running max only moves (and O is only rescaled)
//     when some row's max grew by more than 2^8;
//   * K tile [64][128] in LDS, 16-B chunks XOR-swizzled by (row & 15) ->
//     conflict-free ds_read_b128 for the A operand;
//   * V tile in LDS as [key/4][dim/16][4][16] blocks read with
//     ds_read_b64_tr_b16 (hardware transpose) -> V^T A operand with no shuffles;
//   * K/V tiles arrive by LDS-DMA (global_load_lds_dwordx4, swizzle / block
//     layout applied on the per-lane source address) into a THREE-deep LDS ring:
//     tile t+2 is requested while tile t is consumed, and the only wait is a
//     counted s_waitcnt vmcnt(4) (= "tile t+1 has landed") in front of ONE raw
//     s_barrier per tile — the loads stay in flight across the barrier;
//   * causal tiles beyond a wave's last row are skipped per wave; blocks are
//     ordered heaviest-first, and the q heads that share a kv head are mapped to
//     the same XCD (block id % 8) so K/V tiles are shared through one L2.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "duo_prefill_common.h"
#include "duo_prefill_w64.h"

namespace {

template <bool USE_TR, bool F16>
__global__ __launch_bounds__(512) void duo_prefill_kernel(const PrefillParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int lane15 = lane & 15;

    // ---- block -> (class, q tile, kv head, q head) --------------------------
    int b = blockIdx.x;
    const int ci = b < P.nblk_full ? 0 : 1;
    if (ci) b -= P.nblk_full;
    const int by = blockIdx.y;        // batch row
    DuoClassDev Crow = duo_select(P.cls[0], P.cls[1], ci != 0);
    duo_class_batch_row(Crow, by);
    const DuoClassDev C = Crow;
    // key-range split (retrieval class only): the splits of one (q tile, q head) are adjacent block ids
    const int ks = ci == 0 ? P.ksplit : 1;
    const int split = b % ks;
    const int part_id = b + by * P.nblk_full;   // index of this workgroup's partial in the workspace
    b /= ks;
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
        const bf16_t *qp = P.q + (int64_t)by * P.q_bs + (int64_t)my_q_ld * P.q_ts + (int64_t)qh * P.q_hs + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qfrag[kk] = *reinterpret_cast<const bf16x8 *>(qp + kk * 16);
    }

    const int lenA = C.a.len;
    const int nA = (lenA + KVBLK - 1) / KVBLK;
    // segment B may be longer than the query block: the S queries are its LAST S rows (bottom-right
    // causal alignment, as flash_attn_func with seqlen_q < seqlen_k) — query i sees B keys 0 .. i + qoff.
    // qoff > 0 is how a chunk is processed in row blocks (layer-pipeline wavefront): queries [r0, r1) of
    // the chunk against chunk rows [0, r1).
    const int lenB = C.b.len;
    const int qoff = lenB - S;
    const int last_q = min(q0 + QBLK - 1, S - 1);
    const int nB = (last_q + qoff) / KVBLK + 1;
    const int nT_all = nA + nB;
    // this workgroup's share of the tile sequence (segment A tiles, then the causal tiles of segment B)
    const int t_begin = (int)((int64_t)split * nT_all / ks);
    const int nT = (int)((int64_t)(split + 1) * nT_all / ks);   // exclusive end: the loops below run [t_begin, nT)

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float mrow = -INFINITY;
    float lsum = 0.f;
    const float c = P.scale_log2e;

    // ---- loop invariants: LDS read offsets and DMA lane offsets -----------------
    const uint32_t smem_lds = lds_addr(smem);
    uint32_t koff[8];    // K fragment of k-step kk, key block 0 (block 1: +8192), ring slot 0
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) koff[kk] = smem_lds + k_lds_off(l31, 2 * kk + hi);
    // V^T fragment base: key quad hi, dim block (l31>>4), this lane's 8-byte piece of the 4x16 block
    const uint32_t vaddr = smem_lds + K_TILE_BYTES + hi * 1024 + (l31 >> 4) * 128 + lane15 * 8;
    const DmaLane dmaA = dma_lane(tid, C.a.token_stride);
    const DmaLane dmaB = dma_lane(tid, C.b.token_stride);

    auto issue_dma = [&](int t, int slot_) {
        const TileSrc ts_ = tile_src(C, kvh, t, nA, lenB);
        const uint32_t dst = smem_lds + slot_ * STAGE_BYTES;
        if (ts_.cnt == KVBLK) stage_dma_full<8>(ts_, t < nA ? dmaA : dmaB, dst, tid);
        else stage_dma_tail<8>(ts_, dst, tid);
    };

    // ---- prologue: tiles 0 and 1 in flight, wait for tile 0 only ---------------
    // (the Q loads above are older in the VMEM queue, so either wait also covers them)
    if (t_begin < nT) issue_dma(t_begin, 0);
    if (t_begin + 1 < nT) {
        issue_dma(t_begin + 1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // Touch the Q fragments here so that hipcc waits for their loads NOW.  Otherwise it places the
    // s_waitcnt vmcnt ladder at their first use inside the loop, where it re-executes every
    // iteration and drains the (asm-issued, to it invisible) LDS-DMA each time.
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" ::"v"(qfrag[kk]));
    __builtin_amdgcn_sched_barrier(0);

#define DUO_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
    // One tile.  SLOT (= t % 3) is a compile-time constant so that every LDS address of the body is
    // a loop-invariant VGPR plus an immediate: the tile loop is unrolled by the ring depth.
    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int SOFF = SLOT * STAGE_BYTES;
        const bool more2 = t + 2 < nT;
        // ring slot (t+2)%3 == (t-1)%3 was last read in iteration t-1, which every wave left through
        // that iteration's barrier
        if (more2) issue_dma(t + 2, (SLOT + 2) % NSTAGE);
        const bool inB = t >= nA;
        const int key0 = inB ? (t - nA) * KVBLK : t * KVBLK;   // first key of the tile in its segment
        const int cnt = inB ? min(KVBLK, lenB - key0) : min(KVBLK, lenA - key0);
        // a causal tile that starts after this wave's last row contributes nothing
        const bool skip = inB && key0 > wq0 + qoff + 31;

        if (!skip) {
            // ---- S^T = K . Q^T  (two 32-key blocks) ---------------------------
            f32x16 sc[2];
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            DUO_SETPRIO(1);
            // kk outer, key block inner: consecutive MFMAs alternate between the two accumulators.
            // The K fragments of k-step kk+1 are requested BEFORE the MFMAs of k-step kk (two register
            // sets); left to hipcc, each step's two ds_read_b128 are issued into the same registers only after
            // the previous step's MFMAs, so every step pays the LDS latency.  asm reads + counted lgkmcnt, as
            // for the V^T reads below.
            {
                constexpr int KO = SOFF >= 32768 ? 0 : SOFF;          // 16-bit ds offset field
                u32x4 kf[2][2];
#define DUO_K_READ(dst, kk_, bb_)                                                                         \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(koff[kk_] + (SOFF >= 32768 ? SOFF : 0)),  \
                 "n"(KO + (bb_) * 8192) : "memory")
                __builtin_amdgcn_sched_barrier(0);
                DUO_K_READ(kf[0][0], 0, 0);
                DUO_K_READ(kf[0][1], 0, 1);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if (kk < 7) {
                        DUO_K_READ(kf[(kk + 1) & 1][0], kk + 1, 0);
                        DUO_K_READ(kf[(kk + 1) & 1][1], kk + 1, 1);
                        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
                        sc[bb] = mfma32x32x16<F16>(*reinterpret_cast<const bf16x8 *>(&kf[kk & 1][bb]), qfrag[kk],
                                                   kk == 0 ? zero16 : sc[bb]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef DUO_K_READ
            }
            DUO_SETPRIO(0);
            // ---- mask ----------------------------------------------------------
            const bool need_mask = inB ? (key0 + KVBLK - 1 > wq0 + qoff) : (cnt < KVBLK);
            if (need_mask) {
                const int lim = inB ? min(my_q + qoff - key0, cnt - 1) : cnt - 1;   // last visible key (tile-local)
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
            // row max over both 32-key halves: lanes l and l^32 hold the two halves of a query row.
            // v_permlane32_swap exchanges the upper half of one register with the lower half of another in the
            // VALU — no trip through the LDS crossbar (ds_bpermute) on the per-tile critical path
            {
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                tmax = fmaxf(__uint_as_float(sw.x), __uint_as_float(sw.y));
            }
            // Deferred rescale: while no row of the wave grows its max by more than 2^kDeferLog2 the
            // old reference point is kept (P <= 2^kDeferLog2, exact in fp32/bf16 ranges) and the
            // 64-register O rescale is skipped.  First tile: mrow = -inf forces the rescale path.
            if (!__all((tmax - mrow) * c <= kDeferLog2)) {
                const float mnew = fmaxf(mrow, tmax);
                // (a row that has seen no key yet — possible when a key-range split starts on causal tiles
                // beyond it — keeps m = -inf; -inf - -inf must not reach exp2)
                const float alpha = mnew == -INFINITY ? 1.f : fast_exp2((mrow - mnew) * c);
                lsum *= alpha;
                mrow = mnew;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
            const float mc = mrow == -INFINITY ? 0.f : mrow * c;   // all scores -inf: p = exp2(-inf - 0) = 0
            float psum = 0.f;
            bf16x8 pf[4];   // P^T B operands of the four PV k-steps (step = 2*bb + s)
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
                    w.x = cvt_pk16<F16>(pv[8 * s + 0], pv[8 * s + 1]);
                    w.y = cvt_pk16<F16>(pv[8 * s + 2], pv[8 * s + 3]);
                    w.z = cvt_pk16<F16>(pv[8 * s + 4], pv[8 * s + 5]);
                    w.w = cvt_pk16<F16>(pv[8 * s + 6], pv[8 * s + 7]);
                    pf[2 * bb + s] = *reinterpret_cast<bf16x8 *>(&w);
                }
            }
            lsum += psum;

            // ---- O^T += V^T . P^T ----------------------------------------------
            // k-step `step` covers keys 32*bb + 16*s + {4hi..4hi+3, 8+4hi..8+4hi+3}: key quads
            // kq = 4*step + hi and kq + 2; quad kq / dim block blk16 sits at byte (kq*8 + blk16)*128.
            if constexpr (USE_TR) {
                // hand-pipelined: the 8 transpose reads of k-step n+1 are issued before the 4 MFMAs of
                // k-step n, completion counted with lgkmcnt (asm loads are invisible to hipcc's waitcnt
                // pass, rule 18: sched_barrier after each wait).
                // the ds_read offset field is 16 bits: slot 2 needs its base folded into the address
                const uint32_t va_ = SOFF >= 32768 ? vaddr + SOFF : vaddr;
                constexpr int VO = SOFF >= 32768 ? 0 : SOFF;
                u32x2 va[8], vb[8];
                __builtin_amdgcn_sched_barrier(0);
                DUO_TR_STEP(va, va_, VO, 0);
                DUO_TR_STEP(vb, va_, VO, 1);
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                DUO_SETPRIO(1);
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    o[db] = mfma32x32x16<F16>(join_frag(va[2 * db], va[2 * db + 1]), pf[0], o[db]);
                __builtin_amdgcn_sched_barrier(0);
                DUO_TR_STEP(va, va_, VO, 2);
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    o[db] = mfma32x32x16<F16>(join_frag(vb[2 * db], vb[2 * db + 1]), pf[1], o[db]);
                __builtin_amdgcn_sched_barrier(0);
                DUO_TR_STEP(vb, va_, VO, 3);
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    o[db] = mfma32x32x16<F16>(join_frag(va[2 * db], va[2 * db + 1]), pf[2], o[db]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    o[db] = mfma32x32x16<F16>(join_frag(vb[2 * db], vb[2 * db + 1]), pf[3], o[db]);
                DUO_SETPRIO(0);
            } else {
                // debugging aid (duo_set_debug_flags bit 0): scalar LDS gathers instead of the transpose read
                const char *vst = smem + SOFF + K_TILE_BYTES;
#pragma unroll
                for (int step = 0; step < 4; ++step)
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        const int blk16 = 2 * db + (l31 >> 4);
                        const int b0 = (((4 * step + hi) * 8 + blk16) << 7);
                        bf16x8 vf;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            vf[j] = *reinterpret_cast<const short *>(vst + b0 + j * 32 + lane15 * 2);
                            vf[4 + j] = *reinterpret_cast<const short *>(vst + b0 + 2048 + j * 32 + lane15 * 2);
                        }
                        o[db] = mfma32x32x16<F16>(vf, pf[step], o[db]);
                    }
            }
        }

        // ---- tile t+1 must have landed (tile t+2 may stay in flight), then ONE barrier:
        //      it publishes tile t+1 and retires every read of ring slot t%3
        __builtin_amdgcn_sched_barrier(0);
        if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int t = t_begin; t < nT; t += NSTAGE) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nT) tile_body(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < nT) tile_body(std::integral_constant<int, 2>{}, t + 2);
    }

    // ---- epilogue: O^T / l -> out[q][qh][d], or the un-normalised partial -> workspace ---------
    lsum += __shfl_xor(lsum, 32);
    if (ks > 1) {
        const int64_t row = (int64_t)part_id * QBLK + wave * 32 + l31;
        float *wo = P.ws_o + row * DUO_HEAD_DIM;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = 32 * db + 8 * rq + 4 * hi;
                const f32x4 w = {o[db][4 * rq + 0], o[db][4 * rq + 1], o[db][4 * rq + 2], o[db][4 * rq + 3]};
                *reinterpret_cast<f32x4 *>(wo + d) = w;
            }
        if (hi == 0) {
            P.ws_ml[row * 2 + 0] = mrow;
            P.ws_ml[row * 2 + 1] = lsum;
        }
        return;
    }
    const float inv = 1.f / lsum;
    if (my_q < S) {
        bf16_t *op = P.out + (int64_t)by * P.o_bs + (int64_t)my_q * P.o_ts + (int64_t)qh * P.o_hs;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = 32 * db + 8 * rq + 4 * hi;   // rows (r&3)+8*(r>>2)+4*hi, r = 4rq..4rq+3
                u32x2 w;
                w.x = cvt_pk16<F16>(o[db][4 * rq + 0] * inv, o[db][4 * rq + 1] * inv);
                w.y = cvt_pk16<F16>(o[db][4 * rq + 2] * inv, o[db][4 * rq + 3] * inv);
                *reinterpret_cast<u32x2 *>(op + d) = w;
            }
    }
}

// Combine the `ksplit` partials of one (q tile, q head): out = sum_s 2^((m_s - M) c) O_s / sum_s 2^((m_s - M) c) l_s.
// Block = (q tile, q head of the retrieval class) in the prefill kernel's block order; thread = 32 rows x 8
// column groups of 16 dims per pass, 8 passes.
template <bool F16>
__global__ __launch_bounds__(256) void duo_prefill_merge_kernel(const PrefillParams P) {
    const DuoClassDev C = P.cls[0];
    const int nq_c = C.n_kv_heads * P.group;
    const int b = blockIdx.x;
    const int tile = P.n_qtiles - 1 - b / nq_c;
    const int p = b % nq_c;
    const int kvh = p % C.n_kv_heads;
    const int g = p / C.n_kv_heads;
    const int qh = C.q_head_offset + kvh * P.group + g;
    const int ks = P.ksplit;
    const int j = threadIdx.x & 7;
    for (int pass = 0; pass < 8; ++pass) {
        const int r = pass * 32 + (threadIdx.x >> 3);
        const int q = tile * QBLK + r;
        if (q >= P.S) continue;
        const int64_t row0 = ((int64_t)b * ks + (int64_t)blockIdx.y * P.nblk_full) * QBLK + r;     // split s: + s * QBLK
        float M = -INFINITY;
        for (int s = 0; s < ks; ++s) M = fmaxf(M, P.ws_ml[(row0 + (int64_t)s * QBLK) * 2]);
        float L = 0.f;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < ks; ++s) {
            const int64_t row = row0 + (int64_t)s * QBLK;
            const float m = P.ws_ml[row * 2], l = P.ws_ml[row * 2 + 1];
            if (m == -INFINITY) continue;      // this split saw no key of the row: nothing to add
            const float w = fast_exp2((m - M) * P.scale_log2e);
            L = fmaf(l, w, L);
            const f32x4 *src = reinterpret_cast<const f32x4 *>(P.ws_o + row * DUO_HEAD_DIM + 16 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = acc[i] + src[i] * w;
        }
        const float inv = 1.f / L;
        bf16_t *op = P.out + (int64_t)blockIdx.y * P.o_bs + (int64_t)q * P.o_ts + (int64_t)qh * P.o_hs + 16 * j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 w2;
            w2.x = cvt_pk16<F16>(acc[i].x * inv, acc[i].y * inv);
            w2.y = cvt_pk16<F16>(acc[i].z * inv, acc[i].w * inv);
            *reinterpret_cast<u32x2 *>(op + 4 * i) = w2;
        }
    }
}

}  // namespace

static uint32_t g_debug_flags = 0;
extern "C" void duo_set_debug_flags(uint32_t flags) { g_debug_flags = flags; }
extern "C" uint32_t duo_get_debug_flags(void) { return g_debug_flags; }

// Key-range splits of the retrieval class.  One workgroup per CU is resident (96 KiB of LDS), so a launch
// whose long workgroups (retrieval q heads x q tiles) are fewer than the CUs — small chunks, layers with
// one or two retrieval kv heads — or not a multiple of them leaves CUs idle for the whole launch while
// the streaming-head workgroups are short.  With k splits the long work becomes L0*k workgroups of 1/k the
// length: pick the k that minimises rounds(L0*k) / k, with a small charge per split for the merge pass.
constexpr int64_t kPrefillPartialBytes = (int64_t)QBLK * (DUO_HEAD_DIM + 2) * sizeof(float);
static int prefill_choose_ksplit(int long_wgs, int min_tiles, int64_t workspace_bytes) {
    static const int forced = [] {
        const char *e = getenv("DUO_PREFILL_KSPLIT");   // tuning / test knob: force a split count
        return e ? atoi(e) : 0;
    }();
    if (long_wgs <= 0 || workspace_bytes <= 0) return 1;
    const int kmax = (int)std::min<int64_t>(8, std::min<int64_t>(min_tiles, workspace_bytes / (kPrefillPartialBytes * long_wgs)));
    if (g_debug_flags & 256u) return 1;      // debug bit 8: no key-range split
    if (forced > 0) return std::max(1, std::min(forced, kmax));
    if ((g_debug_flags >> 12) & 15u) return std::max(1, std::min((int)((g_debug_flags >> 12) & 15u), kmax));   // bits 12-15: tests force a count
    int best = 1;
    double best_cost = 1e30;
    for (int k = 1; k <= kmax; ++k) {
        const double rounds = (double)((long_wgs * k + 255) / 256);
        const double cost = rounds / k + 0.03 * (k - 1);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = k; }
    }
    return best;
}

extern "C" int64_t duo_attn_prefill_workspace_bytes(void) { return 1024 * kPrefillPartialBytes; }

template <bool F16>
static int prefill_impl(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                        void *out, int64_t out_token_stride, int64_t out_head_stride,
                        int32_t n_tokens, int32_t group, const duo_head_class *full,
                        const duo_head_class *stream_cls, float scale, int32_t head_dim,
                        void *workspace, int64_t workspace_bytes, void *stream,
                        int32_t n_batch = 1, int64_t q_batch_stride = 0, int64_t out_batch_stride = 0) {
    if (head_dim != DUO_HEAD_DIM) return DUO_EHEADDIM;
    if (q == nullptr || out == nullptr || group <= 0 || n_tokens < 0 || n_batch < 0 || n_batch > 65535) return DUO_EINVAL;
    if (n_tokens == 0 || n_batch == 0) return 0;
    if (n_batch > 1 && ((q_batch_stride | out_batch_stride) & 7)) return DUO_EINVAL;
    PrefillParams P;
    P.q_bs = q_batch_stride;
    P.o_bs = out_batch_stride;
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
        if (C.b.len < n_tokens || !C.b.k || !C.b.v) return DUO_EINVAL;   // segB ends with the query rows
        if (C.a.len < 0 || (C.a.len > 0 && (!C.a.k || !C.a.v))) return DUO_EINVAL;
        if ((C.a.token_stride | C.a.head_stride | C.b.token_stride | C.b.head_stride) & 7) {
            // rows must be 16-byte aligned for the dwordx4 tile loads
            if (C.a.len > 0 || ((C.b.token_stride | C.b.head_stride) & 7)) return DUO_EINVAL;
        }
        nblk += C.n_kv_heads * group * P.n_qtiles;
    }
    // split the retrieval class when that fills the chip better (needs the caller's workspace)
    const int long_wgs = P.cls[0].n_kv_heads * group * P.n_qtiles;
    P.ksplit = 1;
    P.ws_o = nullptr;
    P.ws_ml = nullptr;
    if (workspace && long_wgs > 0) {
        const int min_tiles = (P.cls[0].a.len + KVBLK - 1) / KVBLK + 1;   // tiles of the first q tile
        P.ksplit = prefill_choose_ksplit(long_wgs, min_tiles, workspace_bytes / n_batch);   // every batch row has its own partials
        if (P.ksplit > 1) {
            P.ws_o = (float *)workspace;
            P.ws_ml = P.ws_o + (int64_t)n_batch * long_wgs * P.ksplit * QBLK * DUO_HEAD_DIM;
            nblk += long_wgs * (P.ksplit - 1);
        }
    }
    if ((q_token_stride | q_head_stride) & 7) return DUO_EINVAL;
    if ((out_token_stride | out_head_stride) & 3) return DUO_EINVAL;
    if (nblk == 0) return 0;
    P.nblk_full = long_wgs * P.ksplit;
    P.xmap_rows = P.xmap_q = 0;

    hipStream_t st = (hipStream_t)stream;
    const bool tr = !(g_debug_flags & 1u);
    // hipFuncSetAttribute is cheap but not free: once per (device, kernel instantiation).  The attribute belongs to
    // the function as loaded on ONE device, so a process that drives several GPUs (layer pipeline in one process,
    // accelerate-style placement) must set it on each; atomics because any host thread may get here.
    static std::atomic<bool> attr_done[64][2];   // [device][transpose-read variant], per element type (template)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return DUO_EINVAL;
    if (dev >= 64 || !attr_done[dev][tr].load(std::memory_order_acquire)) {
        const void *fn = tr ? (const void *)duo_prefill_kernel<true, F16> : (const void *)duo_prefill_kernel<false, F16>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        if (dev < 64) attr_done[dev][tr].store(true, std::memory_order_release);
    }
    // 4-wave x 64-row kernel (duo_prefill_w64.h): the default for bf16 and fp16 with the
    // transposed-V LDS layout; the gather debug path runs on the 8-wave kernel above.  DUO_PREFILL_W64=0 (or debug
    // flag bit 7) keeps the 8-wave kernel everywhere (same-box A/B, tests of both kernels).
    static const bool want_w64 = [] { const char *e = getenv("DUO_PREFILL_W64"); return !e || atoi(e) != 0; }();
    {
        bool w64_ok = want_w64 && tr && !(g_debug_flags & 128u);
        // (with the generated bulk schedule it wins on every launch shape, first chunks and streaming-only launches
        // included: +9 ... +14 %, profiles/r2_prefill_w64.md; debug bit 8 = never split the key range, so tests reach it
        // on short launches too)
        if (w64_ok) {
            static std::atomic<bool> w64_attr[64][2];
            const void *wfn = F16 ? (const void *)duo_prefill_w64_f16_kernel : (const void *)duo_prefill_w64_kernel;
            if (dev >= 64 || !w64_attr[dev][F16].load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute(wfn,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
                if (e != hipSuccess) return (int)e;
                if (dev < 64) w64_attr[dev][F16].store(true, std::memory_order_release);
            }
            // XCD-aware order of the retrieval class (unsplit launches; DUO_PREFILL_XMAP=0 / debug bit 10: plain order)
            static const bool want_xmap = [] { const char *e = getenv("DUO_PREFILL_XMAP"); return !e || atoi(e) != 0; }();
            if (want_xmap && !(g_debug_flags & 1024u) && P.ksplit == 1 && long_wgs > 0) {
                const int row_items = P.cls[0].n_kv_heads * group;
                int rows = 1;
                while ((rows * row_items) % 8 != 0) rows *= 2;       // 1, 2, 4 or 8 rows: the first multiple of 8 workgroups
                const int periods = (P.n_qtiles + rows - 1) / rows;
                P.xmap_rows = rows;
                P.xmap_q = rows * row_items / 8;
                nblk += periods * rows * row_items - P.nblk_full;     // the padded last period
                P.nblk_full = periods * rows * row_items;
            }
            if constexpr (F16) hipLaunchKernelGGL(duo_prefill_w64_f16_kernel, dim3(nblk, n_batch), dim3(256), LDS_BYTES, st, P);
            else hipLaunchKernelGGL(duo_prefill_w64_kernel, dim3(nblk, n_batch), dim3(256), LDS_BYTES, st, P);
            DUO_HIP_CHECK_LAUNCH();
            if (P.ksplit > 1) {
                hipLaunchKernelGGL((duo_prefill_merge_kernel<F16>), dim3(long_wgs, n_batch), dim3(256), 0, st, P);
                DUO_HIP_CHECK_LAUNCH();
            }
            return 0;
        }
    }
    if (tr) hipLaunchKernelGGL((duo_prefill_kernel<true, F16>), dim3(nblk, n_batch), dim3(512), LDS_BYTES, st, P);
    else hipLaunchKernelGGL((duo_prefill_kernel<false, F16>), dim3(nblk, n_batch), dim3(512), LDS_BYTES, st, P);
    DUO_HIP_CHECK_LAUNCH();
    if (P.ksplit > 1) {
        hipLaunchKernelGGL((duo_prefill_merge_kernel<F16>), dim3(long_wgs, n_batch), dim3(256), 0, st, P);
        DUO_HIP_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int duo_attn_prefill_bf16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                     void *out, int64_t out_token_stride, int64_t out_head_stride,
                                     int32_t n_tokens, int32_t group, const duo_head_class *full,
                                     const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                     void *stream) {
    return prefill_impl<false>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                               group, full, stream_cls, scale, head_dim, nullptr, 0, stream);
}

// Same, with a caller-owned workspace (duo_attn_prefill_workspace_bytes()) that lets the launcher split the
// retrieval class over key ranges when its workgroups would not fill the chip.
extern "C" int duo_attn_prefill_ws_bf16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                        void *out, int64_t out_token_stride, int64_t out_head_stride,
                                        int32_t n_tokens, int32_t group, const duo_head_class *full,
                                        const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                        void *workspace, int64_t workspace_bytes, void *stream) {
    return prefill_impl<false>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                               group, full, stream_cls, scale, head_dim, workspace, workspace_bytes, stream);
}

// fp16 twin (q, K, V, out all fp16): the attention of the INT4 path's chunked prefill over dequantised pools
// (demo/w8a8kv4_llama.py:226-274) and of fp16 models.
// Batched forms (n_batch rows of equal length — the reference's pools and forward carry a batch dimension,
// static_kv_cache.py:60-99, and flash_attn_func batches natively): the batch row is grid.y of the same launches.
extern "C" int duo_attn_prefill_batched_bf16(const void *q, int64_t q_batch_stride, int64_t q_token_stride,
                                             int64_t q_head_stride, void *out, int64_t out_batch_stride,
                                             int64_t out_token_stride, int64_t out_head_stride, int32_t n_batch,
                                             int32_t n_tokens, int32_t group, const duo_head_class *full,
                                             const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                             void *workspace, int64_t workspace_bytes, void *stream) {
    return prefill_impl<false>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                               group, full, stream_cls, scale, head_dim, workspace, workspace_bytes, stream, n_batch,
                               q_batch_stride, out_batch_stride);
}
extern "C" int duo_attn_prefill_batched_f16(const void *q, int64_t q_batch_stride, int64_t q_token_stride,
                                            int64_t q_head_stride, void *out, int64_t out_batch_stride,
                                            int64_t out_token_stride, int64_t out_head_stride, int32_t n_batch,
                                            int32_t n_tokens, int32_t group, const duo_head_class *full,
                                            const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                            void *workspace, int64_t workspace_bytes, void *stream) {
    return prefill_impl<true>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                              group, full, stream_cls, scale, head_dim, workspace, workspace_bytes, stream, n_batch,
                              q_batch_stride, out_batch_stride);
}

extern "C" int duo_attn_prefill_f16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                    void *out, int64_t out_token_stride, int64_t out_head_stride,
                                    int32_t n_tokens, int32_t group, const duo_head_class *full,
                                    const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                    void *stream) {
    return prefill_impl<true>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                              group, full, stream_cls, scale, head_dim, nullptr, 0, stream);
}

extern "C" int duo_attn_prefill_ws_f16(const void *q, int64_t q_token_stride, int64_t q_head_stride,
                                       void *out, int64_t out_token_stride, int64_t out_head_stride,
                                       int32_t n_tokens, int32_t group, const duo_head_class *full,
                                       const duo_head_class *stream_cls, float scale, int32_t head_dim,
                                       void *workspace, int64_t workspace_bytes, void *stream) {
    return prefill_impl<true>(q, q_token_stride, q_head_stride, out, out_token_stride, out_head_stride, n_tokens,
                              group, full, stream_cls, scale, head_dim, workspace, workspace_bytes, stream);
}

#ifdef W64_TIMING      /* measurement builds only (tools/debug): per-phase cycle sums of the last w64 launch */
extern "C" int duo_debug_w64_timing(uint32_t *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(w64_timing), 8 * sizeof(uint32_t));
}
#endif
