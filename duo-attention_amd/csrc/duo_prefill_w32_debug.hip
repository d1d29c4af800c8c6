// duo_prefill_w32_debug.hip — the round-1 prefill kernel (8 waves x 32 rows), kept for DEBUGGING AND CROSS-CHECKS ONLY.
//
// No product launch reaches this file: every prefill launch runs duo_prefill_w64_kernel (duo_prefill_w64.h) since round 2.
// What still comes here, through duo_prefill_w32_launch() from the launcher in duo_prefill.hip:
//   * debug bit 0 — the gather (non-transposed-LDS) V layout, a debugging aid for ds_read_b64_tr_b16 layouts;
//   * debug bit 7 / DUO_PREFILL_W64=0 — the same launch on this kernel: the same-box A/B of the two kernels
//     (tools/debug/ab_prefill.sh) and an independent second implementation for the GPU parity tests, which run every
//     prefill case on both.
// Same semantics as the w64 kernel (see duo_prefill.hip): keys = segA (all visible) ++ segB (causal, bottom-right aligned),
// fp32 scores / softmax, P rounded to the element type before P.V.
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
#include <atomic>
#include "duo_prefill_common.h"

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

    // ---- block -> (class, q tile, kv head, q head, key-range piece): prefill_map_block (duo_prefill_common.h) ----
    const PrefillItem I = prefill_map_block(P, blockIdx.x);
    if (I.tile < 0) return;           // padding of the last XCD period
    const int ci = I.ci;
    const int by = blockIdx.y;        // batch row
    DuoClassDev Crow = duo_select(P.cls[0], P.cls[1], ci != 0);
    duo_class_batch_row(Crow, by);
    const DuoClassDev C = Crow;
    const int ks = I.ks;
    const int split = I.split;
    const int part_id = I.part + by * P.nparts;   // index of this workgroup's partial in the workspace
    const int tile = I.tile, kvh = I.kvh, g = I.g;
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

}  // namespace

// attribute + launch of the 8-wave kernel for the launcher in duo_prefill.hip (grid = (nblk, n_batch), 512 threads)
int duo_prefill_w32_launch(const void *params, bool tr, bool f16, int nblk, int n_batch, int dev, hipStream_t st) {
    const PrefillParams &P = *static_cast<const PrefillParams *>(params);
    // hipFuncSetAttribute is cheap but not free: once per (device, kernel instantiation).  The attribute belongs to
    // the function as loaded on ONE device, so a process that drives several GPUs must set it on each; atomics because
    // any host thread may get here.
    static std::atomic<bool> attr_done[64][2][2];   // [device][element type][transpose-read variant]
    const void *fn = f16 ? (tr ? (const void *)duo_prefill_kernel<true, true> : (const void *)duo_prefill_kernel<false, true>)
                         : (tr ? (const void *)duo_prefill_kernel<true, false> : (const void *)duo_prefill_kernel<false, false>);
    if (dev >= 64 || !attr_done[dev][f16][tr].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        if (dev < 64) attr_done[dev][f16][tr].store(true, std::memory_order_release);
    }
    const dim3 grid(nblk, n_batch), block(512);
    if (f16) {
        if (tr) hipLaunchKernelGGL((duo_prefill_kernel<true, true>), grid, block, LDS_BYTES, st, P);
        else hipLaunchKernelGGL((duo_prefill_kernel<false, true>), grid, block, LDS_BYTES, st, P);
    } else {
        if (tr) hipLaunchKernelGGL((duo_prefill_kernel<true, false>), grid, block, LDS_BYTES, st, P);
        else hipLaunchKernelGGL((duo_prefill_kernel<false, false>), grid, block, LDS_BYTES, st, P);
    }
    return (int)hipGetLastError();
}
