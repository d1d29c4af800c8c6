// duo_prefill_w64.h — the 4-wave prefill kernel: one wave per SIMD, 64 query rows per wave.
// Included by duo_prefill.hip (same translation unit as the 8-wave kernel and the launcher).
//
// Same mathematics, LDS layouts, LDS-DMA ring and tile walk as duo_prefill_kernel; what changes is
// the work decomposition inside the workgroup (256 query rows of one q head):
//   * 4 waves x 64 rows: each wave owns TWO 32-row blocks (A, B).  Every K and V^T fragment read
//     from LDS is used by both blocks -> half the LDS traffic per FLOP, half the barriers per FLOP;
//   * one wave per SIMD with the whole 512-entry register file: the two blocks are independent
//     instruction streams of the same wave, so the MFMAs of one block run under the softmax VALU
//     work of the other (an in-order wave keeps issuing VALU while its MFMA executes):
//         P1  S_A = K.Q_A^T                        16 MFMA
//         P2  S_B = K.Q_B^T                        16 MFMA   under  rowmax(A), rescale?(A), exp(A)
//         P3  O_A += V^T.P_A^T                     16 MFMA   under  rowmax(B), rescale?(B), exp(B)
//         P4  O_B += V^T.P_B^T                     16 MFMA
//     the (rare) deferred-rescale branch of a block sits after the 4th MFMA of the phase it hides
//     under, which balances the VALU on both sides of it against the MFMAs on both sides;
//   * the 16 V^T fragments of a tile are register-resident for P3/P4 (K fragments are re-read per
//     block: with Q_A, Q_B, S_A, S_B live there is no room for them in the 256 architected VGPRs).
#pragma once
#include "duo_prefill_common.h"

namespace {

// O += A.B with the accumulator PINNED in the AGPR half of the register file ("+a").  With the builtin,
// hipcc (ROCm 7.2) keeps the long-lived O tiles in architected VGPRs and pushes Q / S / loop invariants
// into AGPRs instead, then shuttles them back with ~430 v_accvgpr moves per tile.  An asm MFMA is
// opaque to its hazard recogniser: the accumulate chain needs no wait states, a VALU-written B operand
// needs two (`nop_first`), and compiler code reading O after the last MFMA needs the full MFMA
// latency (duo_mfma_drain below).
template <bool NOP_FIRST>
__device__ __forceinline__ void mfma_acc_agpr(f32x16 &acc, const bf16x8 &a, const bf16x8 &b) {
    if constexpr (NOP_FIRST)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// S = A.B (+ S) with the accumulator pinned in architected VGPRs ("+v"/"=v"): the scores are consumed by
// VALU code, an AGPR home would cost one v_accvgpr_read per element.  First k-step: C = 0 (inline constant).
__device__ __forceinline__ void mfma_first_vgpr(f32x16 &acc, const bf16x8 &a, const bf16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_acc_vgpr(f32x16 &acc, const bf16x8 &a, const bf16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// the last MFMA of a score tile: pad the full MFMA latency so that compiler VALU code may read it
__device__ __forceinline__ void duo_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

constexpr int W64_NW = 4;
constexpr int W64_DMA_PER_TILE = 2 * 16 / W64_NW;   // 8 global_load_lds per wave per tile

template <bool USE_TR>
__global__ __launch_bounds__(256, 1) void duo_prefill_w64_kernel(const PrefillParams P) {
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
    const DuoClassDev C = duo_select(P.cls[0], P.cls[1], ci != 0);
    const int nq_c = C.n_kv_heads * P.group;
    const int tile = P.n_qtiles - 1 - b / nq_c;   // heaviest (latest) tiles first
    const int p = b % nq_c;
    const int kvh = p % C.n_kv_heads;             // group mates sit 8 blocks apart -> same XCD
    const int g = p / C.n_kv_heads;
    const int qh = C.q_head_offset + kvh * P.group + g;

    const int S = P.S;
    const int q0 = tile * QBLK;
    const int wq0 = q0 + wave * 64;               // first query row of this wave
    const int my_q[2] = {wq0 + l31, wq0 + 32 + l31};

    // ---- Q fragments of both row blocks (B operands of the swapped QK^T) -------
    bf16x8 qf[2][8];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const bf16_t *qp = P.q + (int64_t)min(my_q[x], S - 1) * P.q_ts + (int64_t)qh * P.q_hs + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[x][kk] = *reinterpret_cast<const bf16x8 *>(qp + kk * 16);
    }

    const int lenA = C.a.len;
    const int nA = (lenA + KVBLK - 1) / KVBLK;
    const int last_q = min(q0 + QBLK - 1, S - 1);
    const int nB = last_q / KVBLK + 1;
    const int nT = nA + nB;
    // tiles this wave computes: all of segment A, and the causal tiles up to its last row
    const int nTw = nA + min(nB, (wq0 + 63) / KVBLK + 1);

    f32x16 o[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[x][i][r] = 0.f;
    float mrow[2] = {-INFINITY, -INFINITY};
    float lsum[2] = {0.f, 0.f};
    const float c = P.scale_log2e;

    // ---- loop invariants -----------------------------------------------------------
    const uint32_t smem_lds = lds_addr(smem);
    uint32_t koff[8];    // K fragment of k-step kk, key block 0 (block 1: +8192), ring slot 0
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) koff[kk] = smem_lds + k_lds_off(l31, 2 * kk + hi);
    // V^T fragment base: key quad hi, dim block (l31>>4), this lane's 8-byte piece of the 4x16 block
    const uint32_t vaddr = smem_lds + K_TILE_BYTES + hi * 1024 + (l31 >> 4) * 128 + lane15 * 8;
    const DmaLane dmaA = dma_lane(tid, C.a.token_stride);
    const DmaLane dmaB = dma_lane(tid, C.b.token_stride);

    auto issue_dma = [&](int t, int slot_) {
        const TileSrc ts_ = tile_src(C, kvh, t, nA, S);
        const uint32_t dst = smem_lds + slot_ * STAGE_BYTES;
        if (ts_.cnt == KVBLK) stage_dma_full<W64_NW>(ts_, t < nA ? dmaA : dmaB, dst, tid);
        else stage_dma_tail<W64_NW>(ts_, dst, tid);
    };

    typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // mask + row max of block x's score tile
    auto row_max = [&](f32x16 (&sx)[2], int x, bool inB, int key0, int cnt) -> float {
        const bool need_mask = inB ? (key0 + KVBLK - 1 > wq0 + 32 * x) : (cnt < KVBLK);
        if (need_mask) {
            const int lim = inB ? min(my_q[x] - key0, cnt - 1) : cnt - 1;   // last visible key (tile-local)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = bb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (kl > lim) sx[bb][r] = -INFINITY;
                }
        }
        float t0 = fmaxf(fmaxf(sx[0][0], sx[0][1]), sx[0][2]);
        float t1 = fmaxf(fmaxf(sx[1][0], sx[1][1]), sx[1][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) {
            t0 = fmaxf(fmaxf(t0, sx[0][r]), sx[0][r + 1]);
            t1 = fmaxf(fmaxf(t1, sx[1][r]), sx[1][r + 1]);
        }
        const float tmax = fmaxf(fmaxf(t0, t1), fmaxf(sx[0][15], sx[1][15]));
        return fmaxf(tmax, __shfl_xor(tmax, 32));   // partner lane holds the other 32 keys
    };

    // deferred rescale of block x (see duo_prefill_kernel): every P.V of the block issued so far has
    // been accumulated when this runs, so O and lsum are all there is at the old scale
    auto decide = [&](int x, float tmax) {
        if (!__all((tmax - mrow[x]) * c <= kDeferLog2)) {
            duo_mfma_drain();   // O is about to be read by compiler-generated code
            const float mnew = fmaxf(mrow[x], tmax);
            const float alpha = fast_exp2((mrow[x] - mnew) * c);
            lsum[x] *= alpha;
            mrow[x] = mnew;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[x][i][r] *= alpha;
        }
    };

    // exponentiation slice i (0..15): two scores -> two P values, one packed word
    auto exp_slice = [&](const f32x16 (&sx)[2], int i, float mc, float &psum, uint32_t (&pk)[16]) {
        const int bb = i >> 3, r0 = 2 * (i & 7);
        const float p0 = fast_exp2(fmaf(sx[bb][r0], c, -mc));
        const float p1 = fast_exp2(fmaf(sx[bb][r0 + 1], c, -mc));
        psum += p0;
        psum += p1;
        pk[i] = cvt_pk_bf16(p0, p1);
    };
    auto pack_pf = [&](const uint32_t (&pk)[16], bf16x8 (&pf)[4]) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            u32x4 w = {pk[4 * st], pk[4 * st + 1], pk[4 * st + 2], pk[4 * st + 3]};
            pf[st] = *reinterpret_cast<bf16x8 *>(&w);
        }
    };

    // ---- prologue: tiles 0 and 1 in flight, wait for tile 0 only ---------------
    issue_dma(0, 0);
    if (nT > 1) {
        issue_dma(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // make hipcc wait for the Q loads here, not inside the loop (see duo_prefill_kernel)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" ::"v"(qf[x][kk]));
    __builtin_amdgcn_sched_barrier(0);

    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int SOFF = SLOT * STAGE_BYTES;
        const bool more2 = t + 2 < nT;
        // ring slot (t+2)%3 == (t-1)%3 was last read in iteration t-1 (left through its barrier)
        if (more2) issue_dma(t + 2, (SLOT + 2) % NSTAGE);

        if (t < nTw) {
            const bool inB = t >= nA;
            const int key0 = inB ? (t - nA) * KVBLK : t * KVBLK;   // first key of the tile in its segment
            const int cnt = inB ? min(KVBLK, S - key0) : min(KVBLK, lenA - key0);
            // block B starts 32 rows later: a causal tile may concern block B only
            const bool skipA = inB && key0 > wq0 + 31;

            f32x16 sa[2], sb[2];
            uint32_t pk[16];
            bf16x8 pfa[4], pfb[4];
            float psum;
            // K fragment i = (k-step i>>1, key block i&1); read just in time for each block (keeping all
            // 16 resident next to Q_A, Q_B, S_A, S_B overflows the 256 architected VGPRs and hipcc
            // then shuttles values through AGPRs with v_accvgpr moves — measured 0.72x)
            auto kfrag = [&](int i) -> bf16x8 {
                return *(lds_frag_t *)(uintptr_t)(koff[i >> 1] + SOFF + (i & 1) * 8192);
            };

            // ---- P1: S_A ----------------------------------------------------------------
            if (!skipA) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i < 2) mfma_first_vgpr(sa[i & 1], kfrag(i), qf[0][i >> 1]);
                    else mfma_acc_vgpr(sa[i & 1], kfrag(i), qf[0][i >> 1]);
                }
            }
            // ---- P2: S_B under softmax(A) -------------------------------------------------
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < 2) mfma_first_vgpr(sb[i & 1], kfrag(i), qf[1][i >> 1]);
                else mfma_acc_vgpr(sb[i & 1], kfrag(i), qf[1][i >> 1]);
            }
            // (S_A's last MFMA was issued >= 4 MFMAs = 128 cycles ago: its result is readable)
            if (!skipA) decide(0, row_max(sa, 0, inB, key0, cnt));
            {
                const float mc = mrow[0] * c;
                psum = 0.f;
#pragma unroll
                for (int i = 4; i < 16; ++i) {
                    mfma_acc_vgpr(sb[i & 1], kfrag(i), qf[1][i >> 1]);
                    if (!skipA) {
                        exp_slice(sa, i - 4, mc, psum, pk);
                        if (i >= 12) exp_slice(sa, i, mc, psum, pk);   // 16 slices over 12 MFMAs
                    }
                }
                if (!skipA) {
                    lsum[0] += psum;
                    pack_pf(pk, pfa);
                }
            }

            // ---- V^T fragments: register-resident for both blocks -------------------------
            bf16x8 vf[16];   // [step][db]
            if constexpr (USE_TR) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t a0 = vaddr + SOFF + (i >> 2) * 4096 + (i & 3) * 256;
                    const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)a0);
                    const s16x4 y0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)(a0 + 2048));
                    vf[i][0] = x0[0]; vf[i][1] = x0[1]; vf[i][2] = x0[2]; vf[i][3] = x0[3];
                    vf[i][4] = y0[0]; vf[i][5] = y0[1]; vf[i][6] = y0[2]; vf[i][7] = y0[3];
                }
            } else {
                const char *vst = smem + SOFF + K_TILE_BYTES;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int blk16 = 2 * (i & 3) + (l31 >> 4);
                    const int b0 = (((4 * (i >> 2) + hi) * 8 + blk16) << 7);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vf[i][j] = *reinterpret_cast<const short *>(vst + b0 + j * 32 + lane15 * 2);
                        vf[i][4 + j] = *reinterpret_cast<const short *>(vst + b0 + 2048 + j * 32 + lane15 * 2);
                    }
                }
            }

            // ---- P3: O_A under softmax(B) ---------------------------------------------------
            if (!skipA) {
#pragma unroll
                mfma_acc_agpr<true>(o[0][0], vf[0], pfa[0]);   // pfa was packed by VALU just above
#pragma unroll
                for (int i = 1; i < 4; ++i) mfma_acc_agpr<false>(o[0][i & 3], vf[i], pfa[i >> 2]);
            }
            // S_B's last MFMA: 4 P.V MFMAs ago, unless block A is skipped on this tile
            if (skipA) duo_mfma_drain();
            decide(1, row_max(sb, 1, inB, key0, cnt));
            {
                const float mc = mrow[1] * c;
                psum = 0.f;
#pragma unroll
                for (int i = 4; i < 16; ++i) {
                    if (!skipA) mfma_acc_agpr<false>(o[0][i & 3], vf[i], pfa[i >> 2]);
                    exp_slice(sb, i - 4, mc, psum, pk);
                    if (i >= 12) exp_slice(sb, i, mc, psum, pk);
                }
                lsum[1] += psum;
                pack_pf(pk, pfb);
            }
            // ---- P4: O_B --------------------------------------------------------------------
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i == 0) mfma_acc_agpr<true>(o[1][0], vf[0], pfb[0]);   // pfb was packed by VALU just above
                else mfma_acc_agpr<false>(o[1][i & 3], vf[i], pfb[i >> 2]);
            }
        }

        // ---- tile t+1 must have landed (tile t+2 may stay in flight), then ONE barrier ----------
        __builtin_amdgcn_sched_barrier(0);
        if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int t = 0; t < nT; t += NSTAGE) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nT) tile_body(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < nT) tile_body(std::integral_constant<int, 2>{}, t + 2);
    }

    // ---- epilogue: O^T / l -> out[q][qh][d] -----------------------------------
    duo_mfma_drain();
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const float l = lsum[x] + __shfl_xor(lsum[x], 32);
        const float inv = 1.f / l;
        if (my_q[x] < S) {
            bf16_t *op = P.out + (int64_t)my_q[x] * P.o_ts + (int64_t)qh * P.o_hs;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = 32 * db + 8 * rq + 4 * hi;   // rows (r&3)+8*(r>>2)+4*hi, r = 4rq..4rq+3
                    u32x2 w;
                    w.x = cvt_pk_bf16(o[x][db][4 * rq + 0] * inv, o[x][db][4 * rq + 1] * inv);
                    w.y = cvt_pk_bf16(o[x][db][4 * rq + 2] * inv, o[x][db][4 * rq + 3] * inv);
                    *reinterpret_cast<u32x2 *>(op + d) = w;
                }
        }
    }
}

}  // namespace
