// duo_prefill_w64.h — the 4-wave prefill kernel: one wave per SIMD, 64 query rows per wave, the whole
// 512-entry register file per wave.  Included by duo_prefill.hip (same translation unit as the 8-wave kernel
// and the launcher).  bf16, no key-range splits, segment B == the query rows (the launcher falls back to the
// 8-wave kernel otherwise).
//
// Why (profiles/r2_prefill_ablations.md): in the 8-wave kernel every wave reads the whole K and V tile from LDS
// for its 32 query rows — that operand traffic is its largest single cost (18 % of a launch), ahead of the
// softmax and the tile staging.  Here a wave owns TWO 32-row blocks (A, B): every K / V^T fragment read from
// LDS feeds two MFMAs, and per-tile fixed work (barrier, DMA issue, waits) is shared by twice the rows.
//
// One wave per SIMD means nothing else fills the matrix pipe while this wave does softmax arithmetic, so the
// two blocks are software-pipelined against each other inside the wave (an in-order wave keeps issuing VALU
// and LDS instructions while its MFMA executes):
//     P1  S_A = K.Q_A^T        16 MFMA   under  V^T(t) transpose reads + the LDS-DMA of tile t+2
//     P2  S_B = K.Q_B^T        16 MFMA   under  row max / (rare) rescale / exp / pack of block A
//     P3  O_A += V^T.P_A^T     16 MFMA   under  the same for block B
//         -- wait: tile t+1 landed; ONE s_barrier per tile --
//     P4  O_B += V^T.P_B^T     16 MFMA   under  the K(t+1) fragment reads
// K fragments (16 x 4 registers) and V^T fragments (16 x 4) of a tile are register-resident: LDS reads are a
// whole phase ahead of their MFMAs, never on the critical path.
//
// Register ownership.  hipcc cannot be trusted with ~450 live registers (given "a"/"v" constraints it spilled Q and K
// to scratch and reloaded them — with a vmcnt(0) — in front of every MFMA).  So the ACCUMULATOR half of the file is
// owned by this file's asm statements through literal register numbers, and the compiler only manages the
// architected half (scores, P, V^T fragments, softmax temporaries):
//     a[0:63]    O_A  (dim block db: a[16db : 16db+15])          a[64:127]   O_B
//     a[128:159] Q_A  (k-step kk: a[128+4kk : +3])               a[160:191]  Q_B
//     a[192:255] K    (fragment i = (k-step i>>1, key block i&1): a[192+4i : +3])
// The compiler must never touch an AGPR: audit every build for `.vgpr_spill_count 0`, scratch 0 and no
// v_accvgpr_* outside ;;#ASMSTART/;;#ASMEND (tools/debug/audit_w64.sh).  An asm MFMA is opaque to the compiler's
// hazard recogniser; the rules kept by hand are noted at each site.
#pragma once
#include "duo_prefill_common.h"

namespace {

constexpr int W64_NW = 4;
constexpr int W64_DMA_PER_TILE = 2 * 16 / W64_NW;   // 8 global_load_lds per wave per tile

// S = K.Q^T: block x (0 = A, 1 = B), K fragment i, first k-step (C = 0) / accumulate.  D: architected VGPRs.
#define W64_QK0(sacc, i, x)                                                                                   \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(sacc)                        \
                 : "n"(192 + 4 * (i)), "n"(195 + 4 * (i)), "n"(128 + 32 * (x) + 4 * ((i) >> 1)),              \
                   "n"(131 + 32 * (x) + 4 * ((i) >> 1)))
#define W64_QKA(sacc, i, x)                                                                                   \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(sacc)                       \
                 : "n"(192 + 4 * (i)), "n"(195 + 4 * (i)), "n"(128 + 32 * (x) + 4 * ((i) >> 1)),              \
                   "n"(131 + 32 * (x) + 4 * ((i) >> 1)))
#define W64_QK(sacc, i, x)                     \
    do {                                       \
        if constexpr ((i) < 2) W64_QK0(sacc, i, x); \
        else W64_QKA(sacc, i, x);              \
    } while (0)
// O_x[db = i&3] += V^T fragment . P^T fragment.  NOPS: "s_nop 1\n\t" when the P fragment was written by VALU just before.
#define W64_PV(x, i, vfrag, pfrag, NOPS)                                                                      \
    asm volatile(NOPS "v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]"                               \
                 :: "v"(vfrag), "v"(pfrag), "n"(64 * (x) + 16 * ((i) & 3)), "n"(64 * (x) + 16 * ((i) & 3) + 15))
// K fragment i of the tile in the ring slot addressed by (addr, off) -> a[192+4i : +3]
#define W64_KRD(i, addr, off)                                                                                 \
    asm volatile("ds_read_b128 a[%c2:%c3], %0 offset:%c1" :: "v"(addr), "n"(off), "n"(192 + 4 * (i)),         \
                 "n"(195 + 4 * (i)) : "memory")
// One softmax "slice" = two scores -> two P values (exp2 domain), row-sum update, one packed bf16x2 word.  Always
// fused into the asm statement of the MFMA it rides under: as separate statements hipcc pads an s_nop behind every
// one, and as plain C++ it sinks the arithmetic out of the MFMA interleave altogether.
// (gfx950: a VALU may not read a transcendental's result in the very next slot: exp0, exp1, add0, add1 keeps one
// independent instruction behind each v_exp.)
// The slice is a two-stage pipeline across consecutive statements, so that no instruction waits on the one just
// before it (v_exp has a long latency, and one wave per SIMD has nothing else to issue meanwhile):
//   stage A(k):   e0 = exp2(s0 c - m c), e1 = exp2(s1 c - m c)        (fma, fma, exp, exp)  -> temporaries pair k&1
//   stage B(k-1): psum += e0' + e1';  pk = bf16x2(e0', e1')           (add, add, cvt)       <- pair (k-1)&1
// W64_SL_AB = A(k) interleaved with B(k-1);  W64_SL_A = A only (first slice);  W64_SL_B = B only (after the last).
// (the packed word may be an operand of the very next statement's MFMA: v_cvt_pk sits three instructions before the
// end — a VALU-written MFMA operand needs two wait states)
#define W64_SL_AB_TXT                                  \
    "v_fma_f32 %[a0], %[s0], %[c], %[nm]\n\t"          \
    "v_fma_f32 %[a1], %[s1], %[c], %[nm]\n\t"          \
    "v_add_f32 %[ps], %[ps], %[b0]\n\t"                \
    "v_cvt_pk_bf16_f32 %[pk], %[b0], %[b1]\n\t"        \
    "v_exp_f32 %[a0], %[a0]\n\t"                       \
    "v_add_f32 %[ps], %[ps], %[b1]\n\t"                \
    "v_exp_f32 %[a1], %[a1]"
#define W64_SL_A_TXT                                   \
    "v_fma_f32 %[a0], %[s0], %[c], %[nm]\n\t"          \
    "v_fma_f32 %[a1], %[s1], %[c], %[nm]\n\t"          \
    "v_exp_f32 %[a0], %[a0]\n\t"                       \
    "v_exp_f32 %[a1], %[a1]"
#define W64_SL_B_TXT                                   \
    "v_cvt_pk_bf16_f32 %[pk], %[b0], %[b1]\n\t"        \
    "v_add_f32 %[ps], %[ps], %[b0]\n\t"                \
    "v_add_f32 %[ps], %[ps], %[b1]"
// operand lists: sl = the slice whose stage A runs (scores sx[..]); its stage-B partner is slice sl-1 (word pk_[sl-1])
#define W64_A_OUT(e_, sl) [a0] "=&v"(e_[(sl) & 1][0]), [a1] "=&v"(e_[(sl) & 1][1])
#define W64_A_IN(sx, sl, nm_)                                                                                 \
    [s0] "v"(sx[(sl) >> 3][2 * ((sl) & 7)]), [s1] "v"(sx[(sl) >> 3][2 * ((sl) & 7) + 1]), [c] "s"(c), [nm] "v"(nm_)
#define W64_B_OUT(ps_, pk_, slb) [ps] "+v"(ps_), [pk] "=&v"(pk_[slb])
#define W64_B_IN(e_, slb) [b0] "v"(e_[(slb) & 1][0]), [b1] "v"(e_[(slb) & 1][1])
// MFMA texts
#define W64_TXT_QKB "v_mfma_f32_32x32x16_bf16 %[acc], a[%c[k0]:%c[k1]], a[%c[q0]:%c[q1]], %[acc]\n\t"
#define W64_TXT_PV "v_mfma_f32_32x32x16_bf16 a[%c[o0]:%c[o1]], %[vf], %[pf], a[%c[o0]:%c[o1]]\n\t"
#define W64_TXT_KRD "ds_read_b128 a[%c[k0]:%c[k1]], %[ka] offset:%c[ko]\n\t"
#define W64_TXT_KRD2 "ds_read_b128 a[%c[k2]:%c[k3]], %[kb] offset:%c[kp]\n\t"
#define W64_OPS_QKB(i) [k0] "n"(192 + 4 * (i)), [k1] "n"(195 + 4 * (i)), [q0] "n"(160 + 4 * ((i) >> 1)), [q1] "n"(163 + 4 * ((i) >> 1))
#define W64_OPS_PV(x, i, vfrag, pfrag) [vf] "v"(vfrag), [pf] "v"(pfrag), [o0] "n"(64 * (x) + 16 * ((i) & 3)), [o1] "n"(64 * (x) + 16 * ((i) & 3) + 15)
#define W64_OPS_KRD(j, addr, off) [ka] "v"(addr), [ko] "n"(off), [k0] "n"(192 + 4 * (j)), [k1] "n"(195 + 4 * (j))
#define W64_OPS_KRD2(j, addr, off) [kb] "v"(addr), [kp] "n"(off), [k2] "n"(192 + 4 * (j)), [k3] "n"(195 + 4 * (j))
// S_B MFMA i (accumulate form) + stage A of A-slice sl [+ stage B of A-slice sl-1]
#define W64_QKB_A(i, sl)                                                                                      \
    asm volatile(W64_TXT_QKB W64_SL_A_TXT : [acc] "+v"(sb[(i) & 1]), W64_A_OUT(eA, sl)                         \
                 : W64_A_IN(sa, sl, nmA), W64_OPS_QKB(i))
#define W64_QKB_AB(i, sl)                                                                                     \
    asm volatile(W64_TXT_QKB W64_SL_AB_TXT                                                                    \
                 : [acc] "+v"(sb[(i) & 1]), W64_A_OUT(eA, sl), W64_B_OUT(psumA, pkA, (sl) - 1)                 \
                 : W64_A_IN(sa, sl, nmA), W64_B_IN(eA, (sl) - 1), W64_OPS_QKB(i))
// O_x MFMA i + stage A of slice sl of block y [+ stage B of slice sl-1]   (e_/ps_/pk_/nm_/sx of block y)
#define W64_PV_A(x, i, vfrag, pfrag, sx, sl, e_, nm_)                                                         \
    asm volatile(W64_TXT_PV W64_SL_A_TXT : W64_A_OUT(e_, sl) : W64_A_IN(sx, sl, nm_), W64_OPS_PV(x, i, vfrag, pfrag))
#define W64_PV_AB(x, i, vfrag, pfrag, sx, sl, e_, ps_, pk_, nm_)                                              \
    asm volatile(W64_TXT_PV W64_SL_AB_TXT : W64_A_OUT(e_, sl), W64_B_OUT(ps_, pk_, (sl) - 1)                   \
                 : W64_A_IN(sx, sl, nm_), W64_B_IN(e_, (sl) - 1), W64_OPS_PV(x, i, vfrag, pfrag))
#define W64_PV_B(x, i, vfrag, pfrag, slb, e_, ps_, pk_)                                                       \
    asm volatile(W64_TXT_PV W64_SL_B_TXT : W64_B_OUT(ps_, pk_, slb) : W64_B_IN(e_, slb), W64_OPS_PV(x, i, vfrag, pfrag))
// O_B MFMA i + TWO K(t+1) fragment reads (fragments 2i, 2i+1: all 16 requested under the first 8 MFMAs, so the
// last one has half a phase to land before P1 waits for it) [+ B-slice stages]
// (W64_KA(j) / W64_KOF(j): LDS address register and immediate offset of K fragment j in the NEXT tile's ring slot —
// defined at the use site)
#define W64_PVB_K2_AB(i, vfrag, pfrag, sl)                                                                    \
    asm volatile(W64_TXT_PV W64_TXT_KRD W64_TXT_KRD2 W64_SL_AB_TXT                                            \
                 : W64_A_OUT(eB, sl), W64_B_OUT(psumB, pkB, (sl) - 1)                                          \
                 : W64_A_IN(sb, sl, nmB), W64_B_IN(eB, (sl) - 1), W64_OPS_PV(1, i, vfrag, pfrag),              \
                   W64_OPS_KRD(2 * (i), W64_KA(2 * (i)), W64_KOF(2 * (i))),                                    \
                   W64_OPS_KRD2(2 * (i) + 1, W64_KA(2 * (i) + 1), W64_KOF(2 * (i) + 1)) : "memory")
#define W64_PVB_K2(i, vfrag, pfrag)                                                                           \
    asm volatile(W64_TXT_PV W64_TXT_KRD W64_TXT_KRD2 "s_nop 0"                                                \
                 :: W64_OPS_PV(1, i, vfrag, pfrag), W64_OPS_KRD(2 * (i), W64_KA(2 * (i)), W64_KOF(2 * (i))),   \
                    W64_OPS_KRD2(2 * (i) + 1, W64_KA(2 * (i) + 1), W64_KOF(2 * (i) + 1)) : "memory")
// compiler-generated code is about to read (or an MFMA to re-read) registers an asm MFMA may still be writing
__device__ __forceinline__ void w64_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// three-input max in ONE instruction (clang puts canonicalising v_max x,x in front of fmaxf on asm-MFMA outputs)
__device__ __forceinline__ float w64_max3(float a, float b, float c3) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c3));   // volatile: stays in its MFMA gap
    return r;
}
// K piece + V piece of one LDS-DMA step in ONE statement: M0 saved / restored once, set by s_add with an immediate
// (lds_k / lds_v: wave-uniform LDS byte addresses; kofs / vofs: per-lane source byte offsets; kb / vb: tile bases)
__device__ __forceinline__ void w64_dma_pair(const bf16_t *kb, const bf16_t *vb, uint32_t kofs, uint32_t vofs,
                                             uint32_t lds_k, uint32_t lds_v) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(kofs), "v"(vofs), "s"(kb), "s"(vb), "s"(lds_k), "s"(lds_v)
        : "memory");
}

template <int N, int END>
__device__ __forceinline__ void w64_acc_zero() {
    if constexpr (N < END) {
        asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(N));
        w64_acc_zero<N + 1, END>();
    }
}
template <int N, int END>
__device__ __forceinline__ void w64_acc_scale(float alpha) {      // a[N..END) *= alpha  (rare: deferred rescale)
    if constexpr (N < END) {
        float tmp;
        asm volatile("v_accvgpr_read_b32 %0, a%c2\n\ts_nop 1\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\t"
                     "v_accvgpr_write_b32 a%c2, %0\n\ts_nop 1" : "=&v"(tmp) : "v"(alpha), "n"(N));
        w64_acc_scale<N + 1, END>(alpha);
    }
}
template <int N, int END>
__device__ __forceinline__ void w64_acc_read(float *dst) {        // dst[k] = a[N + k]
    if constexpr (N < END) {
        asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(dst[0]) : "n"(N));
        w64_acc_read<N + 1, END>(dst + 1);
    }
}
template <int N, int END>
__device__ __forceinline__ void w64_q_write(const uint32_t *src) {   // a[N + k] = src[k]
    if constexpr (N < END) {
        asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(src[0]), "n"(N));
        w64_q_write<N + 1, END>(src + 1);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void duo_prefill_w64_kernel(const PrefillParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // every AGPR belongs to the asm statements of this kernel: the clobber list makes the kernel descriptor allocate them
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int lane15 = lane & 15;

    // ---- block -> (class, q tile, kv head, q head): same order as the 8-wave kernel ----------------
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

    // ---- Q fragments of both row blocks (B operands of the swapped QK^T) -> a[128:191] -------------
    {
        uint32_t qw[64];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16_t *qp = P.q + (int64_t)min(my_q[x], S - 1) * P.q_ts + (int64_t)qh * P.q_hs + hi * 8;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const u32x4 w = *reinterpret_cast<const u32x4 *>(qp + kk * 16);
                qw[32 * x + 4 * kk + 0] = w.x;
                qw[32 * x + 4 * kk + 1] = w.y;
                qw[32 * x + 4 * kk + 2] = w.z;
                qw[32 * x + 4 * kk + 3] = w.w;
            }
        }
        w64_q_write<128, 192>(qw);
    }
    w64_acc_zero<0, 128>();

    const int lenA = C.a.len;
    const int nA = (lenA + KVBLK - 1) / KVBLK;
    const int last_q = min(q0 + QBLK - 1, S - 1);
    const int nB = last_q / KVBLK + 1;
    const int nT = nA + nB;
    // tiles this wave computes: all of segment A, and the causal tiles up to its last row
    const int nTw = nA + min(nB, (wq0 + 63) / KVBLK + 1);

    float mrow[2] = {-INFINITY, -INFINITY};
    float lsum[2] = {0.f, 0.f};
    const float c = P.scale_log2e;

    // ---- loop invariants ---------------------------------------------------------------------------
    const uint32_t smem_lds = lds_addr(smem);
    uint32_t koff[8];    // K fragment of k-step kk, key block 0 (block 1: +8192), ring slot 0
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) koff[kk] = smem_lds + k_lds_off(l31, 2 * kk + hi);
    // V^T fragment base: key quad hi, dim block (l31>>4), this lane's 8-byte piece of the 4x16 block
    const uint32_t vaddr = smem_lds + K_TILE_BYTES + hi * 1024 + (l31 >> 4) * 128 + lane15 * 8;
    const DmaLane dmaA = dma_lane(tid, C.a.token_stride);
    const DmaLane dmaB = dma_lane(tid, C.b.token_stride);

    // LDS-DMA of tile t into ring slot `slot_`: 8 pieces per wave (K piece j, V piece j for j = 0..3); a full
    // tile can be issued piecewise (interleaved with MFMAs), a tail tile goes out in one go
    struct DmaPlan {
        const bf16_t *kb, *vb;
        int64_t step;
        uint32_t kofs, vofs, base;
        bool full;
    };
    auto dma_plan = [&](int t, int slot_) -> DmaPlan {
        const TileSrc ts_ = tile_src(C, kvh, t, nA, S);
        DmaPlan d;
        d.full = ts_.cnt == KVBLK;
        d.base = __builtin_amdgcn_readfirstlane(smem_lds + slot_ * STAGE_BYTES);
        d.kb = d.vb = nullptr;
        d.step = 0;
        d.kofs = d.vofs = 0;
        if (!d.full) {
            stage_dma_tail<W64_NW>(ts_, d.base, tid);
            return d;
        }
        const DmaLane &L = t < nA ? dmaA : dmaB;
        d.kb = ts_.k + (int64_t)ts_.row0 * ts_.ts;
        d.vb = ts_.v + (int64_t)ts_.row0 * ts_.ts;
        d.step = 4 * W64_NW * ts_.ts;
        d.kofs = L.kofs;
        d.vofs = L.vofs;
        d.base += wave * 1024;
        return d;
    };
    auto dma_piece = [&](const DmaPlan &d, int j) {   // j = 0..3: K piece j and V piece j of this wave
        if (!d.full) return;
        glds16_s(d.kb + j * d.step, d.kofs, d.base + j * W64_NW * 1024);
        glds16_s(d.vb + j * d.step, d.vofs, d.base + K_TILE_BYTES + j * W64_NW * 1024);
    };

    // max over one 32-key half of a score tile (8 x v_max3), and the combination of both halves across the two lanes
    // of a query row
    auto half_max = [&](const f32x16 &h) -> float {
        float t0 = w64_max3(h[0], h[1], h[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) t0 = w64_max3(t0, h[r], h[r + 1]);
        return w64_max3(t0, h[15], h[15]);
    };
    // a quarter of a score tile (8 registers) folded into a running maximum: 4 x v_max3
    auto quarter_max = [&](float run, const f32x16 &h, int r0) -> float {
        run = w64_max3(run, h[r0], h[r0 + 1]);
        run = w64_max3(run, h[r0 + 2], h[r0 + 3]);
        run = w64_max3(run, h[r0 + 4], h[r0 + 5]);
        return w64_max3(run, h[r0 + 6], h[r0 + 7]);
    };
    auto row_max_finish = [&](float t0, float t1) -> float {
        const float tmax = w64_max3(t0, t1, t1);
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        return w64_max3(__uint_as_float(sw.x), __uint_as_float(sw.y), __uint_as_float(sw.y));   // partner lane: the other 32 keys
    };
    auto mask_tile = [&](f32x16 (&sx)[2], int x, bool inB, int key0, int cnt) {
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
    };
    // mask + row max of block x's score tile
    auto row_max = [&](f32x16 (&sx)[2], int x, bool inB, int key0, int cnt, bool bulk = false) -> float {
        const bool need_mask = !bulk && (inB ? (key0 + KVBLK - 1 > wq0 + 32 * x) : (cnt < KVBLK));
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
        return row_max_finish(half_max(sx[0]), half_max(sx[1]));
    };

    // deferred rescale of block x (see duo_prefill_kernel).  Block x's last P.V MFMA was issued at least a whole
    // phase (16 MFMAs) ago, so O_x is quiescent; the drains cover the asm-MFMA <-> accvgpr hazards (rare path).
    auto decide = [&](auto xc, float tmax) {
        constexpr int x = decltype(xc)::value;
        if (!__all((tmax - mrow[x]) * c <= kDeferLog2)) {
            w64_mfma_drain();
            const float mnew = fmaxf(mrow[x], tmax);
            const float alpha = mnew == -INFINITY ? 1.f : fast_exp2((mrow[x] - mnew) * c);
            lsum[x] *= alpha;
            mrow[x] = mnew;
            w64_acc_scale<64 * x, 64 * x + 64>(alpha);
            w64_mfma_drain();   // v_accvgpr_write -> MFMA SrcC
        }
    };

    // exponentiation slice i (0..15): two scores -> two P values, one packed word.  ONE asm statement: left to
    // the compiler, these instructions are sunk out of the MFMA interleave (it moved whole runs of fma/exp/add behind
    // the last MFMA of a phase), and with one wave per SIMD nothing else covers the matrix pipe meanwhile.
    // (gfx950: a VALU may not read a transcendental's result in the very next slot — the two adds are ordered so
    // that each v_exp has an independent instruction behind it.)
    auto exp_slice = [&](const f32x16 (&sx)[2], int i, float mc, float &psum, uint32_t (&pk)[16]) {
        const int bb = i >> 3, r0 = 2 * (i & 7);
        float p0, p1;
        asm volatile(
            "v_fma_f32 %0, %4, %6, %7\n\t"
            "v_fma_f32 %1, %5, %6, %7\n\t"
            "v_exp_f32 %0, %0\n\t"
            "v_exp_f32 %1, %1\n\t"
            "v_add_f32 %2, %2, %0\n\t"
            "v_add_f32 %2, %2, %1\n\t"
            "v_cvt_pk_bf16_f32 %3, %0, %1"
            : "=&v"(p0), "=&v"(p1), "+v"(psum), "=&v"(pk[i])
            : "v"(sx[bb][r0]), "v"(sx[bb][r0 + 1]), "s"(c), "v"(-mc));
    };

    // ---- prologue: tiles 0 and 1 in flight, tile 0 landed, K(0) fragments requested ------------------
    {
        const DmaPlan d0 = dma_plan(0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_piece(d0, j);
        if (nT > 1) {
            const DmaPlan d1 = dma_plan(1, 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_piece(d1, j);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#define W64_K0(i) W64_KRD(i, koff[(i) >> 1], ((i) & 1) * 8192)
    W64_K0(0); W64_K0(1); W64_K0(2); W64_K0(3); W64_K0(4); W64_K0(5); W64_K0(6); W64_K0(7);
    W64_K0(8); W64_K0(9); W64_K0(10); W64_K0(11); W64_K0(12); W64_K0(13); W64_K0(14); W64_K0(15);
#undef W64_K0
    __builtin_amdgcn_sched_barrier(0);

    // ---- bulk tiles: full 64-row tiles of segment A whose tile t+2 is one too — no masks, no skipping, no tail
    //      handling, LDS-DMA from running scalar tile bases + loop-invariant per-lane offsets (one asm statement
    //      per K/V piece pair).  With one wave per SIMD every instruction is a serial issue slot next to the MFMAs.
    const int nBulk = (max(0, lenA / KVBLK - 2) / NSTAGE) * NSTAGE;
    const int64_t tileA_elems = (int64_t)KVBLK * C.a.token_stride;
    const int64_t tileB_elems = (int64_t)KVBLK * C.b.token_stride;
    const bf16_t *run_k = C.a.k + (int64_t)kvh * C.a.head_stride + 2 * tileA_elems;    // tile t+2, t = 0
    const bf16_t *run_v = C.a.v + (int64_t)kvh * C.a.head_stride + 2 * tileA_elems;
    uint32_t bk_ofs[4], bv_ofs[4];      // segment A: per-lane source byte offsets of piece pair j
    uint32_t ck_ofs[4], cv_ofs[4];      // segment B
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bk_ofs[j] = dmaA.kofs + (uint32_t)(j * 4 * W64_NW * C.a.token_stride * 2);
        bv_ofs[j] = dmaA.vofs + (uint32_t)(j * 4 * W64_NW * C.a.token_stride * 2);
        ck_ofs[j] = dmaB.kofs + (uint32_t)(j * 4 * W64_NW * C.b.token_stride * 2);
        cv_ofs[j] = dmaB.vofs + (uint32_t)(j * 4 * W64_NW * C.b.token_stride * 2);
    }
    const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(smem_lds + wave * 1024);
    // The same for the chunk's own tiles (segment B) that EVERY row of the workgroup sees in full: tiles whose last
    // key is not after the workgroup's first query row.  Bulk runs start on a ring-slot-0 tile and cover whole
    // ring turns; tile t+2 of every bulk tile must itself be a full tile of segment B.
    //   tB0: first t >= max(nA, end of the general tiles after bulk A) with t % 3 == 0
    //   visible in full:  (t - nA) * 64 + 63 <= q0;   tile t+2 full and existing: (t + 3 - nA) * 64 <= S
    const int tB0 = ((max(nA, nBulk) + NSTAGE - 1) / NSTAGE) * NSTAGE;
    const int lastB_vis = nA + (q0 + 1) / KVBLK;                   // exclusive: tiles [nA, lastB_vis) are fully visible
    const int lastB_dma = nA + S / KVBLK - 2;                      // exclusive: tile t+2 is a full tile of segment B
    const int nBulkB = tB0 < min(lastB_vis, lastB_dma) ? ((min(lastB_vis, lastB_dma) - tB0) / NSTAGE) * NSTAGE : 0;
    const int tB1 = tB0 + nBulkB;

    auto tile_body = [&](auto slot_c, auto bulk_c, int t) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int MODE = decltype(bulk_c)::value;     // 0 general, 1 bulk tile of segment A, 2 of segment B
        constexpr bool BULK = MODE != 0;
        constexpr int SOFF = SLOT * STAGE_BYTES;
        constexpr int NSLOT = (SLOT + 1) % NSTAGE;            // ring slot of tile t+1
        constexpr int NSOFF = NSLOT * STAGE_BYTES;
        const bool more1 = BULK || t + 1 < nT, more2 = BULK || t + 2 < nT;
        const bool active = BULK || t < nTw;
        const bool inB = !BULK && t >= nA;
        const int key0 = inB ? (t - nA) * KVBLK : t * KVBLK;   // first key of the tile in its segment
        const int cnt = BULK ? KVBLK : (inB ? min(KVBLK, S - key0) : min(KVBLK, lenA - key0));
        // block B starts 32 rows later: a causal tile may concern block B only
        const bool skipA = !BULK && inB && key0 > wq0 + 31;

        // ring slot (t+2)%3 == (t-1)%3: its K was read in P4(t-2), its V^T in P1(t-1) of every wave — all of them
        // behind the barrier of tile t-1, which this wave has passed
        DmaPlan dn;
        dn.full = false;
        dn.kb = dn.vb = nullptr;
        dn.step = 0;
        dn.kofs = dn.vofs = dn.base = 0;
        if constexpr (!BULK) {
            if (more2) dn = dma_plan(t + 2, (SLOT + 2) % NSTAGE);
        }
        constexpr uint32_t DSLOT = ((SLOT + 2) % NSTAGE) * STAGE_BYTES;     // ring slot of tile t+2

        f32x16 sa[2], sb[2];
        u32x2 vlo[16], vhi[16];     // V^T fragment i = (k-step i>>2, dim block i&3): key quads kq and kq+2
        uint32_t pkA[16], pkB[16];  // packed P words of blocks A and B (slice i -> word i)
        float eA[2][2], eB[2][2];   // slice pipeline temporaries (pair k&1 of slice k), blocks A and B
        float psumA = 0.f, psumB = 0.f, nmA = 0.f, nmB = 0.f, hmA = 0.f, hmB = 0.f;
        bool fast_p4 = false;       // block B's slices 8..15 and its row sum are finished inside P4
        // the 16-bit ds offset field: ring slot 2 needs its base folded into the address
        const uint32_t va_ = SOFF >= 32768 ? vaddr + SOFF : vaddr;
        constexpr int VO = SOFF >= 32768 ? 0 : SOFF;
#define W64_PFA(st) (u32x4{pkA[4 * (st)], pkA[4 * (st) + 1], pkA[4 * (st) + 2], pkA[4 * (st) + 3]})
#define W64_PFB(st) (u32x4{pkB[4 * (st)], pkB[4 * (st) + 1], pkB[4 * (st) + 2], pkB[4 * (st) + 3]})
#define W64_VF(i) join_u(vlo[i], vhi[i])
#define W64_V_READ(i)                                                                  \
    do {                                                                               \
        DUO_TR_READ(vlo[i], va_, VO + ((i) >> 2) * 4096 + ((i) & 3) * 256);            \
        DUO_TR_READ(vhi[i], va_, VO + ((i) >> 2) * 4096 + ((i) & 3) * 256 + 2048);     \
    } while (0)

        if (active) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // K(t) fragments (requested in P4(t-1))
            __builtin_amdgcn_sched_barrier(0);
            if (!skipA) {
                // ---- P1: S_A, under the V^T(t) reads and the DMA of tile t+2 -----------------------------
#define W64_P1_STEP(i)                                                                   \
    do {                                                                                 \
        W64_QK(sa[(i) & 1], i, 0);                                                       \
        W64_V_READ(i);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
                W64_P1_STEP(0); W64_P1_STEP(1); W64_P1_STEP(2); W64_P1_STEP(3);
                W64_P1_STEP(4); W64_P1_STEP(5); W64_P1_STEP(6); W64_P1_STEP(7);
                W64_P1_STEP(8); W64_P1_STEP(9); W64_P1_STEP(10); W64_P1_STEP(11);
                W64_P1_STEP(12); W64_P1_STEP(13); W64_P1_STEP(14); W64_P1_STEP(15);
#undef W64_P1_STEP
                // ---- P2: S_B, under row max / rescale decision of block A (gaps 2-3) and A's slices 0..11 -------
                // (S_A's last MFMA precedes two more MFMAs before any VALU reads it: complete)
                W64_QK(sb[0], 0, 1);
                W64_QK(sb[1], 1, 1);
                W64_QK(sb[0], 2, 1);
                if constexpr (!BULK) mask_tile(sa, 0, inB, key0, cnt);
                hmA = quarter_max(sa[0][0], sa[0], 0);
                __builtin_amdgcn_sched_barrier(0);
                W64_QK(sb[1], 3, 1);
                hmA = quarter_max(hmA, sa[0], 8);
                __builtin_amdgcn_sched_barrier(0);
                W64_QK(sb[0], 4, 1);
                hmA = quarter_max(hmA, sa[1], 0);
                __builtin_amdgcn_sched_barrier(0);
                W64_QK(sb[1], 5, 1);
                decide(std::integral_constant<int, 0>{}, row_max_finish(hmA, quarter_max(sa[1][8], sa[1], 8)));
                nmA = (!BULK && mrow[0] == -INFINITY) ? 0.f : -mrow[0] * c;
                __builtin_amdgcn_sched_barrier(0);
                W64_QKB_A(6, 0); W64_QKB_AB(7, 1); W64_QKB_AB(8, 2); W64_QKB_AB(9, 3); W64_QKB_AB(10, 4);
                W64_QKB_AB(11, 5); W64_QKB_AB(12, 6); W64_QKB_AB(13, 7); W64_QKB_AB(14, 8); W64_QKB_AB(15, 9);
                __builtin_amdgcn_sched_barrier(0);
                // ---- P3: O_A, under A's slices 10..15 (their P words feed MFMAs 8..15), B's row max (gaps 7-10)
                //      and B's slices 0..4 ------------------------------------------------------------------------
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // V^T(t) fragments
                __builtin_amdgcn_sched_barrier(0);
                W64_PV_AB(0, 0, W64_VF(0), W64_PFA(0), sa, 10, eA, psumA, pkA, nmA);
                W64_PV_AB(0, 1, W64_VF(1), W64_PFA(0), sa, 11, eA, psumA, pkA, nmA);
                W64_PV_AB(0, 2, W64_VF(2), W64_PFA(0), sa, 12, eA, psumA, pkA, nmA);
                W64_PV_AB(0, 3, W64_VF(3), W64_PFA(0), sa, 13, eA, psumA, pkA, nmA);
                W64_PV_AB(0, 4, W64_VF(4), W64_PFA(1), sa, 14, eA, psumA, pkA, nmA);
                W64_PV_AB(0, 5, W64_VF(5), W64_PFA(1), sa, 15, eA, psumA, pkA, nmA);
                W64_PV_B(0, 6, W64_VF(6), W64_PFA(1), 15, eA, psumA, pkA);
                if constexpr (!BULK) mask_tile(sb, 1, inB, key0, cnt);
                __builtin_amdgcn_sched_barrier(0);
                W64_PV(0, 7, W64_VF(7), W64_PFA(1), "");
                hmB = quarter_max(sb[0][0], sb[0], 0);
                __builtin_amdgcn_sched_barrier(0);
                W64_PV(0, 8, W64_VF(8), W64_PFA(2), "");
                hmB = quarter_max(hmB, sb[0], 8);
                __builtin_amdgcn_sched_barrier(0);
                W64_PV(0, 9, W64_VF(9), W64_PFA(2), "");
                hmB = quarter_max(hmB, sb[1], 0);
                __builtin_amdgcn_sched_barrier(0);
                W64_PV(0, 10, W64_VF(10), W64_PFA(2), "");
                decide(std::integral_constant<int, 1>{}, row_max_finish(hmB, quarter_max(sb[1][8], sb[1], 8)));
                nmB = (!BULK && mrow[1] == -INFINITY) ? 0.f : -mrow[1] * c;
                __builtin_amdgcn_sched_barrier(0);
                W64_PV_A(0, 11, W64_VF(11), W64_PFA(2), sb, 0, eB, nmB);
                W64_PV_AB(0, 12, W64_VF(12), W64_PFA(3), sb, 1, eB, psumB, pkB, nmB);
                W64_PV_AB(0, 13, W64_VF(13), W64_PFA(3), sb, 2, eB, psumB, pkB, nmB);
                W64_PV_AB(0, 14, W64_VF(14), W64_PFA(3), sb, 3, eB, psumB, pkB, nmB);
                W64_PV_AB(0, 15, W64_VF(15), W64_PFA(3), sb, 4, eB, psumB, pkB, nmB);
                lsum[0] += psumA;
                fast_p4 = true;
            } else {
                // ---- block A sees nothing of this (diagonal) tile: plain sequence for block B -------------
                W64_V_READ(0); W64_V_READ(1); W64_V_READ(2); W64_V_READ(3); W64_V_READ(4); W64_V_READ(5);
                W64_V_READ(6); W64_V_READ(7); W64_V_READ(8); W64_V_READ(9); W64_V_READ(10); W64_V_READ(11);
                W64_V_READ(12); W64_V_READ(13); W64_V_READ(14); W64_V_READ(15);
#define W64_QB(i) W64_QK(sb[(i) & 1], i, 1)
                W64_QB(0); W64_QB(1); W64_QB(2); W64_QB(3); W64_QB(4); W64_QB(5); W64_QB(6); W64_QB(7);
                W64_QB(8); W64_QB(9); W64_QB(10); W64_QB(11); W64_QB(12); W64_QB(13); W64_QB(14); W64_QB(15);
#undef W64_QB
                w64_mfma_drain();
                decide(std::integral_constant<int, 1>{}, row_max(sb, 1, inB, key0, cnt));
                const float mcB = mrow[1] == -INFINITY ? 0.f : mrow[1] * c;
                psumB = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) exp_slice(sb, i, mcB, psumB, pkB);
                lsum[1] += psumB;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // V^T(t) fragments
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- tile t+1 (requested in P4 of the previous tile) must have landed, then ONE barrier: it publishes
        //      tile t+1 and retires this tile's LDS reads (K(t) in P4(t-1), V^T(t) in P1: both complete above)
        __builtin_amdgcn_sched_barrier(0);
#ifndef W64_MEAS_NO_VMWAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef W64_MEAS_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // LDS-DMA of tile t+2 into ring slot (t+2)%3 == (t-1)%3 (its K was read in P4(t-2), its V^T in P1(t-1) of
        // every wave: all behind the barrier above).  Issued in the bare-MFMA gaps of P4's second half — an
        // LDS-DMA instruction costs this wave 60-180 cycles of issue depending on what else its phase carries, and
        // with one wave per SIMD nothing hides that; among bare MFMAs it is cheapest.
#define W64_DMA(j)                                                                       \
    do {                                                                                 \
        if constexpr (MODE == 1)                                                         \
            w64_dma_pair(run_k, run_v, bk_ofs[j], bv_ofs[j], wave_lds + DSLOT + (j) * W64_NW * 1024,   \
                         wave_lds + DSLOT + K_TILE_BYTES + (j) * W64_NW * 1024);         \
        else if constexpr (MODE == 2)                                                    \
            w64_dma_pair(run_k, run_v, ck_ofs[j], cv_ofs[j], wave_lds + DSLOT + (j) * W64_NW * 1024,   \
                         wave_lds + DSLOT + K_TILE_BYTES + (j) * W64_NW * 1024);         \
        else dma_piece(dn, j);                                                           \
    } while (0)

        // ---- P4: O_B, under the K(t+1) fragment reads (a[192:255] is free: its last use was P2) -----------------
        const bool next_active = BULK || (more1 && t + 1 < nTw);
        const uint32_t ka_ = NSOFF >= 32768 ? NSOFF : 0;
        constexpr int KO = NSOFF >= 32768 ? 0 : NSOFF;
        if (active) {
#define W64_KA(j) (koff[(j) >> 1] + ka_)
#define W64_KOF(j) (KO + ((j) & 1) * 8192)
            if (fast_p4) {
                // block B's slices 5..15 ride under MFMAs 0..10 (P words 4..7 feed MFMA 4.., 8..11 MFMA 8.., 12..15 MFMA 12..)
                if (next_active) {
                    W64_PVB_K2_AB(0, W64_VF(0), W64_PFB(0), 5);
                    W64_PVB_K2_AB(1, W64_VF(1), W64_PFB(0), 6);
                    W64_PVB_K2_AB(2, W64_VF(2), W64_PFB(0), 7);
                    W64_PVB_K2_AB(3, W64_VF(3), W64_PFB(0), 8);
                    W64_PVB_K2_AB(4, W64_VF(4), W64_PFB(1), 9);
                    W64_PVB_K2_AB(5, W64_VF(5), W64_PFB(1), 10);
                    W64_PVB_K2_AB(6, W64_VF(6), W64_PFB(1), 11);
                    W64_PVB_K2_AB(7, W64_VF(7), W64_PFB(1), 12);
                } else {
                    W64_PV_AB(1, 0, W64_VF(0), W64_PFB(0), sb, 5, eB, psumB, pkB, nmB);
                    W64_PV_AB(1, 1, W64_VF(1), W64_PFB(0), sb, 6, eB, psumB, pkB, nmB);
                    W64_PV_AB(1, 2, W64_VF(2), W64_PFB(0), sb, 7, eB, psumB, pkB, nmB);
                    W64_PV_AB(1, 3, W64_VF(3), W64_PFB(0), sb, 8, eB, psumB, pkB, nmB);
                    W64_PV_AB(1, 4, W64_VF(4), W64_PFB(1), sb, 9, eB, psumB, pkB, nmB);
                    W64_PV_AB(1, 5, W64_VF(5), W64_PFB(1), sb, 10, eB, psumB, pkB, nmB);
                    W64_PV_AB(1, 6, W64_VF(6), W64_PFB(1), sb, 11, eB, psumB, pkB, nmB);
                    W64_PV_AB(1, 7, W64_VF(7), W64_PFB(1), sb, 12, eB, psumB, pkB, nmB);
                }
                W64_PV_AB(1, 8, W64_VF(8), W64_PFB(2), sb, 13, eB, psumB, pkB, nmB);
                W64_PV_AB(1, 9, W64_VF(9), W64_PFB(2), sb, 14, eB, psumB, pkB, nmB);
                W64_PV_AB(1, 10, W64_VF(10), W64_PFB(2), sb, 15, eB, psumB, pkB, nmB);
                W64_PV_B(1, 11, W64_VF(11), W64_PFB(2), 15, eB, psumB, pkB);
                lsum[1] += psumB;
            } else {
                if (next_active) {
#define W64_P4K(i) W64_PVB_K2(i, W64_VF(i), W64_PFB((i) >> 2))
                    W64_P4K(0); W64_P4K(1); W64_P4K(2); W64_P4K(3); W64_P4K(4); W64_P4K(5); W64_P4K(6); W64_P4K(7);
#undef W64_P4K
                } else {
                    W64_PV(1, 0, W64_VF(0), W64_PFB(0), ""); W64_PV(1, 1, W64_VF(1), W64_PFB(0), "");
                    W64_PV(1, 2, W64_VF(2), W64_PFB(0), ""); W64_PV(1, 3, W64_VF(3), W64_PFB(0), "");
                    W64_PV(1, 4, W64_VF(4), W64_PFB(1), ""); W64_PV(1, 5, W64_VF(5), W64_PFB(1), "");
                    W64_PV(1, 6, W64_VF(6), W64_PFB(1), ""); W64_PV(1, 7, W64_VF(7), W64_PFB(1), "");
                }
                W64_PV(1, 8, W64_VF(8), W64_PFB(2), "");
            }
            // the remaining MFMAs of each variant, with the DMA of tile t+2 in their (bare) gaps
            if (!fast_p4) {
                W64_PV(1, 9, W64_VF(9), W64_PFB(2), ""); W64_PV(1, 10, W64_VF(10), W64_PFB(2), "");
                W64_PV(1, 11, W64_VF(11), W64_PFB(2), "");
            }
            W64_PV(1, 12, W64_VF(12), W64_PFB(3), "");
            if (more2) W64_DMA(0);
            W64_PV(1, 13, W64_VF(13), W64_PFB(3), "");
            if (more2) W64_DMA(1);
            W64_PV(1, 14, W64_VF(14), W64_PFB(3), "");
            if (more2) W64_DMA(2);
            W64_PV(1, 15, W64_VF(15), W64_PFB(3), "");
            if (more2) W64_DMA(3);
#undef W64_KA
#undef W64_KOF
        } else if (more2) {
            W64_DMA(0); W64_DMA(1); W64_DMA(2); W64_DMA(3);
        }
        if constexpr (MODE == 1) {
            run_k += tileA_elems;
            run_v += tileA_elems;
        } else if constexpr (MODE == 2) {
            run_k += tileB_elems;
            run_v += tileB_elems;
        }
#undef W64_DMA
        __builtin_amdgcn_sched_barrier(0);
#undef W64_V_READ
#undef W64_PFA
#undef W64_PFB
#undef W64_VF
    };

    typedef std::integral_constant<int, 0> general_t;
    typedef std::integral_constant<int, 1> bulkA_t;
    typedef std::integral_constant<int, 2> bulkB_t;
    // one loop over ring turns (3 tiles); the flavour of a turn is wave-uniform
    for (int t = 0; t < nT; t += NSTAGE) {
        if (t < nBulk) {
            tile_body(std::integral_constant<int, 0>{}, bulkA_t{}, t);
            tile_body(std::integral_constant<int, 1>{}, bulkA_t{}, t + 1);
            tile_body(std::integral_constant<int, 2>{}, bulkA_t{}, t + 2);
        } else if (t >= tB0 && t < tB1) {
            if (t == tB0) {
                run_k = C.b.k + (int64_t)kvh * C.b.head_stride + (int64_t)(tB0 + 2 - nA) * tileB_elems;
                run_v = C.b.v + (int64_t)kvh * C.b.head_stride + (int64_t)(tB0 + 2 - nA) * tileB_elems;
            }
            tile_body(std::integral_constant<int, 0>{}, bulkB_t{}, t);
            tile_body(std::integral_constant<int, 1>{}, bulkB_t{}, t + 1);
            tile_body(std::integral_constant<int, 2>{}, bulkB_t{}, t + 2);
        } else {
            tile_body(std::integral_constant<int, 0>{}, general_t{}, t);
            if (t + 1 < nT) tile_body(std::integral_constant<int, 1>{}, general_t{}, t + 1);
            if (t + 2 < nT) tile_body(std::integral_constant<int, 2>{}, general_t{}, t + 2);
        }
    }

    // ---- epilogue: O^T / l -> out[q][qh][d] -----------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    w64_mfma_drain();
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        float ov[64];
        if (x == 0) w64_acc_read<0, 64>(ov);
        else w64_acc_read<64, 128>(ov);
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
                    w.x = cvt_pk_bf16(ov[16 * db + 4 * rq + 0] * inv, ov[16 * db + 4 * rq + 1] * inv);
                    w.y = cvt_pk_bf16(ov[16 * db + 4 * rq + 2] * inv, ov[16 * db + 4 * rq + 3] * inv);
                    *reinterpret_cast<u32x2 *>(op + d) = w;
                }
        }
    }
}

}  // namespace
