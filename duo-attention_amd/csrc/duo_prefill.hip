// duo_prefill.hip — chunked-prefill flash attention for both DuoAttention head
// classes in one launch (gfx950, bf16 MFMA 32x32x16): the launcher / C entry points and the key-range-split merge kernel.
//
// Which kernel runs: every product launch runs duo_prefill_w64_kernel (4 waves x 64 rows, duo_prefill_w64.h) since
// round 2.  The round-1 kernel (8 waves x 32 rows) lives in its own translation unit, duo_prefill_w32_debug.hip, behind
// duo_prefill_w32_launch(): it serves the gather (non-transposed-LDS) debug layout of debug bit 0 and the same-box A/B /
// cross-check against the w64 kernel (debug bit 7, DUO_PREFILL_W64=0) — the GPU tests run every prefill case on both.
//
// Replaces flash_attn_func at duo_attn/patch/llama.py:366-372 (first chunk: all
// heads causal over the chunk) and llama.py:392-421 (later chunks: retrieval
// heads over the whole full-KV pool, streaming heads over
// [sink+recent pool rows ++ the chunk]).  Semantics: keys = segA (all visible)
// ++ segB (the S new rows, causal: query i sees j <= i) — flash-attn's
// bottom-right aligned causal mask; fp32 scores/softmax, P rounded to bf16
// before P.V (as FA2 does), bf16 output.
//
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "duo_prefill_common.h"
#include "duo_prefill_w64.h"

namespace {

// Combine the ks[c] partials of one (q tile, q head): out = sum_s 2^((m_s - M) c) O_s / sum_s 2^((m_s - M) c) l_s.
// Block = (split item, 32-row slice of its 256 rows): items of class 0 first, in the canonical item order of
// prefill_map_block (q-tile rank, kv head, q head of the group); thread = one row x one group of 16 dims.
// (Rounds 1-5 ran one block per item that walked its 8 row slices and its pieces one after the other — sixteen to eighty
// DEPENDENT memory round trips per block on a grid of a few dozen blocks: 50-70 us per launch whatever the byte count,
// profiles/r6_prefill_plan.md.  Here the grid is 8x larger, every (m, l) pair of the row is requested up front and the
// accumulator rows follow four pieces at a time.)
template <bool F16>
__global__ __launch_bounds__(256) void duo_prefill_merge_kernel(const PrefillParams P, int n_merge0) {
    int j = blockIdx.x >> 3;
    const int pass = blockIdx.x & 7;
    const int ci = j < n_merge0 ? 0 : 1;
    if (ci) j -= n_merge0;
    const DuoClassDev &C = ci ? P.cls[1] : P.cls[0];
    const int ks = ci ? P.ks[1] : P.ks[0];
    const int nq_c = C.n_kv_heads * P.group;
    const int rank = j / nq_c;
    const int tile = P.n_qtiles - 1 - rank;
    const int p = j - rank * nq_c;
    const int kvh = p / P.group;
    const int g = p - kvh * P.group;
    const int qh = C.q_head_offset + kvh * P.group + g;
    const int64_t part0 = (int64_t)(ci ? P.pbase[1] : P.pbase[0]) + (int64_t)j * ks + (int64_t)blockIdx.y * P.nparts;
    const int jd = threadIdx.x & 7;
    const int r = pass * 32 + (threadIdx.x >> 3);
    const int q = tile * QBLK + r;
    if (q >= P.S) return;
    bf16_t *op = P.out + (int64_t)blockIdx.y * P.o_bs + (int64_t)q * P.o_ts + (int64_t)qh * P.o_hs;
    prefill_merge_row<F16>(P, ks, part0 * QBLK + r, op, jd);
}

}  // namespace

static uint32_t g_debug_flags = 0;
extern "C" void duo_set_debug_flags(uint32_t flags) { g_debug_flags = flags; }
extern "C" uint32_t duo_get_debug_flags(void) { return g_debug_flags; }

// ------------------------------------------------------------------------------------------------------------------
// Launch planner: key-range splits per head class.
//
// One workgroup per CU is resident (96 KiB of LDS), the dispatcher hands the next block id to the first CU that frees up,
// and the blocks of a launch are far from equal: retrieval-class workgroups walk the whole pool (hundreds to thousands of
// 64-key tiles), streaming-class ones the window plus the chunk's own rows (6 ... 262 tiles).  A launch whose long
// workgroups are fewer than the CUs, or not a multiple of them — small chunks, ROW BLOCKS of the layer pipeline, layers
// with one to three retrieval kv heads — leaves CUs idle while the last round drains, and the short workgroups that
// backfill behind 256 equal long ones run on however few CUs are free first.  Rounds 1-5 chose the retrieval class's
// split count k from rounds(long * k) / k; that ignores the streaming class and the q tiles' different lengths, and it
// measured 0.935 / 0.872 / 0.748 of the whole-chunk rate on 4096- / 2048- / 1024-row blocks (profiles/r5_scaling_model.md).
//
// Now the launch is REPLAYED on paper for every candidate (k0, k1): list scheduling of the blocks, in block order, on
// 256 CUs, a block costing t_fix + tiles * t_tile, plus the merge pass (t_merge + partials * t_part) when anything is
// split — the model that reproduces those three measured figures to 1 % (tools/prefill_plan_model.py) — and the cheapest
// candidate wins.  Both classes can be split (a streaming head's 262-tile workgroups halve), the count goes up to 16, and
// the XCD-aware block order now covers split launches too (one key-range piece of one kv head = one K/V stream).
// Plans are memoised by launch shape: one replay per distinct (head counts, past, rows) — the 32 layers of a chunk share
// seven or eight of them.
// ------------------------------------------------------------------------------------------------------------------
constexpr int64_t kPrefillPartialBytes = (int64_t)QBLK * (DUO_HEAD_DIM + 2) * sizeof(float);
constexpr int kPrefillMaxSplit = 16;
constexpr int kPrefillCUs = 256;

namespace {
struct PlanCost {
    double t_tile, t_fix, t_merge, t_part, t_pad;     // microseconds
    double c0;      // cost of a unit of work with (nearly) all CUs idle, relative to the full chip (tail of a launch)
};
static const PlanCost &plan_cost() {
    // defaults: fitted to same-box probes of this kernel (profiles/r6_prefill_plan.md); DUO_PREFILL_PLAN_COST overrides
    static const PlanCost c = [] {
        // t_fix is what a workgroup costs beyond its bulk tiles: prologue (Q fragments, two tiles in flight), the handful of
        // tiles in the general (masked / run-boundary) form, epilogue; c0: fewer active CUs clock higher and share the fabric
        // with fewer others — a lone 1800-tile workgroup walks a tile in 1.2 us, 256 of them in 1.6
        // (fit over 280 timed launches, tools/prefill_plan_model.py fit: rms 3.7 %; t_merge = the merge launch incl. the
        //  kernel boundary in front of it and the partial stores of the last round)
        PlanCost v{1.62, 26.5, 41.0, 0.0, 1.0, 0.62};
        if (const char *e = getenv("DUO_PREFILL_PLAN_COST"))
            sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf", &v.t_tile, &v.t_fix, &v.t_merge, &v.t_part, &v.t_pad, &v.c0);
        return v;
    }();
    return c;
}

struct PlanShape {
    int32_t nkv[2], group, nq, S, lenA[2], lenB[2], max_parts, xmap1;
    uint32_t force;
    bool operator==(const PlanShape &o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
};
struct PlanShapeHash {
    size_t operator()(const PlanShape &k) const {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&k);
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < sizeof(PlanShape) / 4; ++i) h = (h ^ w[i]) * 1099511628211ull;
        return (size_t)h;
    }
};
struct PrefillPlan {
    int ks[2];
    double est_us, est_unsplit_us;
};

// tiles of class c's (q tile of rank r): segment A + the causal tiles of segment B up to the tile's last row
static inline int plan_tiles(const PlanShape &K, int c, int rank) {
    const int tile = K.nq - 1 - rank;
    const int last_q = std::min(tile * QBLK + QBLK - 1, K.S - 1);
    const int nA = (K.lenA[c] + KVBLK - 1) / KVBLK;
    return nA + (last_q + (K.lenB[c] - K.S)) / KVBLK + 1;
}

// fills the order-related fields of P for the split counts (k0, k1); returns the number of blocks of the launch
static int plan_layout(PrefillParams &P, const PlanShape &K, int k0, int k1, bool xmap0, bool xmap1) {
    const int ks[2] = {k0, k1};
    int nblk_c[2] = {0, 0}, parts = 0;
    for (int c = 0; c < 2; ++c) {
        P.ks[c] = ks[c];
        P.xmap_rows[c] = P.xmap_q[c] = 0;
        P.pbase[c] = parts;
        if (K.nkv[c] <= 0) continue;
        const int items = K.nkv[c] * K.group * K.nq;
        if (ks[c] > 1) parts += items * ks[c];
        nblk_c[c] = items * ks[c];
        if (c == 0 ? xmap0 : xmap1) {
            const int row_items = K.nkv[c] * ks[c] * K.group;
            int rows = 1;
            while ((rows * row_items) % 8 != 0) rows *= 2;       // 1, 2, 4 or 8 rows: the first multiple of 8 workgroups
            const int periods = (K.nq + rows - 1) / rows;
            P.xmap_rows[c] = rows;
            P.xmap_q[c] = rows * row_items / 8;
            nblk_c[c] = periods * rows * row_items;               // the padded last period
        } else if (c == 0) {
            nblk_c[0] = (nblk_c[0] + 7) & ~7;                     // class 1's block ids keep their XCD (b % 8); extra blocks map past n_qtiles
        }
    }
    P.nparts = parts;
    P.nblk_full = nblk_c[0];
    return nblk_c[0] + nblk_c[1];
}

static double plan_replay(const PlanShape &K, int k0, int k1, bool xmap0, bool xmap1, const int *tiles /* [2][nq] */) {
    PrefillParams P{};
    P.group = K.group;
    P.n_qtiles = K.nq;
    P.cls[0].n_kv_heads = K.nkv[0];
    P.cls[1].n_kv_heads = K.nkv[1];
    const int nblk = plan_layout(P, K, k0, k1, xmap0, xmap1);
    const PlanCost &C = plan_cost();
    // min-heap of the CUs' free times
    double heap[kPrefillCUs];
    for (int i = 0; i < kPrefillCUs; ++i) heap[i] = 0.0;
    double last_start = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const PrefillItem I = prefill_map_block(P, b);
        double cost = C.t_pad;
        if (I.tile >= 0) {
            const int nT = tiles[I.ci * K.nq + (K.nq - 1 - I.tile)];
            const int n = (int)((int64_t)(I.split + 1) * nT / I.ks) - (int)((int64_t)I.split * nT / I.ks);
            cost = C.t_fix + C.t_tile * n;
        }
        // replace the root (earliest free CU) and sift down
        last_start = heap[0];
        const double t = heap[0] + cost;
        int i = 0;
        for (;;) {
            int l = 2 * i + 1, r = l + 1, m = i;
            double mv = t;
            if (l < kPrefillCUs && heap[l] < mv) { m = l; mv = heap[l]; }
            if (r < kPrefillCUs && heap[r] < mv) { m = r; mv = heap[r]; }
            if (m == i) break;
            heap[i] = heap[m];
            i = m;
        }
        heap[i] = t;
    }
    // While blocks are waiting every CU is busy and a unit of work costs 1; behind the start of the last block the busy
    // CUs drain one by one, and with `a` of 256 busy a unit costs c0 + (1 - c0) a / 256.
    std::sort(heap, heap + kPrefillCUs);
    double end = last_start, prev = last_start;
    int first = 0;
    while (first < kPrefillCUs && heap[first] <= last_start) ++first;      // CUs already idle when the last block started
    for (int i = first, a = kPrefillCUs - first; i < kPrefillCUs; ++i, --a) {
        end += (heap[i] - prev) * (C.c0 + (1.0 - C.c0) * a / kPrefillCUs);
        prev = heap[i];
    }
    if (P.nparts > 0) end += C.t_merge + C.t_part * P.nparts;
    return end;
}

static PrefillPlan plan_compute(const PlanShape &K, bool xmap0) {
    PrefillPlan best{{1, 1}, 0.0, 0.0};
    std::vector<int> tiles(2 * (size_t)K.nq, 0);
    int kmax[2] = {1, 1};
    for (int c = 0; c < 2; ++c) {
        if (K.nkv[c] <= 0) continue;
        int mn = 1 << 30;
        for (int r = 0; r < K.nq; ++r) {
            tiles[c * K.nq + r] = plan_tiles(K, c, r);
            mn = std::min(mn, tiles[c * K.nq + r]);
        }
        kmax[c] = std::max(1, std::min(kPrefillMaxSplit, mn));     // every piece walks at least one tile
    }
    const int items[2] = {K.nkv[0] * K.group * K.nq, K.nkv[1] * K.group * K.nq};
    auto fits = [&](int k0, int k1) {
        const int64_t parts = (k0 > 1 ? (int64_t)items[0] * k0 : 0) + (k1 > 1 ? (int64_t)items[1] * k1 : 0);
        return parts <= K.max_parts;
    };
    const int f0 = (int)(K.force & 0xffu), f1 = (int)((K.force >> 8) & 0xffu);
    if (K.force & 0x10000u) return best;                      // no split at all
    if (f0 || f1) {                                            // forced counts (tests, tuning): clamp to what is legal
        int k0 = f0 ? std::min(f0, kmax[0]) : 1, k1 = f1 ? std::min(f1, kmax[1]) : 1;
        while (k0 > 1 && !fits(k0, k1)) --k0;
        while (k1 > 1 && !fits(k0, k1)) --k1;
        best.ks[0] = k0;
        best.ks[1] = k1;
        best.est_unsplit_us = plan_replay(K, 1, 1, xmap0, K.xmap1 != 0, tiles.data());
        best.est_us = (k0 == 1 && k1 == 1) ? best.est_unsplit_us : plan_replay(K, k0, k1, xmap0, K.xmap1 != 0, tiles.data());
        return best;
    }
    best.est_unsplit_us = best.est_us = plan_replay(K, 1, 1, xmap0, K.xmap1 != 0, tiles.data());
    static const int cand0[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16}, cand1[] = {1, 2, 3, 4};
    for (int k0 : cand0) {
        if (k0 > kmax[0]) break;
        for (int k1 : cand1) {
            if (k1 > kmax[1]) break;
            if (k0 == 1 && k1 == 1) continue;
            if (!fits(k0, k1)) continue;
            // more than 16 rounds of workgroups: the last round's quantisation is below what another split costs (and the
            // replay stays cheap)
            if ((int64_t)items[0] * k0 + (int64_t)items[1] * k1 > 16 * kPrefillCUs) continue;
            const double t = plan_replay(K, k0, k1, xmap0, K.xmap1 != 0, tiles.data());
            if (t < best.est_us * 0.995) {      // a split must buy at least half a per cent
                best.est_us = t;
                best.ks[0] = k0;
                best.ks[1] = k1;
            }
        }
    }
    return best;
}

static PrefillPlan plan_lookup(const PlanShape &K, bool xmap0) {
    static std::mutex mu;
    static std::unordered_map<PlanShape, PrefillPlan, PlanShapeHash> memo;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = memo.find(K);
        if (it != memo.end()) return it->second;
    }
    const PrefillPlan p = plan_compute(K, xmap0);
    std::lock_guard<std::mutex> g(mu);
    if (memo.size() > 65536) memo.clear();
    memo.emplace(K, p);
    return p;
}
}  // namespace

// the plan of the last prefill launch of this thread (tests, tools/debug probes): {k0, k1, estimated us, estimated us unsplit}
static thread_local double g_last_plan[4] = {1, 1, 0, 0};
extern "C" void duo_debug_prefill_last_plan(double *out4) {
    for (int i = 0; i < 4; ++i) out4[i] = g_last_plan[i];
}
// Host-only: the plan the launcher would choose for a launch shape, and (blocks != NULL) the work of every block id of
// that launch — {class, q tile or -1, kv head, q head of the group, piece, partial slot} — as the kernels map it.  No GPU
// needed: tests/test_prefill_plan.py checks on the CPU that every (class, q tile, q head, piece) is covered exactly once.
// shape = {n_kv_heads of class 0, of class 1, group, n_tokens, lenA0, lenB0, lenA1, lenB1, max partials, xmap (bit 0: class
// 0, bit 1: class 1)}; force: bits 0-7 / 8-15 forced piece counts, bit 16 no split.  out = {k0, k1, blocks, partials,
// blocks of class 0}.  Returns the number of blocks (also when `blocks` is too small to hold them).
extern "C" int32_t duo_debug_prefill_plan(const int32_t *shape, uint32_t force, int32_t *out5, double *est2,
                                          int32_t *blocks, int32_t blocks_cap) {
    if (!shape || !out5) return DUO_EINVAL;
    PlanShape K;
    memset(&K, 0, sizeof(K));
    K.nkv[0] = shape[0]; K.nkv[1] = shape[1]; K.group = shape[2]; K.S = shape[3];
    K.lenA[0] = K.nkv[0] > 0 ? shape[4] : 0; K.lenB[0] = K.nkv[0] > 0 ? shape[5] : 0;
    K.lenA[1] = K.nkv[1] > 0 ? shape[6] : 0; K.lenB[1] = K.nkv[1] > 0 ? shape[7] : 0;
    K.max_parts = shape[8];
    K.nq = (K.S + QBLK - 1) / QBLK;
    const bool xmap0 = shape[9] & 1, xmap1 = (shape[9] & 2) != 0;
    K.xmap1 = xmap1;
    K.force = K.max_parts <= 0 ? 0x10000u : force;
    if (K.group <= 0 || K.S <= 0 || K.nkv[0] < 0 || K.nkv[1] < 0) return DUO_EINVAL;
    const PrefillPlan plan = plan_compute(K, xmap0);
    PrefillParams P{};
    P.group = K.group;
    P.n_qtiles = K.nq;
    P.cls[0].n_kv_heads = K.nkv[0];
    P.cls[1].n_kv_heads = K.nkv[1];
    const int nblk = plan_layout(P, K, plan.ks[0], plan.ks[1], xmap0, xmap1);
    out5[0] = plan.ks[0]; out5[1] = plan.ks[1]; out5[2] = nblk; out5[3] = P.nparts; out5[4] = P.nblk_full;
    if (est2) { est2[0] = plan.est_us; est2[1] = plan.est_unsplit_us; }
    for (int b = 0; blocks && b < nblk && b < blocks_cap; ++b) {
        const PrefillItem I = prefill_map_block(P, b);
        int32_t *o = blocks + 6 * (int64_t)b;
        o[0] = I.ci; o[1] = I.tile; o[2] = I.kvh; o[3] = I.g; o[4] = I.split; o[5] = I.part;
    }
    return nblk;
}

extern "C" int64_t duo_attn_prefill_workspace_bytes(void) { return 2048 * kPrefillPartialBytes; }

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
    if ((q_token_stride | q_head_stride) & 7) return DUO_EINVAL;
    if ((out_token_stride | out_head_stride) & 3) return DUO_EINVAL;
    if (nblk == 0) return 0;

    hipStream_t st = (hipStream_t)stream;
    const bool tr = !(g_debug_flags & 1u);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return DUO_EINVAL;
    // 4-wave x 64-row kernel (duo_prefill_w64.h): the default for bf16 and fp16 with the
    // transposed-V LDS layout; the gather debug path runs on the 8-wave kernel.  DUO_PREFILL_W64=0 (or debug
    // flag bit 7) keeps the 8-wave kernel everywhere (same-box A/B, tests of both kernels).
    static const bool want_w64 = [] { const char *e = getenv("DUO_PREFILL_W64"); return !e || atoi(e) != 0; }();
    const bool w64_ok = want_w64 && tr && !(g_debug_flags & 128u);
    // XCD-aware block order (DUO_PREFILL_XMAP=0 / debug bit 10: plain order; DUO_PREFILL_XMAP1=0: plain order for the
    // streaming class only)
    static const bool want_xmap = [] { const char *e = getenv("DUO_PREFILL_XMAP"); return !e || atoi(e) != 0; }();
    static const bool want_xmap1 = [] { const char *e = getenv("DUO_PREFILL_XMAP1"); return !e || atoi(e) != 0; }();
    const bool xmap0 = w64_ok && want_xmap && !(g_debug_flags & 1024u);
    const bool xmap1 = xmap0 && want_xmap1;

    // ---- plan: key-range splits per class (needs the caller's workspace) ----------------------------------------------
    PlanShape K;
    memset(&K, 0, sizeof(K));
    for (int c = 0; c < 2; ++c) {
        K.nkv[c] = P.cls[c].n_kv_heads;
        K.lenA[c] = K.nkv[c] > 0 ? P.cls[c].a.len : 0;
        K.lenB[c] = K.nkv[c] > 0 ? P.cls[c].b.len : 0;
    }
    K.group = group;
    K.nq = P.n_qtiles;
    K.S = n_tokens;
    K.xmap1 = xmap1;
    K.max_parts = workspace ? (int32_t)std::min<int64_t>(workspace_bytes / n_batch / kPrefillPartialBytes, 1 << 20) : 0;   // every batch row has its own partials
    {
        static const int forced0 = [] { const char *e = getenv("DUO_PREFILL_KSPLIT"); return e ? atoi(e) : 0; }();    // tuning / test knobs: force a split count
        static const int forced1 = [] { const char *e = getenv("DUO_PREFILL_KSPLIT1"); return e ? atoi(e) : 0; }();
        static const bool legacy = [] { const char *e = getenv("DUO_PREFILL_PLANNER"); return e && atoi(e) == 0; }();
        int f0 = forced0 > 0 ? forced0 : (int)((g_debug_flags >> 12) & 15u);     // debug bits 12-15 / 16-19: tests force a count
        int f1 = forced1 > 0 ? forced1 : (int)((g_debug_flags >> 16) & 15u);
        if (legacy && !f0 && !f1) {
            // the round-1..5 policy (same-box A/B): retrieval class only, k <= 8, rounds(long * k) / k + 0.03 (k - 1)
            const int long_wgs = K.nkv[0] * group * K.nq;
            f0 = 1;
            if (long_wgs > 0 && K.max_parts > 0) {
                const int kmax = (int)std::min<int64_t>(8, std::min<int64_t>(plan_tiles(K, 0, K.nq - 1), K.max_parts / long_wgs));
                double best_cost = 1e30;
                for (int k = 1; k <= kmax; ++k) {
                    const double cost = (double)((long_wgs * k + 255) / 256) / k + 0.03 * (k - 1);
                    if (cost < best_cost - 1e-9) { best_cost = cost; f0 = k; }
                }
            }
        }
        K.force = (uint32_t)std::min(f0, 255) | ((uint32_t)std::min(f1, 255) << 8);
        if ((g_debug_flags & 256u) || K.max_parts <= 0) K.force = 0x10000u;      // debug bit 8: no key-range split
    }
    const PrefillPlan plan = plan_lookup(K, xmap0);
    g_last_plan[0] = plan.ks[0]; g_last_plan[1] = plan.ks[1]; g_last_plan[2] = plan.est_us; g_last_plan[3] = plan.est_unsplit_us;
    nblk = plan_layout(P, K, plan.ks[0], plan.ks[1], xmap0, xmap1);
    P.ws_o = nullptr;
    P.ws_ml = nullptr;
    if (P.nparts > 0) {
        P.ws_o = (float *)workspace;
        P.ws_ml = P.ws_o + (int64_t)n_batch * P.nparts * QBLK * DUO_HEAD_DIM;
    }
    const int n_merge0 = P.ks[0] > 1 ? K.nkv[0] * group * K.nq : 0, n_merge1 = P.ks[1] > 1 ? K.nkv[1] * group * K.nq : 0;

    if (w64_ok) {
        // (with the generated bulk schedule it wins on every launch shape, first chunks and streaming-only launches
        // included: +9 ... +14 %, profiles/r2_prefill_w64.md)
        static std::atomic<bool> w64_attr[64][2];
        const void *wfn = F16 ? (const void *)duo_prefill_w64_f16_kernel : (const void *)duo_prefill_w64_kernel;
        if (dev >= 64 || !w64_attr[dev][F16].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(wfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            if (dev < 64) w64_attr[dev][F16].store(true, std::memory_order_release);
        }
        if constexpr (F16) hipLaunchKernelGGL(duo_prefill_w64_f16_kernel, dim3(nblk, n_batch), dim3(256), LDS_BYTES, st, P);
        else hipLaunchKernelGGL(duo_prefill_w64_kernel, dim3(nblk, n_batch), dim3(256), LDS_BYTES, st, P);
        DUO_HIP_CHECK_LAUNCH();
        if (n_merge0 + n_merge1 > 0) {
            hipLaunchKernelGGL((duo_prefill_merge_kernel<F16>), dim3(8 * (n_merge0 + n_merge1), n_batch), dim3(256), 0, st, P, n_merge0);
            DUO_HIP_CHECK_LAUNCH();
        }
        return 0;
    }
    {       // debug / cross-check paths: the 8-wave x 32-row kernel (duo_prefill_w32_debug.hip)
        const int rc = duo_prefill_w32_launch(&P, tr, F16, nblk, n_batch, dev, st);
        if (rc) return rc;
    }
    DUO_HIP_CHECK_LAUNCH();
    if (n_merge0 + n_merge1 > 0) {
        hipLaunchKernelGGL((duo_prefill_merge_kernel<F16>), dim3(8 * (n_merge0 + n_merge1), n_batch), dim3(256), 0, st, P, n_merge0);
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

#ifdef W64_WGTIME      /* measurement builds only (tools/debug/w64_wgtime.py): the life of every workgroup of the last bf16 launch */
extern "C" int duo_debug_w64_wgtime(unsigned long long *host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(w64_wgtime), sizeof(unsigned long long) * 8 * (size_t)std::min(n_blocks, 8192));
}
#endif
#ifdef W64_TIMING      /* measurement builds only (tools/debug): per-phase cycle sums of the last w64 launch */
extern "C" int duo_debug_w64_timing(uint32_t *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(w64_timing), 8 * sizeof(uint32_t));
}
#endif
