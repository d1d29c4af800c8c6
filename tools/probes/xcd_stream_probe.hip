// xcd_stream_probe — does a PLAIN streaming read show the odd/even XCD skew of the decode scan?
//
// `duo_decode_scan_kernel` finishes 3-10 % later on odd-numbered XCDs for identical work
// (profiles/r3_decode.md).  This probe takes the arithmetic away: 256 workgroups x 256 threads with the scan's exact
// workgroup -> (kv head, split) map, chunk sizes and load shape (per wave and 16-token group: four 1-KiB K loads + four
// 1-KiB V loads, 16 B per lane, non-temporal), XOR-folded so nothing is dead, `s_memrealtime` at entry and end of every
// workgroup and the hardware XCC id next to it.
//
//   mode 0  the scan's map: workgroup b -> head b / splits, split b % splits, K chunk + V chunk of that head
//   mode 1  the same chunks handed to the workgroup b ^ 1 (odd XCDs read what even XCDs read in mode 0): a skew that
//           follows the XCD is the hardware's, one that follows the chunk is an address effect
//   mode 2  one contiguous stream per workgroup (K only, pool twice as long): no second stream
//   mode 3  mode 0 with temporal (plain) loads
//   mode 4  mode 0, but the chunks of one head are dealt to workgroups of ONE XCD (head h -> XCDs 2h, 2h+1 for nf = 4)
//
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_stream_probe xcd_stream_probe.hip
// run:   ./xcd_stream_probe [nf=4] [tokens=131072] [reps=7]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));             \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Stamp {
    unsigned long long t0, t1;
    uint32_t xcc, pad;
};

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const char *p) {
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}

template <bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const char *__restrict__ kpool, const char *__restrict__ vpool,
                                                     int tokens, int splits, int mode, int nf, Stamp *stamps,
                                                     uint32_t *sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, tg = lane >> 4;
    int b = blockIdx.x;
    if (mode == 1) b ^= 1;
    if (mode == 4) {
        // workgroup b sits on XCD b % 8, slot b / 8 of it; head h owns XCDs [h * 8 / nf, (h + 1) * 8 / nf)
        const int x = b & 7, slot = b >> 3, per = 8 / nf;         // (nf in {1, 2, 4, 8})
        const int head = x / per;
        b = head * splits + (x % per) * 32 + slot;
    }
    const int head = b / splits, split = b - head * splits;
    const int units = (tokens + 63) >> 6;
    const int uq = units / splits, ur = units - uq * splits;
    const int u0 = split * uq + min(split, ur), u1 = u0 + uq + (split < ur ? 1 : 0);
    const int c0 = u0 << 6, c1 = min(u1 << 6, tokens);
    const int per_wave = (((c1 - c0 + 3) >> 2) + 15) & ~15;
    const int w0 = c0 + wave * per_wave, w1 = min(w0 + per_wave, c1);
    const size_t head_bytes = (size_t)tokens * 256;
    const char *kA = kpool + (size_t)head * head_bytes, *vA = vpool + (size_t)head * head_bytes;
    if (mode == 2) {   // one stream: the pool is 2 * tokens rows per head, this workgroup's chunk doubled
        kA = kpool + (size_t)head * 2 * head_bytes;
        vA = kA + (size_t)(c1 - c0) * 256;    // second half of the same contiguous run
    }
    const uint32_t roff = (uint32_t)tg * 256u + (uint32_t)sub * 16u;
    u32x4 acc = {0u, 0u, 0u, 0u};
    u32x4 k0[4], v0[4], k1[4], v1[4];
    auto load = [&](int t, u32x4 (&kb)[4], u32x4 (&vb)[4]) {
        const size_t base = (mode == 2) ? ((size_t)c0 * 512 + (size_t)(t - c0) * 256) : (size_t)t * 256;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kb[u] = ld16<NT>(kA + base + (size_t)u * 1024 + roff);
            vb[u] = ld16<NT>(vA + base + (size_t)u * 1024 + roff);
        }
    };
    auto fold = [&](u32x4 (&kb)[4], u32x4 (&vb)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= kb[u] ^ vb[u];
    };
    const int n = (w1 - w0) / 16;
    if (n > 0) {
        const int t_last = w0 + (n - 1) * 16;
        load(w0, k0, v0);
        for (int t = w0;; t += 32) {
            load(min(t + 16, t_last), k1, v1);
            __builtin_amdgcn_sched_barrier(0);
            fold(k0, v0);
            if (t >= t_last) break;
            load(min(t + 32, t_last), k0, v0);
            __builtin_amdgcn_sched_barrier(0);
            fold(k1, v1);
            if (t + 16 >= t_last) break;
        }
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x9e3779b9u) sink[0] = r;    // keeps the loads alive
    __syncthreads();
    if (threadIdx.x == 0) {
        Stamp s;
        s.t0 = t0;
        s.t1 = __builtin_amdgcn_s_memrealtime();
        s.xcc = xcc;
        s.pad = 0;
        stamps[blockIdx.x] = s;
    }
}

int main(int argc, char **argv) {
    const int nf = argc > 1 ? atoi(argv[1]) : 4;
    const int tokens = argc > 2 ? atoi(argv[2]) : 131072;
    const int reps = argc > 3 ? atoi(argv[3]) : 7;
    const int wgs = 256, splits = wgs / nf;
    const size_t pool = (size_t)nf * tokens * 256;
    char *k, *v;
    Stamp *st;
    uint32_t *sink;
    CK(hipMalloc(&k, 2 * pool));      // (mode 2 reads a K pool twice as long)
    CK(hipMalloc(&v, pool));
    CK(hipMalloc(&st, wgs * sizeof(Stamp)));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(k, 1, 2 * pool));
    CK(hipMemset(v, 2, pool));
    printf("xcd_stream_probe: nf=%d tokens=%d splits=%d, %.1f MB per launch, %d reps (10 ns ticks from the first entry)\n", nf,
           tokens, splits, 2.0 * pool / 1e6, reps);
    std::vector<Stamp> h(wgs);
    for (int mode = 0; mode < 5; ++mode) {
        if (mode == 4 && (8 % nf)) continue;
        std::vector<std::vector<double>> ends(8), durs(8);
        std::vector<double> spans;
        int mism = 0;
        for (int r = 0; r < reps + 2; ++r) {
            if (mode == 3) stream_kernel<false><<<wgs, 256>>>(k, v, tokens, splits, 0, nf, st, sink);
            else stream_kernel<true><<<wgs, 256>>>(k, v, tokens, splits, mode, nf, st, sink);
            CK(hipDeviceSynchronize());
            if (r < 2) continue;
            CK(hipMemcpy(h.data(), st, wgs * sizeof(Stamp), hipMemcpyDeviceToHost));
            unsigned long long first = ~0ull, last = 0;
            for (auto &s : h) first = std::min(first, s.t0), last = std::max(last, s.t1);
            double xe[8] = {0}, xd[8] = {0};
            int xn[8] = {0};
            for (int b = 0; b < wgs; ++b) {
                const int x = h[b].xcc & 7;
                if (x != (b & 7)) ++mism;
                xe[x] = std::max(xe[x], (double)(h[b].t1 - first));
                xd[x] += (double)(h[b].t1 - h[b].t0);
                ++xn[x];
            }
            for (int x = 0; x < 8; ++x) ends[x].push_back(xe[x]), durs[x].push_back(xn[x] ? xd[x] / xn[x] : 0);
            spans.push_back((double)(last - first));
        }
        auto med = [](std::vector<double> a) { std::sort(a.begin(), a.end()); return a[a.size() / 2]; };
        const double span = med(spans);
        printf("mode %d  span %6.0f ticks = %6.2f us  %.2f TB/s  (xcc != b%%8: %d)\n", mode, span, span / 100.0,
               2.0 * pool / (span * 1e-8) / 1e12, mism);
        printf("   last end per XCD :");
        for (int x = 0; x < 8; ++x) printf(" %6.0f", med(ends[x]));
        printf("\n   mean WG duration :");
        for (int x = 0; x < 8; ++x) printf(" %6.0f", med(durs[x]));
        double ev = 0, od = 0;
        for (int x = 0; x < 8; ++x) (x & 1 ? od : ev) += med(durs[x]) / 4;
        printf("\n   odd / even mean duration = %.4f\n", od / ev);
    }
    return 0;
}
