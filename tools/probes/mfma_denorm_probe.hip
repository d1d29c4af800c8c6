// Does v_mfma_f32_16x16x32_f16 keep fp16 DENORMAL inputs?  (INT4 decode: a nibble masked out of a packed word, read as
// fp16, is the denormal n * 2^-24 — usable as an MFMA operand without any conversion if the matrix core does not flush it.)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_denorm_probe mfma_denorm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
    const int lane = threadIdx.x;
    union { uint16_t u[8]; f16x8 v; } a, b;
    for (int i = 0; i < 8; ++i) {
        a.u[i] = (uint16_t)((lane + i) & 15);        // denormal n * 2^-24
        b.u[i] = 0x5C00;                              // 256.0
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
    // and the 16 n * 2^-24 form (bits 4-7)
    union { uint16_t u[8]; f16x8 v; } a2;
    for (int i = 0; i < 8; ++i) a2.u[i] = (uint16_t)(((lane + i) & 15) << 4);
    f32x4 c2 = {0.f, 0.f, 0.f, 0.f};
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2.v, b.v, c2, 0, 0, 0);
    out[lane * 2] = c[0];
    out[lane * 2 + 1] = c2[0];
}
int main() {
    float *d, h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // row 0 of the result: sum over k of A[0][k] * 256; A[row = lane & 15][k = 8 * (lane >> 4) + i] = ((lane + i) & 15) * 2^-24
    double want = 0;
    for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 8; ++i) want += (double)((g * 16 + i) & 15) * 256.0 / 16777216.0;
    printf("mfma f16 denormal inputs: got %.9g (x16 form %.9g), exact %.9g (x16: %.9g) -> %s\n", h[0], h[1], want, want * 16,
           h[0] == (float)want && h[1] == (float)(want * 16) ? "KEPT (exact)" : (h[0] == 0.f ? "FLUSHED" : "DIFFERENT"));
    return 0;
}
