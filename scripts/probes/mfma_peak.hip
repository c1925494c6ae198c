// Probe: peak issue rate of v_mfma_f32_16x16x32_bf16 with independent accumulators, 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs_per_cu : {1, 2, 4}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<16>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) {
                const double fl = 2.0 * 16 * 16 * 32 * 16.0 * iters * 4 /*waves per WG*/ * 256.0 * wgs_per_cu;
                printf("%d waves/SIMD: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wgs_per_cu, ms, fl / ms / 1e9,
                       ms * 1e-3 * 2.4e9 / (16.0 * iters * wgs_per_cu));
            }
        }
    }
    return 0;
}
