// Probe: sustained global -> LDS (LDS-DMA) and global -> VGPR rates per CU from an L2-resident buffer.
// hipcc --offload-arch=gfx950 -O3 ldsdma_rate.hip -o ldsdma_rate && ./ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>   // 0: LDS-DMA 16 B/lane, 1: global_load_dwordx4 to registers, 2: DMA + concurrent ds_read_b128 of the ring
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ src, size_t span_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // every workgroup walks the same `span_bytes` window (L2 resident after the first touch), offset by its id
    const size_t wg_off = ((size_t)blockIdx.x * 65536) % span_bytes;
    float acc = 0.f;
    uint4 r[8];
    for (int it = 0; it < iters; ++it) {
        const size_t base = (wg_off + (size_t)it * 65536) % span_bytes;     // 64 KiB per iteration per WG, like one GEMM stage
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const char* p = src + base + (size_t)(wave * 8 + i) * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + (it & 1) * 65536 + (wave * 8 + i) * 1024), 16, 0, 0);
            }
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    const uint4 v = *reinterpret_cast<const uint4*>(smem + ((it + 1) & 1) * 65536 + ((wave * 24 + i) * 1024 + lane * 16) % 65536);
                    acc += __uint_as_float(v.x);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = *reinterpret_cast<const uint4*>(src + base + (size_t)(wave * 8 + i) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += __uint_as_float(r[i].x);
        }
    }
    if (acc == 123.456f) sink[0] = acc + smem[tid];
}

int main() {
    const size_t span = 64ull << 20;   // 64 MiB window: fits the aggregate L2 + MALL
    char* buf; float* sink;
    hipMalloc(&buf, span + (1 << 20)); hipMalloc(&sink, 4);
    hipMemset(buf, 1, span + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400, nwg = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int mode = 0; mode < 3; ++mode) {
        for (size_t sp : {(size_t)4 << 20, (size_t)64 << 20}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(nwg), dim3(512), 131072, 0, buf, sp, iters, sink);
                if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(nwg), dim3(512), 131072, 0, buf, sp, iters, sink);
                if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(nwg), dim3(512), 131072, 0, buf, sp, iters, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    const double bytes = (double)nwg * iters * 65536;
                    printf("mode %d span %3zu MiB: %.3f ms  %.2f TB/s aggregate  %.1f B/clk/CU @2.4GHz\n", mode, sp >> 20, ms, bytes / ms / 1e9,
                           bytes / nwg / (ms * 1e-3 * 2.4e9));
                }
            }
        }
    }
    return 0;
}
