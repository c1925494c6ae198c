// capi.hip -- library-level entry points: error strings, version, hardware probe.
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include "common.hpp"

extern "C" const char* morec_strerror(int code) {
    switch (code) {
        case MOREC_OK: return "ok";
        case MOREC_E_ARG: return "MOREC_E_ARG: null pointer or non-positive size";
        case MOREC_E_ALIGN: return "MOREC_E_ALIGN: pointer/pitch not 16-byte aligned or size not a vector multiple";
        case MOREC_E_UNSUPPORTED: return "MOREC_E_UNSUPPORTED: shape outside kernel limits";
        case MOREC_E_DTYPE: return "MOREC_E_DTYPE: unsupported dtype combination";
        case MOREC_E_COMM: return "MOREC_E_COMM: an RCCL call failed (morec_comm_last_error)";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown morec error";
}

extern "C" int morec_version(void) { return 105; }

// Process-wide dropout seed source (see morec_hip.h): a device uint64 folded into every dropout / DropPath stream at kernel entry.
static const uint64_t* g_drop_seed_src = nullptr;
const uint64_t* morec_drop_seed_src() { return g_drop_seed_src; }
extern "C" int morec_dropout_seed_source(const void* dev_u64) {
    if (reinterpret_cast<uintptr_t>(dev_u64) & 7u) return MOREC_E_ALIGN;
    g_drop_seed_src = reinterpret_cast<const uint64_t*>(dev_u64);
    return MOREC_OK;
}

// ---- deterministic mode (common.hpp) ----------------------------------------------------------------------------------------------
static int g_deterministic = -1;
bool morec_deterministic() {
    if (g_deterministic < 0) {
        const char* e = getenv("MOREC_DETERMINISTIC");
        g_deterministic = (e && atoi(e) != 0) ? 1 : 0;
    }
    return g_deterministic != 0;
}
void morec_set_deterministic(int on) { g_deterministic = on ? 1 : 0; }

// Library-owned partial-sum scratch of the deterministic mode, one buffer per (device, stream) -- launches on a stream are ordered, so they
// share it.  nullptr = "not available": the callers return an error (never a silent fall-back to the atomic kernels).  Growth frees and
// reallocates, which a stream that is being CAPTURED must not do (a graph would keep the freed address): under capture a request the
// buffer cannot hold is refused -- run the shapes once eagerly (TrainStep does: its warm-up steps precede capture) or call
// morec_det_scratch_reserve.  Sixteen (device, stream) pairs; beyond that the least recently used buffer is dropped behind a device sync.
static float* det_scratch_impl(hipStream_t s, size_t n, bool may_grow) {
    struct Slot { hipStream_t s; int dev; float* p; size_t n; unsigned long long used; };
    static Slot slots[16];
    static int n_slots = 0;
    static unsigned long long tick = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    Slot* sl = nullptr;
    for (int i = 0; i < n_slots; ++i)
        if (slots[i].s == s && slots[i].dev == dev) sl = &slots[i];
    if (!sl) {
        if (capturing || !may_grow) return nullptr;
        if (n_slots < 16) {
            sl = &slots[n_slots++];
        } else {      // evict the least recently used pair of THIS device (its stream may be gone: wait for the device, not for the stream)
            for (int i = 0; i < n_slots; ++i)
                if (slots[i].dev == dev && (!sl || slots[i].used < sl->used)) sl = &slots[i];
            if (!sl) return nullptr;
            (void)hipDeviceSynchronize();
            if (sl->p) (void)hipFree(sl->p);
        }
        *sl = Slot{s, dev, nullptr, 0, 0};
    }
    sl->used = ++tick;
    if (sl->n < n) {
        if (capturing || !may_grow) return nullptr;
        if (sl->p) {      // launches that still read the old buffer are on this stream
            (void)hipStreamSynchronize(s);
            (void)hipFree(sl->p);
            sl->p = nullptr; sl->n = 0;
        }
        size_t want = n + n / 4;
        if (want < ((size_t)1 << 20)) want = (size_t)1 << 20;
        if (hipMalloc(reinterpret_cast<void**>(&sl->p), want * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); sl->p = nullptr; return nullptr; }
        sl->n = want;
    }
    return sl->p;
}
float* morec_det_scratch(hipStream_t s, size_t n) { return det_scratch_impl(s, n, true); }
// Pre-size the deterministic mode's scratch of `stream` to `n_floats` (outside graph capture): see include/morec_hip.h.
extern "C" int morec_det_scratch_reserve(size_t n_floats, void* stream) {
    return det_scratch_impl(reinterpret_cast<hipStream_t>(stream), n_floats, true) ? MOREC_OK : (int)hipErrorOutOfMemory;
}

__global__ __launch_bounds__(256) void det_fold_add_kernel(const float* __restrict__ part, float* __restrict__ dst, int n_parts, size_t n, size_t stride) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four chains for the memory latency; their combination order is fixed as well
        int p = 0;
        for (; p + 4 <= n_parts; p += 4) {
            a0 += part[(size_t)p * stride + j];
            a1 += part[(size_t)(p + 1) * stride + j];
            a2 += part[(size_t)(p + 2) * stride + j];
            a3 += part[(size_t)(p + 3) * stride + j];
        }
        for (; p < n_parts; ++p) a0 += part[(size_t)p * stride + j];
        dst[j] += (a0 + a1) + (a2 + a3);
    }
}
int morec_det_fold_add(const float* part, float* dst, int n_parts, size_t n, size_t stride, hipStream_t s) {
    if (n == 0 || n_parts <= 0) return MOREC_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(det_fold_add_kernel, dim3((unsigned)blocks), dim3(256), 0, s, part, dst, n_parts, n, stride);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// `waiting` will not run anything issued after this call before everything issued to `signal` so far has finished: hipEventRecord +
// hipStreamWaitEvent on an event of a process-wide ring (256 entries, created on first use on the calling thread's device; re-recording an
// event that an earlier wait captured is well defined: the wait refers to the record that was current when it was issued).  What
// torch.cuda.Stream.wait_stream does, in ONE foreign call on raw handles -- the weight-gradient stream is ordered behind the backward chain
// once per dW launch (17 ... 57 times per step), and on the launch-bound configurations the host path IS the step time.  Capturable.
extern "C" int morec_stream_wait_stream(void* waiting, void* signal) {
    static hipEvent_t ring[256];
    static std::atomic<unsigned> next{0};
    static const hipError_t made = [] {
        for (auto& e : ring) {
            const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (rc != hipSuccess) return rc;
        }
        return hipSuccess;
    }();
    if (made != hipSuccess) return (int)made;
    hipEvent_t e = ring[next.fetch_add(1u, std::memory_order_relaxed) & 255u];
    hipError_t rc = hipEventRecord(e, reinterpret_cast<hipStream_t>(signal));
    if (rc != hipSuccess) return (int)rc;
    rc = hipStreamWaitEvent(reinterpret_cast<hipStream_t>(waiting), e, 0);
    return rc == hipSuccess ? MOREC_OK : (int)rc;
}

// Probe: (a) MFMA fragment/accumulator layouts with recognisable integer data, (b) what each lane of
// ds_read_b64_tr_b16 receives when lane l supplies address base + 8*l over an LDS image lds16[x] = x.
// out[0..255]     : 16x16x32 bf16: D = A.B with A[i][k] = (k == i) ? 1 : 0 for i<16 (k<16), B[k][j] = 16*k + j
//                   (assumed layouts: lane l gives row/col l&15, k = 8*(l>>4)+e) -> out[l*4+r] should be
//                   D[(l>>4)*4+r][l&15] = 16*((l>>4)*4+r) + (l&15)
// out[256..511]   : 16x16x4 f32 with the same test (k < 4 only -> rows 0..3 nonzero)
// out[512..767]   : tr_b16 values, 4 per lane
__global__ void probe_kernel(int32_t* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds16[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds16[i] = (unsigned short)i;
    __syncthreads();
    {
        bf16x8_t a, b;
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * (l >> 4) + e;
            const float av = (k == (l & 15)) ? 1.f : 0.f;                    // A[i = l&15][k]
            const float bv = (k < 16) ? (float)(16 * k + (l & 15)) : 0.f;    // B[k][j = l&15]
            a[e] = (__bf16)av;
            b[e] = (__bf16)bv;
        }
        f32x4_t c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[l * 4 + r] = (int32_t)c[r];
    }
    {
        const int k = l >> 4;
        const float av = (k == (l & 15)) ? 1.f : 0.f;
        const float bv = (float)(16 * k + (l & 15));
        f32x4_t c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = (int32_t)c[r];
    }
    {
        uint64_t v;
        const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds16 + 8u * l;
        asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        for (int r = 0; r < 4; ++r) out[512 + l * 4 + r] = (int32_t)((v >> (16 * r)) & 0xffffu);
    }
}

extern "C" int morec_probe(int32_t* out, void* stream) {
    if (!out) return MOREC_E_ARG;
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// test hook: the keep-mask the kernels derive from (p, seed) for element indices 0 .. n-1
__global__ void drop_mask_kernel(uint8_t* out, size_t n, DropRng d) {
    d = drop_resolve(d);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = drop_keep(d, i) ? 1 : 0;
}
extern "C" int morec_dropout_keep_mask(uint8_t* out, size_t n, float p, uint64_t seed, void* stream) {
    if (!out || p < 0.f || p >= 1.f) return MOREC_E_ARG;
    if (n == 0) return MOREC_OK;
    hipLaunchKernelGGL(drop_mask_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), out, n, make_drop(p, seed));
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
