// gemm_skinny.hip -- NT GEMM for NARROW outputs over very many rows:  C[M, N] = A[M, K] . B[N, K]^T  with N <= 128 (N <= 288 when K <= 96), K <= 384, M in the
// millions: the Swin stage-1 / stage-2 products whose output is C = 96 ... 128 wide (attention output projection, MLP fc2, the dX
// products of fc1 / o_proj / qkv, the patch embedding: HF modeling_swin.py SwinSelfOutput / SwinOutput / SwinPatchEmbeddings via
// V/model/encoders.py:30-31, and their autograd backward).  These are streaming problems -- 2 N K / ((K + N) 2) < 160 FLOP per byte,
// half the machine balance -- that the 128 x 128 / 256 x 256 tile kernels ran at 2 TB/s: a tile's K loop is 2 - 6 stages, each one
// a full drain of the DMA queue, and a quarter to a half of every tile is padding.
//
// Here the small operand never touches LDS: a wave keeps ITS columns of B (N / 2 columns x all of K, <= 144 registers) in MFMA-fragment
// form for the whole launch, and A streams through a three-stage LDS ring of 32-row tiles filled by LDS-DMA (buffer_load ... lds) with a
// counted vmcnt -- two tiles are in flight while one is multiplied, one workgroup barrier per tile.  Four waves as 2 (16-row block) x 2
// (column half) on v_mfma_f32_16x16x32_{bf16,f16}; two workgroups per CU (<= 80 KiB of LDS each).  The output tile goes through a
// double-buffered LDS staging area so that the global stores are 16-byte lanes along rows (a 32 x N tile is one contiguous run of
// memory when ldc == N).  Persistent grid: workgroup b takes tiles b, b + grid, ...
#include <stdlib.h>
#include "gemm_core.hpp"
#include "gemm_args.hpp"

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int TR = 32;                 // rows per tile
constexpr int NST = 3;                 // LDS ring stages
constexpr int THREADS = 256;

struct SkArgs {
    const bf16* A;
    const bf16* B;
    bf16* C;
    const float* bias;     // fp32 [N] added to every row (nn.Linear bias), or null
    int M, N, K, lda, ldb, ldc, tiles;
};

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// workgroup barrier WITHOUT the full memory fence of __syncthreads() (which is `s_waitcnt vmcnt(0) lgkmcnt(0)`: it would drain the DMA ring
// on every tile): this wave's LDS writes are complete (lgkmcnt), the counted vmcnt in front of the call has retired the DMA pieces the
// barrier publishes
__device__ __forceinline__ void bar_lds() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// KS = MFMA k-steps (32 elements each) that cover K; NBH = 16-column blocks per wave (N = 32 NBH)
// (A / B / C are `__restrict__` parameters of an INLINED body for the sake of hipcc's s_waitcnt insertion, as in gemm8p.hip::tile_body:
// it tags the LDS-DMA instructions with alias scopes and the ds_reads with "does not alias them"; untagged, every ds_read that follows
// an LDS-DMA in program order is preceded by `s_waitcnt vmcnt(0)`, i.e. the ring would be drained before every tile.)
template <typename T16, int KS, int NBH>
__device__ __forceinline__ void skinny_body(const SkArgs& p, char* smem, const bf16* __restrict__ Ag, const bf16* __restrict__ Bg, bf16* __restrict__ Cg) {
    constexpr int SP = ((KS * 4 + 7) / 8) * 8;            // 16-byte slots per LDS row (row pitch a multiple of 128 B: bank-aligned rows)
    constexpr int PITCH = SP * 16;
    constexpr int STAGE = TR * PITCH;
    constexpr int IPW = TR * SP / 64 / 4;                 // LDS-DMA instructions per wave per tile
    static_assert((TR * SP) % 256 == 0, "whole DMA instructions per wave");
    constexpr int NCOL = 32 * NBH;                        // columns of the (padded) output tile
    constexpr int OUT_BYTES = TR * NCOL * 2;              // staged output tile
    char* ring = smem;
    char* stage_out = smem + NST * STAGE;                 // two output buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave & 1, half = wave >> 1;
    const int c16 = lane & 15, q = lane >> 4;

    // ---- B fragments of this wave's columns, all of K: registers for the whole launch (rows >= N / k >= K read as zero)
    bf16x8_t fb[NBH][KS];
    {
        const long bbytes = (long)p.N * p.ldb * 2;
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bg, 0, (int)min(bbytes, 0x7fffffffL), 0x00020000);
#pragma unroll
        for (int cb = 0; cb < NBH; ++cb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int n = (half * NBH + cb) * 16 + c16, k = ks * 32 + q * 8;
                const uint32_t off = (n < p.N && k < p.K) ? (uint32_t)((n * p.ldb + k) * 2) : 0x80000000u;
                fb[cb][ks] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rB, off, 0, 0));
            }
    }
    // bias of this lane's columns (cb: columns (half * NBH + cb) * 16 + 4 q .. + 3); zeros without one
    float4 bv[NBH];
#pragma unroll
    for (int cb = 0; cb < NBH; ++cb) {
        const int n = (half * NBH + cb) * 16 + 4 * q;
        bv[cb] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- DMA geometry of a tile: instruction j of this wave fills slots [(wave * IPW + j) * 64, + 64) of the stage image; slot i = row
    // i / SP, physical 16-byte slot i % SP, which holds logical slot (phys ^ (row & 7)) of its 8-slot group (XOR within the group:
    // the 16 rows of a fragment read then spread over the banks).  Slots past K read out of range -> zeros.
    uint32_t aoff[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int i = (wave * IPW + j) * 64 + lane;
        const int row = i / SP, ph = i % SP;
        const int lg = (ph & ~7) | ((ph & 7) ^ (row & 7));
        aoff[j] = (lg * 8 < p.K) ? (uint32_t)((row * p.lda + lg * 8) * 2) : 0x80000000u;
    }
    auto issue = [&](int tile, int st) {
        const int m0 = tile * TR;
        const long ab = (long)min(TR, p.M - m0) * p.lda * 2;       // rows past M: out of range -> zeros
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(Ag + (size_t)m0 * p.lda), 0, (int)ab, 0x00020000);
        char* dst = ring + st * STAGE + wave * IPW * 1024;
#pragma unroll
        for (int j = 0; j < IPW; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(dst + j * 1024), 16, aoff[j], 0, 0, 0);
    };
    // output: vector v of the staged tile = row v / VPR, 16-byte column group v % VPR; this thread moves vectors tid, tid + 256, ...
    constexpr int VPR = NCOL / 8;
    constexpr int NSTORE = (TR * VPR + THREADS - 1) / THREADS;      // store instructions per thread and tile
    const int vN = p.N / 8;                               // real 16-byte column groups per row
    auto store_tile = [&](int tile, int buf) {
        const int m0 = tile * TR;
        const long cbts = (long)min(TR, p.M - m0) * p.ldc * 2;
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(Cg + (size_t)m0 * p.ldc), 0, (int)cbts, 0x00020000);
        const char* src = stage_out + buf * OUT_BYTES;
#pragma unroll
        for (int h = 0; h < NSTORE; ++h) {
            const int v = tid + h * THREADS;
            const int row = v / VPR, cg = v % VPR;
            const bool in = v < TR * VPR && cg < vN;
            const u32x4_t val = *reinterpret_cast<const u32x4_t*>(src + (in ? v : 0) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(val, rC, in ? (uint32_t)((row * p.ldc + cg * 8) * 2) : 0x80000000u, 0, 0);
        }
    };

    const int G = gridDim.x;
    const int n_my = (p.tiles - (int)blockIdx.x + G - 1) / G;
    if (n_my <= 0) return;
    auto tile_of = [&](int i) { return (int)blockIdx.x + i * G; };
    issue(tile_of(0), 0);
    if (n_my > 1) issue(tile_of(1), 1);
    // fragment read address of this lane within a stage: row rb * 16 + c16, logical slot ks * 4 + q
    const int frow = rb * 16 + c16;
    for (int i = 0; i < n_my; ++i) {
        // outstanding VMEM of this wave, oldest first: DMA(i) [, the 2 stores of tile i - 2's... already waited] , DMA(i + 1), stores(i - 1)?
        // Order of issue per iteration: stores(i - 1), DMA(i + 2).  At this point: DMA(i), DMA(i + 1) and -- issued between them --
        // stores(i - 2) are outstanding at most; everything up to DMA(i) must have landed: at most IPW (DMA i + 1) may stay in flight.
        if (i + 1 < n_my) vm_wait<IPW>();
        else vm_wait<0>();
        bar_lds();                             // tile i has landed for every wave; everyone is done with tile i - 1 (ring + staging)
        if (i >= 1) store_tile(tile_of(i - 1), (i - 1) & 1);
        if (i + 2 < n_my) issue(tile_of(i + 2), (i + 2) % NST);
        const char* st = ring + (i % NST) * STAGE + frow * PITCH;
        f32x4_t acc[NBH];
#pragma unroll
        for (int cb = 0; cb < NBH; ++cb) acc[cb] = f32x4_t{bv[cb].x, bv[cb].y, bv[cb].z, bv[cb].w};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int lg = ks * 4 + q;
            const int ph = (lg & ~7) | ((lg & 7) ^ (frow & 7));
            const bf16x8_t fa = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(st + ph * 16));
#pragma unroll
            for (int cb = 0; cb < NBH; ++cb) acc[cb] = h16<T16>::mma16(fb[cb][ks], fa, acc[cb]);
        }
        // acc[cb][r] = C[row frow][column (half * NBH + cb) * 16 + 4 q + r] -> staging (row-major [TR][NCOL])
        char* so = stage_out + (i & 1) * OUT_BYTES + frow * (NCOL * 2);
#pragma unroll
        for (int cb = 0; cb < NBH; ++cb) {
            const int col = (half * NBH + cb) * 16 + 4 * q;
            *reinterpret_cast<uint2*>(so + col * 2) = make_uint2(h16<T16>::pack2(acc[cb][0], acc[cb][1]), h16<T16>::pack2(acc[cb][2], acc[cb][3]));
        }
    }
    bar_lds();
    store_tile(tile_of(n_my - 1), (n_my - 1) & 1);
}

template <typename T16, int KS, int NBH>
__global__ __launch_bounds__(THREADS, 2) void gemm_skinny_kernel(SkArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    skinny_body<T16, KS, NBH>(p, smem, p.A, p.B, p.C);
}

template <typename T16, int KS, int NBH>
int launch_skinny(const SkArgs& a, hipStream_t s) {
    constexpr int SP = ((KS * 4 + 7) / 8) * 8;
    constexpr int LDS = NST * TR * SP * 16 + 2 * TR * (32 * NBH) * 2;
    static const int n_cu = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<T16, KS, NBH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const int grid = a.tiles < 2 * n_cu ? a.tiles : 2 * n_cu;
    hipLaunchKernelGGL((gemm_skinny_kernel<T16, KS, NBH>), dim3(grid), dim3(THREADS), LDS, s, a);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

template <typename T16, int NBH>
int dispatch_ks(const SkArgs& a, int ks, hipStream_t s) {
    if constexpr (NBH > 4) {      // wide outputs (N <= 288: the stage-1 q|k|v projection) only with short K: 9 column blocks x 3 k-steps of B = 108 registers
        if (ks <= 2) return launch_skinny<T16, 2, NBH>(a, s);
        if (ks == 3) return launch_skinny<T16, 3, NBH>(a, s);
        return G8_NOT_TAKEN;
    }
    switch (ks) {
        case 1: case 2: return launch_skinny<T16, 2, NBH>(a, s);
        case 3: return launch_skinny<T16, 3, NBH>(a, s);
        case 4: return launch_skinny<T16, 4, NBH>(a, s);
        case 5: case 6: return launch_skinny<T16, 6, NBH>(a, s);
        case 7: case 8: return launch_skinny<T16, 8, NBH>(a, s);
        case 9: return launch_skinny<T16, 9, NBH>(a, s);
        case 10: case 11: case 12: return launch_skinny<T16, 12, NBH>(a, s);
        default: return G8_NOT_TAKEN;
    }
}
}  // namespace

// tuning key "gemm_skinny": 0 = automatic (default), 1 = never
int g_skinny_mode = -1;

// Plain products and products with a bias (no activation / accumulation / second output): what the Swin engine asks of its narrow GEMMs.
int gemm_skinny_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    if (g_skinny_mode < 0) {
        const char* e = getenv("MOREC_GEMM_SKINNY");
        g_skinny_mode = e ? atoi(e) : 0;
    }
    if (g_skinny_mode == 1) return G8_NOT_TAKEN;
    if (!is_h16(d->in_dtype) || d->out_dtype != d->in_dtype) return G8_NOT_TAKEN;
    if (a.aux_out || a.dact_in || a.colsum || d->act != MOREC_ACT_NONE || d->dact != MOREC_ACT_NONE || a.accumulate != 0 || d->split_k > 1 ||
        d->alpha != 1.0f)
        return G8_NOT_TAKEN;
    if (d->N > 288 || d->N < 64 || d->N % 8 || d->K % 8 || d->K > 384 || d->M < 8192) return G8_NOT_TAKEN;
    if (d->N > 128 && d->K > 96) return G8_NOT_TAKEN;
    if (a.bias && (d->N % 4 || (reinterpret_cast<uintptr_t>(a.bias) & 15u))) return G8_NOT_TAKEN;
    if (d->lda % 8 || d->ldb % 8 || d->ldc % 8) return G8_NOT_TAKEN;
    if ((long)TR * d->lda * 2 >= 0x7fffffffL) return G8_NOT_TAKEN;
    SkArgs k;
    k.A = reinterpret_cast<const bf16*>(a.A); k.B = reinterpret_cast<const bf16*>(a.B); k.C = reinterpret_cast<bf16*>(a.C);
    k.bias = a.bias;
    k.M = d->M; k.N = d->N; k.K = d->K; k.lda = d->lda; k.ldb = d->ldb; k.ldc = d->ldc; k.tiles = (d->M + TR - 1) / TR;
    const int ks = (d->K + 31) / 32;
    const int nbh = d->N <= 96 ? 3 : (d->N <= 128 ? 4 : 9);
    if (d->in_dtype == MOREC_F16) return nbh == 3 ? dispatch_ks<f16, 3>(k, ks, s) : nbh == 4 ? dispatch_ks<f16, 4>(k, ks, s) : dispatch_ks<f16, 9>(k, ks, s);
    return nbh == 3 ? dispatch_ks<bf16, 3>(k, ks, s) : nbh == 4 ? dispatch_ks<bf16, 4>(k, ks, s) : dispatch_ks<bf16, 9>(k, ks, s);
}
