// gemm_skinny_wide.hip -- the streaming NT GEMM of gemm_skinny.hip for outputs that are 4 C wide over millions of rows, short K (= C):
// the Swin stage-1 MLP (HF modeling_swin.py SwinIntermediate / SwinOutput via V/model/encoders.py:30-31), C = 96 / 128.
//
//   single:  C[M, N] = GELU(A[M, K] . B[N, K]^T + bias)                                     (fc1 + GELU, no second output)
//   dual:    C[M, N] = (A[M, K] . B[N, K]^T) * act'(A2[M, K] . B2[N, K]^T + bias2),  colsum[N] += sum_m C[m, :]
//
// `dual` is the backward of  g = GELU(x W1^T + b1) -> fc2  WITHOUT a stored act' tensor: dU = (dY W2) * GELU'(x W1^T + b1) recomputes
// the pre-activation from x (K = C wide) instead of reading a 4 C wide act'(pre) that the forward would have had to write: at
// C = 96 the forward launch writes 768 instead of 1536 bytes per row and this launch reads 384 instead of 960 (autograd of
// nn.GELU re-reads the saved pre-activation; SwinIntermediate keeps it alive for exactly that).
//
// Eight waves, each owning N / 8 columns of BOTH 16-row blocks of a 32-row tile: the B operand(s) of a wave's columns live in
// registers as MFMA fragments for the whole launch (held once per workgroup, not once per row block), A tiles stream through an
// LDS ring by LDS-DMA with counted vmcnt, the output tile leaves through double-buffered LDS staging as 16-byte row pieces.
// One workgroup per CU (dual: 144 fragment + 48 accumulator registers at C = 128), persistent grid.
#include <stdlib.h>
#include "gemm_core.hpp"
#include "gemm_args.hpp"

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int TR = 32;                 // rows per tile
constexpr int THREADS = 512;
constexpr int NW = 8;                  // waves = column slices

struct WArgs {
    const bf16* A;
    const bf16* B;
    bf16* C;
    const float* bias;     // single: added to the product; dual: added to the SECOND product (the pre-activation)
    const bf16* A2;
    const bf16* B2;
    float* colsum_ws;      // dual: fp32 [grid][N] per-workgroup column sums of C, or null
    int M, N, K, lda, ldb, ldc, lda2, ldb2, tiles;
};

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// see gemm_skinny.hip::bar_lds
__device__ __forceinline__ void bar_lds() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// KS = MFMA k-steps (32 elements) covering K; NBW = 16-column blocks per wave (padded N = 128 NBW); ACT: 0 none, 1 GELU (single) / GELU' (dual)
// (A / B / C as `__restrict__` parameters of an inlined body: gemm_skinny.hip explains why)
// NSL = column slices of 128 NBW columns, one per workgroup (N = 768 with two weights to keep: two workgroups of the same XCD share a row tile)
template <typename T16, int KS, int NBW, int ACT, bool DUAL, int NST, int NSL>
__device__ __forceinline__ void wide_body(const WArgs& p, char* smem, const bf16* __restrict__ Ag, const bf16* __restrict__ Bg,
                                          const bf16* __restrict__ A2g, const bf16* __restrict__ B2g, bf16* __restrict__ Cg) {
    constexpr int SP = ((KS * 4 + 7) / 8) * 8;            // 16-byte slots per LDS row
    constexpr int PITCH = SP * 16;
    constexpr int IPW = (TR * SP + 511) / 512;            // LDS-DMA instructions per wave, tile and operand
    constexpr int STAGE = IPW * 512 * 16;                 // stage image incl. the slots past TR x SP (filled with zeros, never read)
    constexpr int NOP = DUAL ? 2 : 1;
    constexpr int DPT = IPW * NOP;                        // DMA instructions per wave and tile
    constexpr int NCOL = 128 * NBW;
    constexpr int OUT_BYTES = TR * NCOL * 2;
    char* ring = smem;                                    // [NOP][NST][STAGE]
    char* stage_out = smem + NOP * NST * STAGE;           // two output buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q = lane >> 4;
    const int nw0 = wave * NBW * 16;                      // first column of this wave within the slice
    // workgroups b and b + 8 run on the same XCD (round-robin dispatch): they take the two column slices of the same row tiles, so the second
    // read of an A tile is an L2 hit
    const int bid = blockIdx.x;
    const int slice = NSL == 1 ? 0 : (bid >> 3) % NSL;
    const int worker = NSL == 1 ? bid : (bid / (8 * NSL)) * 8 + (bid & 7);
    const int nc0 = slice * NCOL;                         // first column of this workgroup

    // ---- B fragments of this wave's columns (rows >= N / k >= K read as zero)
    bf16x8_t fb[NOP][NBW][KS];
#pragma unroll
    for (int o = 0; o < NOP; ++o) {
        const bf16* Bo = o ? B2g : Bg;
        const int ld = o ? p.ldb2 : p.ldb;
        const long bbytes = (long)p.N * ld * 2;
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bo, 0, (int)min(bbytes, 0x7fffffffL), 0x00020000);
#pragma unroll
        for (int cb = 0; cb < NBW; ++cb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int n = nc0 + nw0 + cb * 16 + c16, k = ks * 32 + q * 8;
                const uint32_t off = (n < p.N && k < p.K) ? (uint32_t)((n * ld + k) * 2) : 0x80000000u;
                fb[o][cb][ks] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rB, off, 0, 0));
            }
    }
    float4 bv[NBW];
#pragma unroll
    for (int cb = 0; cb < NBW; ++cb) {
        const int n = nc0 + nw0 + cb * 16 + 4 * q;
        bv[cb] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- DMA geometry (as gemm_skinny.hip: slot i = row i / SP, physical slot i % SP holding logical slot phys ^ (row & 7) of its group)
    uint32_t aoff[NOP][IPW];
#pragma unroll
    for (int o = 0; o < NOP; ++o)
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int i = (wave * IPW + j) * 64 + lane;
            const int row = i / SP, ph = i % SP;
            const int lg = (ph & ~7) | ((ph & 7) ^ (row & 7));
            aoff[o][j] = (row < TR && lg * 8 < p.K) ? (uint32_t)((row * (o ? p.lda2 : p.lda) + lg * 8) * 2) : 0x80000000u;
        }
    auto issue = [&](int tile, int st) {
        const int m0 = tile * TR;
#pragma unroll
        for (int o = 0; o < NOP; ++o) {
            const int ld = o ? p.lda2 : p.lda;
            const long ab = (long)min(TR, p.M - m0) * ld * 2;       // rows past M: out of range -> zeros
            const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)((o ? A2g : Ag) + (size_t)m0 * ld), 0, (int)ab, 0x00020000);
            char* dst = ring + (o * NST + st) * STAGE + wave * IPW * 1024;
#pragma unroll
            for (int j = 0; j < IPW; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(dst + j * 1024), 16, aoff[o][j], 0, 0, 0);
        }
    };
    constexpr int VPR = NCOL / 8;
    constexpr int NSTORE = (TR * VPR + THREADS - 1) / THREADS;
    const int vN = p.N / 8;
    auto store_tile = [&](int tile, int buf) {
        const int m0 = tile * TR;
        const long cbts = (long)min(TR, p.M - m0) * p.ldc * 2;
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(Cg + (size_t)m0 * p.ldc), 0, (int)cbts, 0x00020000);
        const char* src = stage_out + buf * OUT_BYTES;
#pragma unroll
        for (int h = 0; h < NSTORE; ++h) {
            const int v = tid + h * THREADS;
            const int row = v / VPR, cg = v % VPR;
            const bool in = v < TR * VPR && slice * VPR + cg < vN;
            const u32x4_t val = *reinterpret_cast<const u32x4_t*>(src + (in ? v : 0) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(val, rC, in ? (uint32_t)((row * p.ldc + nc0 + cg * 8) * 2) : 0x80000000u, 0, 0);
        }
        store_b128_guard();                    // MFMAs follow (common.hpp)
    };

    const int G = gridDim.x / NSL;                        // workers
    const int n_my = worker < p.tiles ? (p.tiles - worker + G - 1) / G : 0;
    float cs[NBW][4];
#pragma unroll
    for (int cb = 0; cb < NBW; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[cb][r] = 0.f;
    auto tile_of = [&](int i) { return worker + i * G; };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < n_my) issue(tile_of(t), t);
    for (int i = 0; i < n_my; ++i) {
        // VMEM issued by this wave after DMA(i): stores and the DMAs of min(NST - 2, n_my - 1 - i) later tiles.  Loads return in order,
        // so "at most that many DMA instructions outstanding" implies DMA(i) has landed whatever the stores do.
        const int newer = min(NST - 2, n_my - 1 - i);
        if (newer >= 2) vm_wait<2 * DPT>();
        else if (newer == 1) vm_wait<DPT>();
        else vm_wait<0>();
        bar_lds();                             // tile i has landed for every wave; everyone is done with tile i - 1 (ring + staging)
        if (i >= 1) store_tile(tile_of(i - 1), (i - 1) & 1);
        if (i + NST - 1 < n_my) issue(tile_of(i + NST - 1), (i + NST - 1) % NST);
        const char* st0 = ring + (i % NST) * STAGE;
        // acc[cb][r] = C[row rb * 16 + c16][column nw0 + cb * 16 + 4 q + r]; one 16-row block at a time (dual: 2 x NBW accumulators live)
        char* so = stage_out + (i & 1) * OUT_BYTES;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int frow = rb * 16 + c16;
            f32x4_t acc[NBW], acu[DUAL ? NBW : 1];
#pragma unroll
            for (int cb = 0; cb < NBW; ++cb) {
                const f32x4_t b4 = f32x4_t{bv[cb].x, bv[cb].y, bv[cb].z, bv[cb].w};
                if constexpr (DUAL) {
                    acc[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    acu[cb] = b4;
                } else {
                    acc[cb] = b4;
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int lg = ks * 4 + q;
                const int ph = (lg & ~7) | ((lg & 7) ^ (frow & 7));
                const bf16x8_t fa = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(st0 + frow * PITCH + ph * 16));
#pragma unroll
                for (int cb = 0; cb < NBW; ++cb) acc[cb] = h16<T16>::mma16(fb[0][cb][ks], fa, acc[cb]);
                if constexpr (DUAL) {
                    const bf16x8_t fa2 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(st0 + NST * STAGE + frow * PITCH + ph * 16));
#pragma unroll
                    for (int cb = 0; cb < NBW; ++cb) acu[cb] = h16<T16>::mma16(fb[NOP - 1][cb][ks], fa2, acu[cb]);
                }
            }
#pragma unroll
            for (int cb = 0; cb < NBW; ++cb) {
                float v[4] = {acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]};
                if constexpr (DUAL) {
                    const float u[4] = {acu[cb][0], acu[cb][1], acu[cb][2], acu[cb][3]};
                    if constexpr (ACT == 1) dgelu4_mul(v, u);
#pragma unroll
                    for (int r = 0; r < 4; ++r) cs[cb][r] += v[r];
                } else {
                    if constexpr (ACT == 1) gelu4(v);
                }
                const int col = nw0 + cb * 16 + 4 * q;
                *reinterpret_cast<uint2*>(so + frow * (NCOL * 2) + col * 2) = make_uint2(h16<T16>::pack2(v[0], v[1]), h16<T16>::pack2(v[2], v[3]));
            }
        }
    }
    if (n_my > 0) {
        bar_lds();
        store_tile(tile_of(n_my - 1), (n_my - 1) & 1);
    }
    if constexpr (DUAL) {
        if (p.colsum_ws) {      // this workgroup's column sums: fold the 16 row lanes, one plain store per column (a workgroup without tiles writes zeros)
#pragma unroll
            for (int cb = 0; cb < NBW; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = cs[cb][r];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
                    const int n = nc0 + nw0 + cb * 16 + 4 * q + r;
                    if (c16 == 0 && n < p.N) p.colsum_ws[(size_t)worker * p.N + n] = s;
                }
        }
    }
}

template <typename T16, int KS, int NBW, int ACT, bool DUAL, int NST, int OCC, int NSL>
__global__ __launch_bounds__(THREADS, OCC) void gemm_skinny_wide_kernel(WArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wide_body<T16, KS, NBW, ACT, DUAL, NST, NSL>(p, smem, p.A, p.B, p.A2, p.B2, p.C);
}

int wide_grid(int tiles, int occ) {
    static const int n_cu = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return tiles < occ * n_cu ? tiles : occ * n_cu;
}

// OCC workgroups per CU: 2 where 128 registers and half the LDS do (one product at C = 96)
template <typename T16, int KS, int NBW, int ACT, bool DUAL, int NST, int OCC, int NSL = 1>
int launch_wide(WArgs& a, float* colsum_out, hipStream_t s) {
    constexpr int SP = ((KS * 4 + 7) / 8) * 8;
    constexpr int IPW = (TR * SP + 511) / 512;
    constexpr int LDS = (DUAL ? 2 : 1) * NST * IPW * 512 * 16 + 2 * TR * (128 * NBW) * 2;
    static_assert(LDS * OCC <= 160 * 1024, "LDS of one CU");
    // NSL > 1: whole groups of 8 NSL workgroups (8 workers x NSL slices on 8 XCDs); workers without a tile only write their zero column sums
    const int workers = NSL == 1 ? wide_grid(a.tiles, OCC) : ((wide_grid(a.tiles, OCC) / NSL + 7) / 8) * 8;
    const int grid = workers * NSL;
    static const int once = [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_wide_kernel<T16, KS, NBW, ACT, DUAL, NST, OCC, NSL>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }();
    (void)once;
    hipLaunchKernelGGL((gemm_skinny_wide_kernel<T16, KS, NBW, ACT, DUAL, NST, OCC, NSL>), dim3(grid), dim3(THREADS), LDS, s, a);
    MOREC_CHECK_LAUNCH();
    if (DUAL && colsum_out) return colsum_f32_launch(a.colsum_ws, colsum_out, workers, a.N, s);
    return MOREC_OK;
}

// the (K, N) classes that exist: K <= 96 with N <= 384, K <= 128 with N <= 512 (Swin-T / -S and Swin-B / -L stage 1), K <= 192 with N <= 768 (Swin-T / -S stage 2)
template <typename T16, int ACT, bool DUAL>
int dispatch_wide(WArgs& a, float* colsum_out, hipStream_t s) {
    const int ks = (a.K + 31) / 32, nbw = (a.N + 127) / 128;
    if (ks <= 3 && nbw <= 3) return launch_wide<T16, 3, 3, ACT, DUAL, DUAL ? 4 : 3, DUAL ? 1 : 2>(a, colsum_out, s);
    if (ks <= 4 && nbw <= 4) return launch_wide<T16, 4, 4, ACT, DUAL, 4, 1>(a, colsum_out, s);
    if (ks <= 6 && nbw <= 6) {      // C = 192 (Swin-T / -S stage 2): one product keeps [768, 192] in 144 registers per lane; two need two column slices
        if constexpr (DUAL) return launch_wide<T16, 6, 3, ACT, true, 3, 1, 2>(a, colsum_out, s);
        else return launch_wide<T16, 6, 6, ACT, false, 3, 1>(a, colsum_out, s);
    }
    return G8_NOT_TAKEN;
}

bool wide_shape_ok(int M, int N, int K, int lda, int ldb, int ldc) {
    if (N <= 288 || N > 768 || N % 8 || K % 8 || K > 192 || K < 32 || M < 8192) return false;
    if (lda % 8 || ldb % 8 || ldc % 8) return false;
    if ((long)TR * lda * 2 >= 0x7fffffffL || (long)TR * ldc * 2 >= 0x7fffffffL) return false;
    return true;
}
}  // namespace

// GELU(A . B^T + bias) without a second output: called from morec_gemm_nt after gemm_skinny_try_launch
int gemm_skinny_wide_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    if (g_skinny_mode == 1 || g_skinny_mode == 2) return G8_NOT_TAKEN;
    if (!is_h16(d->in_dtype) || d->out_dtype != d->in_dtype) return G8_NOT_TAKEN;
    // only the GELU product: a plain / bias-only product of these widths is faster on the tile kernel (measured: Swin-B stage-1 q|k|v 306 vs 334 us);
    // what the streaming form buys is the epilogue-heavy, write-heavy launch
    if (a.aux_out || a.dact_in || a.colsum || d->act != MOREC_ACT_GELU || d->dact != MOREC_ACT_NONE || a.accumulate != 0 || d->split_k > 1 ||
        d->alpha != 1.0f)
        return G8_NOT_TAKEN;
    if (!wide_shape_ok(d->M, d->N, d->K, d->lda, d->ldb, d->ldc)) return G8_NOT_TAKEN;
    if (a.bias && (d->N % 4 || (reinterpret_cast<uintptr_t>(a.bias) & 15u))) return G8_NOT_TAKEN;
    WArgs k{};
    k.A = reinterpret_cast<const bf16*>(a.A); k.B = reinterpret_cast<const bf16*>(a.B); k.C = reinterpret_cast<bf16*>(a.C);
    k.bias = a.bias;
    k.M = d->M; k.N = d->N; k.K = d->K; k.lda = d->lda; k.ldb = d->ldb; k.ldc = d->ldc; k.tiles = (d->M + TR - 1) / TR;
    return d->in_dtype == MOREC_F16 ? dispatch_wide<f16, 1, false>(k, nullptr, s) : dispatch_wide<bf16, 1, false>(k, nullptr, s);
}

namespace {
bool dual_class_exists(int M, int N, int K, int dtype) {
    if (!is_h16(dtype) || !wide_shape_ok(M, N, K, K, K, N) || N % 4) return false;
    const int ks = (K + 31) / 32, nbw = (N + 127) / 128;
    return (ks <= 3 && nbw <= 3) || (ks <= 4 && nbw <= 4) || (ks <= 6 && nbw <= 6);
}
}  // namespace

// "Should the caller drop the act' tensor for this shape": the classes where forward + backward together are faster that way.  K <= 128 (stage 1:
// Swin-T 1286 + 1025 -> 620 + 840 ... 980 us, Swin-B 640 + 501 -> 443 + 554); at K = 192 (Swin-T stage 2) the two column slices make the backward
// launch slower than the act' read it replaces (500 + 418 -> 368 + 530 us), so the answer is no although morec_mlp_dact_recompute accepts the shape.
extern "C" int morec_mlp_dact_recompute_supported(int M, int N, int K, int dtype) {
    if (g_skinny_mode < 0) {
        const char* e = getenv("MOREC_GEMM_SKINNY");
        g_skinny_mode = e ? atoi(e) : 0;
    }
    if (g_skinny_mode == 1 || g_skinny_mode == 2) return 0;
    return (K <= 128 && dual_class_exists(M, N, K, dtype)) ? 1 : 0;
}

extern "C" size_t morec_mlp_dact_recompute_workspace_bytes(int N) {
    return (size_t)1024 * (size_t)(N > 0 ? N : 0) * sizeof(float);
}

extern "C" int morec_mlp_dact_recompute(const void* dY, const void* W2t, const void* X, const void* W1, const float* b1, void* dU,
                                        float* colsum_out, float* workspace, int M, int N, int K, int dtype, void* stream) {
    if (!dY || !W2t || !X || !W1 || !dU || M <= 0 || N <= 0 || K <= 0) return MOREC_E_ARG;
    if (colsum_out && !workspace) return MOREC_E_ARG;
    if (!is_h16(dtype)) return MOREC_E_DTYPE;
    if (!dual_class_exists(M, N, K, dtype)) return MOREC_E_UNSUPPORTED;
    if (b1 && (reinterpret_cast<uintptr_t>(b1) & 15u)) return MOREC_E_ALIGN;
    if (!aligned16(dY) || !aligned16(W2t) || !aligned16(X) || !aligned16(W1) || !aligned16(dU)) return MOREC_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    WArgs k{};
    k.A = reinterpret_cast<const bf16*>(dY); k.B = reinterpret_cast<const bf16*>(W2t); k.C = reinterpret_cast<bf16*>(dU);
    k.A2 = reinterpret_cast<const bf16*>(X); k.B2 = reinterpret_cast<const bf16*>(W1);
    k.bias = b1;
    k.M = M; k.N = N; k.K = K; k.lda = K; k.ldb = K; k.ldc = N; k.lda2 = K; k.ldb2 = K; k.tiles = (M + TR - 1) / TR;
    k.colsum_ws = colsum_out ? workspace : nullptr;
    const int rc = dtype == MOREC_F16 ? dispatch_wide<f16, 1, true>(k, colsum_out, s) : dispatch_wide<bf16, 1, true>(k, colsum_out, s);
    return rc == G8_NOT_TAKEN ? MOREC_E_UNSUPPORTED : rc;
}
