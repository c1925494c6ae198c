// gemm8p.hip -- the encoder GEMMs' main loop: 256 x 256 output tile, K advanced 64 elements per K-tile, eight waves as
// 2 (m) x 4 (n), 128 x 64 outputs per wave on v_mfma_f32_32x32x16_bf16, a K-tile split in FOUR phases
//      { ds_read a register sub-tile | issue one 16-KiB LDS-DMA half-tile | counted vmcnt }  s_barrier
//      { 8 MFMA 32x32x16 = one 64 x 32 quadrant of the wave tile over the whole K-tile }     s_barrier
// with the two wave rows running ONE BARRIER APART: while waves 0-3 (one per SIMD) are in their MFMA segment, waves 4-7 (their
// SIMD partners) are in their read / DMA segment and vice versa, so every SIMD's matrix pipe always has a wave feeding it and
// the LDS / DMA issue of one wave is covered by its partner's MFMAs.  The DMA queue is never drained inside the loop: every
// phase waits `vmcnt(8)` (four half-tiles = one whole K-tile in flight per CU across the barriers).
//
// Replaces (for bf16, large problems) the two-buffer one-barrier loop of gemm_core.hpp, which measured 41.7 % MFMA-pipe
// busy on the K = 3072 shape (profiles/r01_gemm_pmc_sq.txt): a stage's LDS-DMA fill took as long as its MFMAs and a third
// of every stage was spent at the barrier that drains it.  Same reference arithmetic (include/morec_hip.h: morec_gemm_nt).
//
// LDS: two 64-KiB buffers (K-tile t lives in buffer t & 1), each [A: 256 rows x 128 B][B: 256 rows x 128 B], row = tile row.
// 16-byte slot s of row r holds logical slot s ^ ((r >> 1) & 7): with 32-row MFMA fragments (lane -> row lane & 31) the 16
// lanes of every ds_read_b128 service group then cover all 64 banks exactly once.  The LDS-DMA writes lane-linearly, so the
// permutation is applied to the per-lane SOURCE address and again to the read address (an involution; destination linear).
//
// Half-tiles are cut by CONSUMPTION ORDER, not by wave: "A-first" = rows every wave reads in phase 0 (its first 64 of 128),
// "A-second" = the rows read in phase 2, "B-first" / "B-second" = the first / second 32 of every wave's 64 columns (read in
// phases 0 / 1).  A region is dead two phases after its last read (one phase for the reading wave row + one because the
// other wave row runs a barrier behind), which is when it is refilled:
//      phase 0: B-second of K-tile t+1   phase 1: A-second of t+1   phase 2: A-first of t+2   phase 3: B-first of t+2
// Every refill is issued 5-6 phases before its first read; a wave may read a region one phase after the counted wait that
// retires it (its own pieces) + the barrier behind that wait (everybody else's).
#include <stdlib.h>
#include <string.h>
#include "gemm_core.hpp"
#include "gemm_args.hpp"

namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int TM = 256, TN = 256, KE = 64;      // tile; K elements per K-tile
constexpr int KB = 128;                          // bytes of K per row per K-tile
constexpr int OP_BYTES = 256 * KB;               // one operand of one K-tile: 32 KiB
constexpr int BUF_BYTES = 2 * OP_BYTES;          // 64 KiB
constexpr int LDS_BYTES = 2 * BUF_BYTES;         // 128 KiB
constexpr int THREADS = 512;

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void bar() {
    pin();
    __builtin_amdgcn_s_barrier();
    pin();
}

typedef __attribute__((address_space(8))) void* rsrc_t;   // 128-bit buffer descriptor (4 SGPRs)

// The two 1-KiB pieces (8 rows each) this wave contributes to a half-tile.  buffer_load ... offen lds: descriptor (SGPRs) +
// per-lane 32-bit byte offset (loop-invariant VGPR) + wave-uniform K offset (SGPR): no per-lane 64-bit pointers to keep alive
// or to advance, which is what made the flat global_load_lds form of this loop spill.
__device__ __forceinline__ void dma2(__amdgpu_buffer_rsrc_t rs, uint32_t o0, uint32_t o1, int kbyte, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, o0, kbyte, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + 1024), 16, o1, kbyte, 0, 0);
}

struct Ctx {
    __amdgpu_buffer_rsrc_t ra, rb;               // descriptors of A / B, based at the tile's first row
    uint32_t a1[2], a2[2], b1[2], b2[2];         // per-lane byte offsets of the wave's two pieces of each half-tile
    int dA1, dA2, dB1, dB2;                      // wave-uniform LDS offsets (within a buffer) of those pieces
    int aoff, boff;                              // LDS offsets (within a buffer) of the wave's first A row / first B row
    int loff[4];                                 // per-lane fragment offset of MFMA k-step ks: row (lane & 31), swizzled slot
};

__device__ __forceinline__ uint4 lds16(const char* p) { return *reinterpret_cast<const uint4*>(p); }

template <int M0, int NQ>
__device__ __forceinline__ void mfma_quadrant(f32x16_t (&acc)[4][2], const uint4 (&fa)[2][4], const uint4 (&fb)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)   // operand-swapped: the lane ends up owning ONE m and runs of 4 consecutive n
            acc[M0 + mi][NQ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[ks]),
                                                                       __builtin_bit_cast(bf16x8_t, fa[mi][ks]), acc[M0 + mi][NQ], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

// vmcnt left in flight after a phase's issue: 8 in the steady state (the refills of the last four phases); the last two
// K-tiles issue fewer, so fewer may be left.  REM = K-tiles after this one, capped at 2 (compile time: no branches in the loop).
template <int REM, int W1, int W0>
__device__ __forceinline__ void vm_wait_tail() {
    vm_wait<(REM >= 2 ? 8 : (REM == 1 ? W1 : W0))>();
}

// One K-tile out of the buffer at byte offset `cb` (0 or BUF_BYTES); kb = byte offset of this K-tile within a row.
template <int REM>
__device__ __forceinline__ void ktile(char* smem, const Ctx& c, int cb, int kb, f32x16_t (&acc)[4][2]) {
    char* cur = smem + cb;
    char* oth = smem + (cb ^ BUF_BYTES);
    uint4 fa[2][4], fb0[4], fb1[4];
    int ada[4], adb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        ada[ks] = cb + c.aoff + c.loff[ks];
        adb[ks] = cb + OP_BYTES + c.boff + c.loff[ks];
    }
    // ---- phase 0: B-first + A-first fragments; refill B-second of t+1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb0[ks] = lds16(smem + adb[ks]);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[mi][ks] = lds16(smem + ada[ks] + mi * (32 * KB));
    if constexpr (REM >= 1) dma2(c.rb, c.b2[0], c.b2[1], kb + KB, oth + OP_BYTES + c.dB2);
    pin();
    vm_wait_tail<REM, 8, 2>();
    bar();
    mfma_quadrant<0, 0>(acc, fa, fb0);
    bar();
    // ---- phase 1: B-second fragments; refill A-second of t+1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb1[ks] = lds16(smem + adb[ks] + 32 * KB);
    if constexpr (REM >= 1) dma2(c.ra, c.a2[0], c.a2[1], kb + KB, oth + c.dA2);
    pin();
    vm_wait_tail<REM, 8, 0>();
    bar();
    mfma_quadrant<0, 1>(acc, fa, fb1);
    bar();
    // ---- phase 2: A-second fragments; refill A-first of t+2 (this buffer)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[mi][ks] = lds16(smem + ada[ks] + (64 + mi * 32) * KB);
    if constexpr (REM >= 2) dma2(c.ra, c.a1[0], c.a1[1], kb + 2 * KB, cur + c.dA1);
    pin();
    vm_wait_tail<REM, 6, 0>();
    bar();
    mfma_quadrant<2, 1>(acc, fa, fb1);
    bar();
    // ---- phase 3: nothing to read (B-first is still in registers); refill B-first of t+2
    if constexpr (REM >= 2) dma2(c.rb, c.b1[0], c.b1[1], kb + 2 * KB, cur + OP_BYTES + c.dB1);
    pin();
    vm_wait_tail<REM, 4, 0>();
    bar();
    mfma_quadrant<2, 0>(acc, fa, fb0);
    bar();
}

// acc += A[m0 .. m0+255, :] . B[n0 .. n0+255, :]^T over K (K % 64 == 0, K >= 128).  Rows past M / N are clamped to the last valid
// row (their products land in accumulator rows / columns that are never stored).  On return every DMA has landed and
// every wave has passed the last barrier: LDS is free.
__device__ __forceinline__ void mainloop8p(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb,
                                           int m0, int n0, char* smem, f32x16_t (&acc)[4][2]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nk = K / KE;
    Ctx c;
    {   // DMA geometry: piece j of a half-tile = 8 rows; lane -> row (lane >> 3) of the piece, physical slot lane & 7
        const int ra = wr * 128 + wc * 16;                       // this wave's 16 rows of A-first (A-second: + 64)
        const int rb = (wave >> 1) * 64 + (wave & 1) * 16;       // this wave's 16 rows of B-first (B-second: + 32)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = j * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((rl >> 1) & 7);       // (row >> 1) & 7 with row = 16-aligned base + rl
            c.a1[j] = (uint32_t)min(ra + rl, M - 1 - m0) * (uint32_t)(lda * 2) + slot * 16;
            c.a2[j] = (uint32_t)min(ra + 64 + rl, M - 1 - m0) * (uint32_t)(lda * 2) + slot * 16;
            c.b1[j] = (uint32_t)min(rb + rl, N - 1 - n0) * (uint32_t)(ldb * 2) + slot * 16;
            c.b2[j] = (uint32_t)min(rb + 32 + rl, N - 1 - n0) * (uint32_t)(ldb * 2) + slot * 16;
        }
        // descriptors: raw (stride 0), extent = the rows of this tile that exist (every offset above stays inside it)
        const long abytes = (long)min(256, M - m0) * lda * 2, bbytes = (long)min(256, N - n0) * ldb * 2;
        c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, (int)min(abytes, 0x7fffffffL), 0x00020000);
        c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * ldb), 0, (int)min(bbytes, 0x7fffffffL), 0x00020000);
        c.dA1 = ra * KB; c.dA2 = (ra + 64) * KB; c.dB1 = rb * KB; c.dB2 = (rb + 32) * KB;
        c.aoff = wr * 128 * KB;
        c.boff = wc * 64 * KB;
        const int r5 = lane & 31, fr = (r5 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c.loff[ks] = r5 * KB + (((2 * ks + (lane >> 5)) ^ fr) << 4);
    }
    // prologue: K-tile 0 entirely, the first halves of K-tile 1
    dma2(c.ra, c.a1[0], c.a1[1], 0, smem + c.dA1);
    dma2(c.rb, c.b1[0], c.b1[1], 0, smem + OP_BYTES + c.dB1);
    dma2(c.rb, c.b2[0], c.b2[1], 0, smem + OP_BYTES + c.dB2);
    dma2(c.ra, c.a2[0], c.a2[1], 0, smem + c.dA2);
    dma2(c.ra, c.a1[0], c.a1[1], KB, smem + BUF_BYTES + c.dA1);
    dma2(c.rb, c.b1[0], c.b1[1], KB, smem + BUF_BYTES + OP_BYTES + c.dB1);
    pin();
    vm_wait<8>();            // A-first, B-first of K-tile 0 (this wave's pieces)
    bar();                   // ... everybody's
    if (wr == 1) bar();      // waves 4-7 run one barrier behind waves 0-3 from here on

    int cb = 0;
    int t = 0;
    for (; t < nk - 2; ++t) {
        ktile<2>(smem, c, cb, t * KB, acc);
        cb ^= BUF_BYTES;
    }
    ktile<1>(smem, c, cb, t * KB, acc);
    ktile<0>(smem, c, cb ^ BUF_BYTES, (t + 1) * KB, acc);
    if (wr == 0) bar();      // waves 0-3 catch the trailing barrier of waves 4-7
}

// -----------------------------------------------------------------------------------------------------------------------
// Epilogue: wave-private.  acc[Mi][Ni][4 g + r] = C[m0 + wr*128 + Mi*32 + (lane & 31)][n0 + wc*64 + Ni*32 + 8 g + 4 (lane >> 5) + r].
// Every 32 x 64 block goes through the wave's own LDS slice so that all global traffic (the stores, and the loads of the
// activation-derivative operand) is 16-byte lanes along rows: full 128-byte lines per row per instruction.
// -----------------------------------------------------------------------------------------------------------------------
template <typename TO, int ACT, bool CS>
__global__ __launch_bounds__(THREADS) void gemm8p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tm = wg / p.tiles_n, tn = wg % p.tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    mainloop8p(reinterpret_cast<const bf16*>(p.A), reinterpret_cast<const bf16*>(p.B), p.M, p.N, p.K, p.lda, p.ldb, m0, n0, smem, acc);

    constexpr int ES = (int)sizeof(TO);
    constexpr int EPV = 16 / ES;                 // elements per 16-byte vector
    constexpr int PITCH = 64 * ES + 16;          // LDS pitch of a staged 64-column row
    constexpr int SLICE = 32 * PITCH;
    constexpr int VPR = 64 * ES / 16;            // 16-byte vectors per row: 8 (bf16) / 16 (f32)
    constexpr int NV = 32 * VPR / 64;            // vectors per lane per block: 4 / 8
    static_assert(8 * SLICE <= LDS_BYTES, "wave slices do not fit");
    const int lane = threadIdx.x & 63, r5 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    char* ws = smem + wave * SLICE;
    TO* C = reinterpret_cast<TO*>(p.C);
    TO* aux = reinterpret_cast<TO*>(p.aux_out);
    const TO* din = reinterpret_cast<const TO*>(p.dact_in);
    const int mw = m0 + wr * 128, nw = n0 + wc * 64;
    auto wfence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // row-wise (coalesced) side of the slice: vector i of this lane = row (lane + 64 i) / VPR, 16-byte column (lane + 64 i) % VPR
    auto rows_store = [&](TO* dst, int Mi) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i, row = v / VPR, cv = v % VPR;
            const int m = mw + Mi * 32 + row, n = nw + cv * EPV;
            const uint4 q = *reinterpret_cast<const uint4*>(ws + row * PITCH + cv * 16);
            if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(dst + (size_t)m * p.ldc + n) = q;
        }
    };
    struct Vecs { uint4 q[NV]; };
    auto rows_fetch = [&](const TO* src, int Mi) {     // clamped, unguarded: one memory round trip
        Vecs r;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i, row = v / VPR, cv = v % VPR;
            const int m = min(mw + Mi * 32 + row, p.M - 1), n = min(nw + cv * EPV, p.N - EPV);
            r.q[i] = *reinterpret_cast<const uint4*>(src + (size_t)m * p.ldc + n);
        }
        return r;
    };
    auto rows_put = [&](const Vecs r) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i, row = v / VPR, cv = v % VPR;
            *reinterpret_cast<uint4*>(ws + row * PITCH + cv * 16) = r.q[i];
        }
    };
    // accumulator side of the slice: this lane's 4-element group (Ni, g)
    auto cell = [&](int Ni, int g) { return reinterpret_cast<TO*>(ws + r5 * PITCH + (Ni * 32 + g * 8 + 4 * h) * ES); };

    float csum = 0.f;
    Vecs dq0, dq1, dq2, dq3;   // the activation-derivative operand of the four 32-row blocks (fetched one block ahead)
    if constexpr (ACT == 3 || ACT == 4) dq0 = rows_fetch(din, 0);
#pragma unroll
    for (int Mi = 0; Mi < 4; ++Mi) {
        const bool row_ok = (mw + Mi * 32 + r5) < p.M;
        float u[2][4][4];
        if constexpr (ACT == 3 || ACT == 4) {
            rows_put(Mi == 0 ? dq0 : Mi == 1 ? dq1 : Mi == 2 ? dq2 : dq3);
            if (Mi == 0) dq1 = rows_fetch(din, 1);
            if (Mi == 1) dq2 = rows_fetch(din, 2);
            if (Mi == 2) dq3 = rows_fetch(din, 3);
            wfence();
#pragma unroll
            for (int Ni = 0; Ni < 2; ++Ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) io<TO>::load4(cell(Ni, g), u[Ni][g]);
            wfence();
        }
        float vv[2][4][4];
#pragma unroll
        for (int Ni = 0; Ni < 2; ++Ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nw + Ni * 32 + g * 8 + 4 * h;
                const bool ok = row_ok && n < p.N;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[Mi][Ni][4 * g + r] * p.alpha;
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + min(n, p.N - 4));
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (aux) {      // pre-activation out first
                    float pre[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre[r] = ok ? v[r] : 0.f;
                    io<TO>::store4(cell(Ni, g), pre);
                }
                if constexpr (ACT == 1) {
                    gelu4(v);
                } else if constexpr (ACT == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if constexpr (ACT == 3) {
                    dgelu4_mul(v, u[Ni][g]);
                } else if constexpr (ACT == 4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (u[Ni][g][r] > 0.f) ? v[r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[Ni][g][r] = ok ? v[r] : 0.f;
            }
        if (aux) {
            wfence();
            rows_store(aux, Mi);
            wfence();
        }
#pragma unroll
        for (int Ni = 0; Ni < 2; ++Ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) io<TO>::store4(cell(Ni, g), vv[Ni][g]);
        wfence();
        if constexpr (CS) {     // column sums of the block as stored (rounded to TO); rows / columns past M / N hold zeros
#pragma unroll
            for (int r = 0; r < 32; ++r) csum += io<TO>::load1(reinterpret_cast<const TO*>(ws + r * PITCH) + lane);
        }
        rows_store(C, Mi);
        wfence();
    }
    if constexpr (CS) {         // one partial row per 128-row wave block; the launcher folds them
        const int n = nw + lane;
        if (n < p.N) p.colsum[(size_t)(tm * 2 + wr) * p.N + n] = csum;
    }
}

template <typename TO, int ACT, bool CS>
int launch8p(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    a.tiles_m = (d->M + TM - 1) / TM;
    a.tiles_n = (d->N + TN - 1) / TN;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<TO, ACT, CS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm8p_kernel<TO, ACT, CS>), dim3(a.tiles_m * a.tiles_n), dim3(THREADS), LDS_BYTES, s, a);
    MOREC_CHECK_LAUNCH();
    if constexpr (CS) return colsum_f32_launch(a.colsum, a.colsum_dst, a.tiles_m * 2, d->N, s);
    return MOREC_OK;
}
}  // namespace

// 0: automatic (eligible large problems), 1: never, 2: every eligible problem regardless of size
static int g_mode8p = -1;
extern "C" int morec_tuning_set(const char* key, int value) {
    if (!key) return MOREC_E_ARG;
    if (!strcmp(key, "gemm8p")) { g_mode8p = value; return MOREC_OK; }
    return MOREC_E_UNSUPPORTED;
}

int gemm8p_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    if (g_mode8p < 0) { const char* e = getenv("MOREC_GEMM8P"); g_mode8p = e ? atoi(e) : 0; }
    if (g_mode8p == 1) return G8_NOT_TAKEN;
    if (d->in_dtype != MOREC_BF16 || a.accumulate != 0 || !a.vec_store || d->split_k > 1) return G8_NOT_TAKEN;
    if (d->K % KE || d->K < 2 * KE || d->N < 64 || d->M < 1) return G8_NOT_TAKEN;
    const long tiles = (long)((d->M + TM - 1) / TM) * ((d->N + TN - 1) / TN);
    if (a.colsum && d->M < 128) return G8_NOT_TAKEN;      // partial-row workspace is sized per 64 rows
    // automatic: enough tiles to fill the 256 CUs, and at most 15 % of the tile columns past N
    if (g_mode8p != 2 && (tiles < 192 || ((d->N + TN - 1) / TN) * TN * 100L > d->N * 115L)) return G8_NOT_TAKEN;
    const int mode = d->dact == MOREC_ACT_GELU ? 3 : d->dact == MOREC_ACT_RELU ? 4 : d->act == MOREC_ACT_GELU ? 1
                     : d->act == MOREC_ACT_RELU ? 2 : 0;
    if (d->out_dtype == MOREC_F32) {
        if (mode != 0 || a.colsum) return G8_NOT_TAKEN;
        return launch8p<float, 0, false>(d, a, s);
    }
    if (d->out_dtype != MOREC_BF16) return G8_NOT_TAKEN;
    if (a.colsum) {
        if (mode == 3) return launch8p<bf16, 3, true>(d, a, s);
        if (mode == 4) return launch8p<bf16, 4, true>(d, a, s);
        return G8_NOT_TAKEN;
    }
    switch (mode) {
        case 1: return launch8p<bf16, 1, false>(d, a, s);
        case 2: return launch8p<bf16, 2, false>(d, a, s);
        case 3: return launch8p<bf16, 3, false>(d, a, s);
        case 4: return launch8p<bf16, 4, false>(d, a, s);
        default: return launch8p<bf16, 0, false>(d, a, s);
    }
}
