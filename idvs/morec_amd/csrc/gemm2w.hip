// gemm2w.hip -- NT GEMM for the EPILOGUE-HEAVY products (FFN-up + GELU + act', x act' + d(b1); the Swin stages' short-K products): a
// 256 x 128 output tile per FOUR-wave workgroup and TWO such workgroups resident per CU.
//
// Why a second tile kernel.  gemm8p.hip keeps one eight-wave workgroup per CU: its waves hold 128 accumulators + fragments in 224-240 of their
// 256 registers and the workgroup owns the whole LDS, so while a tile's epilogue runs (GELU + derivative + two 16-bit outputs: 24 k of a
// K = 768 tile's 57 k cycles; profiles/r03_gemm8p_stamps.txt) the CU's matrix pipes are idle, and VALU work placed into the main loop's read
// segments only lengthens them (profiles/r05_shadow_valu.txt, r06_shadow_setprio.txt: the two wave rows are coupled by a barrier per phase).
// Here the two workgroups of a CU are NOT coupled: each SIMD hosts one wave of either; while one workgroup is in its epilogue (VALU, LDS
// transposes, stores) or waits for a fragment read, the other one's MFMAs have the matrix pipe.  Nothing has to be scheduled by hand for that.
//
// Tile: 256 (m) x 128 (n), four waves as 2 (m) x 2 (n), 128 x 64 outputs per wave on v_mfma_f32_32x32x16 (the accumulator layout, the LDS
// row image -- 128-byte rows, 16-byte slots XOR-swizzled with (row >> 1) & 7 -- the fragment reads and the four-quadrant MFMA order are
// gemm8p's: gemm8p_core.hpp).  LDS: ONE K-tile (64 elements of K: A 32 KiB + B 16 KiB) + four 4-KiB epilogue slices = 64 KiB per workgroup.
// The K-tile is refilled REGION BY REGION, in place: a region (A-first = the 64 rows per wave row read in phase 0, B-first, B-second,
// A-second) is dead as soon as every wave's reads of it have returned, i.e. behind the barrier of the phase that reads it, and K-tile t + 1's
// rows are requested into it right there -- three phases (768 MFMA cycles of this wave + the other workgroup's share of the pipe) before they
// are read.  A phase:   { fragment reads | counted vmcnt: this wave's pieces of the region the NEXT phase reads | lgkmcnt(0) } s_barrier
//                       { LDS-DMA refill of the region just read | 8 MFMA }
// One barrier per phase, four per K-tile; the DMA queue is never drained inside a tile (waits vmcnt(4) / (6) / - / (6)).  K-tiles past the
// end are "refilled" with out-of-range offsets (the DMA writes zeros, no memory traffic), so the counts are the same for every K-tile.
// One tile per workgroup (a plain grid, XCD-contiguous tile order in column groups): a workgroup's fill and drain are covered by its CU
// partner the same way its epilogue is.
#include <stdlib.h>
#include <string.h>
#include "gemm8p_core.hpp"
#include "gemm_args.hpp"

namespace {
using namespace g8;
constexpr int WM = 256, WN = 128;
constexpr int A_BYTES = WM * KB;                 // 32 KiB
constexpr int B_BYTES = WN * KB;                 // 16 KiB
constexpr int RING2 = A_BYTES + B_BYTES;         // one K-tile
constexpr int SLICE2 = 4096;
constexpr int LDS2 = RING2 + 4 * SLICE2;         // 64 KiB: two workgroups per CU
constexpr int THREADS2 = 256;

__device__ __forceinline__ void lgkm_wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct Ctx2 {
    __amdgpu_buffer_rsrc_t ra, rb;   // descriptors of the tile's A / B panels (extent = the rows that exist: rows past them read as zeros)
    uint32_t va[2], vb[2];           // per-lane byte offsets of a piece (8 rows) of parity 0 / 1: row (lane >> 3), swizzled slot
    int sa, sb;                      // wave-uniform byte offset of the wave's first A-first / B-first row
    int pa, pb;                      // bytes of 8 rows of A / B
    int da, db;                      // LDS offset of the wave's first A-first / B-first piece
    int aoff, boff;                  // LDS offset of the wave's first A row / B row
    int loff[4];                     // per-lane fragment offset of MFMA k-step ks
};

__device__ __forceinline__ void make_ctx2(Ctx2& c, int tid, const bf16* At, const bf16* Bt, int rows_a, int rows_b, int lda, int ldb) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int ra0 = wr * 128 + wc * 32;          // this wave's 32 rows of A-first (A-second: + 64)
    const int rb0 = wr * 64 + wc * 16;           // this wave's 16 rows of B-first (B-second: + 32)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {             // (row >> 1) & 7 of row = base (a multiple of 16) + 8 j + (lane >> 3) is 4 (j & 1) + (lane >> 4)
        const int slot = (lane & 7) ^ (4 * jp + (lane >> 4));
        c.va[jp] = (uint32_t)(lane >> 3) * (uint32_t)(lda * 2) + slot * 16;
        c.vb[jp] = (uint32_t)(lane >> 3) * (uint32_t)(ldb * 2) + slot * 16;
    }
    c.sa = ra0 * lda * 2; c.sb = rb0 * ldb * 2;
    c.pa = 8 * lda * 2; c.pb = 8 * ldb * 2;
    const long abytes = (long)min(WM, rows_a) * lda * 2, bbytes = (long)min(WN, rows_b) * ldb * 2;
    c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)At, 0, (int)min(abytes, 0x7fffffffL), 0x00020000);
    c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bt, 0, (int)min(bbytes, 0x7fffffffL), 0x00020000);
    c.da = ra0 * KB; c.db = A_BYTES + rb0 * KB;
    c.aoff = wr * 128 * KB; c.boff = A_BYTES + wc * 64 * KB;
    const int r5 = lane & 31, fr = (r5 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) c.loff[ks] = r5 * KB + (((2 * ks + (lane >> 5)) ^ fr) << 4);
}

// NP pieces (8 rows x 128 B each) of one region: global rows from `srow` bytes on, K offset kb bytes, into LDS at dst.  mz = 0x80000000
// for a K-tile past the end: the offset leaves the descriptor's range and the DMA writes zeros.
template <int NP>
__device__ __forceinline__ void dma_region(__amdgpu_buffer_rsrc_t rs, const uint32_t (&v)[2], uint32_t mz, int srow, int piece, int kb, char* dst) {
#pragma unroll
    for (int j = 0; j < NP; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + j * 1024), 16, v[j & 1] | mz, srow + j * piece + kb, 0, 0);
}

// One K-tile; kbn = byte offset of K-tile t + 1 within a row, mz = 0x80000000 when there is none.
template <typename T16, bool ZERO>
__device__ __forceinline__ void ktile2(char* smem, const Ctx2& c, int kbn, uint32_t mz, f32x16_t (&acc)[4][2]) {
    uint4 fa[2][4], fb0[4], fb1[4];
    // ---- phase 0: B-first + A-first fragments; then their regions take K-tile t + 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb0[ks] = lds16(smem + c.boff + c.loff[ks]);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[mi][ks] = lds16(smem + c.aoff + c.loff[ks] + mi * (32 * KB));
    pin();
    vm_wait<4>();          // B-second of THIS K-tile (this wave's two pieces), read next phase
    lgkm_wait0();
    bar();
    dma_region<4>(c.ra, c.va, mz, c.sa, c.pa, kbn, smem + c.da);
    dma_region<2>(c.rb, c.vb, mz, c.sb, c.pb, kbn, smem + c.db);
    pin();
    mfma_quadrant<T16, 0, 0, ZERO>(acc, fa, fb0);
    // ---- phase 1: B-second fragments
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb1[ks] = lds16(smem + c.boff + c.loff[ks] + 32 * KB);
    pin();
    vm_wait<6>();          // A-second of this K-tile
    lgkm_wait0();
    bar();
    dma_region<2>(c.rb, c.vb, mz, c.sb + 4 * c.pb, c.pb, kbn, smem + c.db + 32 * KB);
    pin();
    mfma_quadrant<T16, 0, 1, ZERO>(acc, fa, fb1);
    // ---- phase 2: A-second fragments
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[mi][ks] = lds16(smem + c.aoff + c.loff[ks] + (64 + mi * 32) * KB);
    pin();
    lgkm_wait0();
    bar();
    dma_region<4>(c.ra, c.va, mz, c.sa + 8 * c.pa, c.pa, kbn, smem + c.da + 64 * KB);
    pin();
    mfma_quadrant<T16, 2, 1, ZERO>(acc, fa, fb1);
    // ---- phase 3: nothing to read; A-first + B-first of K-tile t + 1 must have landed behind this barrier
    pin();
    vm_wait<6>();
    bar();
    mfma_quadrant<T16, 2, 0, ZERO>(acc, fa, fb0);
}

// ACT: 0 = linear (+ bias), 1 = GELU (aux_out optional: pre-activation or act'), 5 = x dact_in (the stored act'); CS: fused column sums
template <typename TI, typename TO, int ACT, bool CS>
__device__ __forceinline__ void tile2_body(const GemmArgs& p, char* smem, const bf16* __restrict__ At, const bf16* __restrict__ Bt, int m0, int n0,
                                           int tm_idx) {
    constexpr int ES = (int)sizeof(TO);
    static_assert(ES == 2, "gemm2w: 16-bit outputs");
    constexpr int EPV = 8, PC = 64, NG = 8;
    struct Vecs { u32x4_t q[4]; };
    f32x16_t acc[4][2];
    float bias_l;
    const int nk = p.K / KE;
    {
        int tid_m = threadIdx.x;
        asm volatile("" : "+v"(tid_m));
        {
            const float* bp = p.bias ? p.bias : reinterpret_cast<const float*>(p.B);
            bias_l = bp[min(n0 + ((tid_m >> 6) & 1) * 64 + (tid_m & 63), p.N - 1)];
        }
        Ctx2 c;
        make_ctx2(c, tid_m, At, Bt, p.M - m0, p.N - n0, p.lda, p.ldb);
        // prologue: K-tile 0, in the order the main loop's counted waits assume: [A-first, B-first] [B-second] [A-second]
        dma_region<4>(c.ra, c.va, 0u, c.sa, c.pa, 0, smem + c.da);
        dma_region<2>(c.rb, c.vb, 0u, c.sb, c.pb, 0, smem + c.db);
        dma_region<2>(c.rb, c.vb, 0u, c.sb + 4 * c.pb, c.pb, 0, smem + c.db + 32 * KB);
        dma_region<4>(c.ra, c.va, 0u, c.sa + 8 * c.pa, c.pa, 0, smem + c.da + 64 * KB);
        pin();
        vm_wait<6>();
        bar();
        ktile2<TI, true>(smem, c, KB, nk > 1 ? 0u : 0x80000000u, acc);
        for (int t = 1; t < nk; ++t) ktile2<TI, false>(smem, c, (t + 1) * KB, t + 1 < nk ? 0u : 0x80000000u, acc);
        pin();
        vm_wait<0>();      // the zero "refills" of the K-tile past the end: drained before the workgroup may leave
    }
    // ---- epilogue (wave-private; gemm8p.hip's, for a 2 x 2 wave grid): every 32-row block through the wave's own 4-KiB slice so that all
    // global traffic is 16-byte lanes along rows
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane = tid_e & 63, r5 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid_e >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    char* ws = smem + RING2 + wave * SLICE2;
    TO* C = reinterpret_cast<TO*>(p.C);
    TO* aux = reinterpret_cast<TO*>(p.aux_out);
    const TO* din = reinterpret_cast<const TO*>(p.dact_in);
    auto wfence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int rs_row = lane >> 3, rs_slot = lane & 7;
    const int rs_off = rs_row * 128 + ((rs_slot ^ (rs_row & 7)) << 4);
    auto cell = [&](int q) { return reinterpret_cast<TO*>(ws + r5 * 128 + ((q ^ (r5 & 7)) << 4) + 8 * h); };
    const int mw = m0 + wr * 128, nw = n0 + wc * 64;
    const int m_end = min(p.M, m0 + WM);
    const long tile_bytes = (long)min(WM, p.M - m0) * p.ldc * ES;
    const int ext = (int)min(tile_bytes, 0x7fffffffL);
    const size_t tile_off = (size_t)m0 * p.ldc;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(C + tile_off), 0, ext, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)(((ACT == 5) ? din : C) + tile_off), 0, ext, 0x00020000);
    const uint32_t lo = (nw + rs_slot * EPV) < p.N ? (uint32_t)((rs_row * p.ldc + nw + rs_slot * EPV) * ES) : 0x80000000u;
    auto row_term = [&](int Mi, int i) { return (uint32_t)((wr * 128 + Mi * 32 + 8 * i) * p.ldc * ES); };
    auto rows_store = [&](const __amdgpu_buffer_rsrc_t& rs, int Mi) {
        u32x4_t q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const u32x4_t*>(ws + rs_off + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q[i], rs, lo + row_term(Mi, i), 0, 0);
    };
    auto rows_fetch = [&](const __amdgpu_buffer_rsrc_t& rs, int Mi) {
        Vecs r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r.q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lo + row_term(Mi, i), 0, 0);
        return r;
    };
    float4 bv[2][4];
    if constexpr (ACT != 5) {
        reinterpret_cast<float*>(ws)[lane] = p.bias ? bias_l : 0.f;
        wfence();
#pragma unroll
        for (int Ni = 0; Ni < 2; ++Ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[Ni][g] = *reinterpret_cast<const float4*>(ws + (Ni * 32 + g * 8 + 4 * h) * 4);
        wfence();
    } else {
#pragma unroll
        for (int Ni = 0; Ni < 2; ++Ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[Ni][g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    Vecs dq[2] = {Vecs(), Vecs()};
    if constexpr (ACT == 5) {
        dq[0] = rows_fetch(rD, 0);
        dq[1] = rows_fetch(rD, 1);
    }
    pin();
    [[maybe_unused]] float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int Mi = 0; Mi < 4; ++Mi) {
        [[maybe_unused]] const bool row_ok = (mw + Mi * 32 + r5) < m_end;
        pin();
        if constexpr (ACT == 5) {
            const int nx = Mi + 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(ws + rs_off + i * 1024) = dq[Mi & 1].q[i];
            if (nx <= 3) dq[Mi & 1] = rows_fetch(rD, nx);
            wfence();
        }
        if (ACT == 1 && aux) {      // (wave-uniform) second output first: the pre-activation, or act'(pre) with aux_deriv
            uint2 park[NG];
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const int Ni = q / 4, g = q % 4;
                const float4 b = bv[Ni][g];
                float v[4], d[4];
                v[0] = fmaf(acc[Mi][Ni][4 * g + 0], p.alpha, b.x);
                v[1] = fmaf(acc[Mi][Ni][4 * g + 1], p.alpha, b.y);
                v[2] = fmaf(acc[Mi][Ni][4 * g + 2], p.alpha, b.z);
                v[3] = fmaf(acc[Mi][Ni][4 * g + 3], p.alpha, b.w);
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = v[r];
                if (p.aux_deriv) gelu4_with_deriv(v, d);
                else gelu4(v);
                io<TO>::store4(cell(q), d);
                park[q] = make_uint2(h16<TI>::pack2(v[0], v[1]), h16<TI>::pack2(v[2], v[3]));
                if (q & 1) pin();
            }
            wfence();
            rows_store(__builtin_amdgcn_make_buffer_rsrc((void*)(aux + tile_off), 0, ext, 0x00020000), Mi);
            wfence();
#pragma unroll
            for (int q = 0; q < NG; ++q) *reinterpret_cast<uint2*>(cell(q)) = park[q];
            wfence();
            rows_store(rC, Mi);
            wfence();
            continue;
        }
        [[maybe_unused]] uint32_t uraw[NG][2];
        if constexpr (ACT == 5) {
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const uint2 t = *reinterpret_cast<const uint2*>(cell(q));
                uraw[q][0] = t.x; uraw[q][1] = t.y;
            }
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int Ni = q / 4, g = q % 4;
            const float4 b = bv[Ni][g];
            float v[4];
            v[0] = fmaf(acc[Mi][Ni][4 * g + 0], p.alpha, b.x);
            v[1] = fmaf(acc[Mi][Ni][4 * g + 1], p.alpha, b.y);
            v[2] = fmaf(acc[Mi][Ni][4 * g + 2], p.alpha, b.z);
            v[3] = fmaf(acc[Mi][Ni][4 * g + 3], p.alpha, b.w);
            if constexpr (ACT == 1) {
                gelu4(v);
            } else if constexpr (ACT == 5) {
                v[0] *= h16<TI>::bits2f(uraw[q][0] & 0xffffu); v[1] *= h16<TI>::bits2f(uraw[q][0] >> 16);
                v[2] *= h16<TI>::bits2f(uraw[q][1] & 0xffffu); v[3] *= h16<TI>::bits2f(uraw[q][1] >> 16);
            }
            if constexpr (CS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = row_ok ? v[r] : 0.f;
            }
            io<TO>::store4(cell(q), v);
            if (ACT != 5 && (q & 1)) pin();
        }
        wfence();
        if constexpr (CS) {
            u32x4_t q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const u32x4_t*>(ws + rs_off + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    cs8[2 * w] += h16<TI>::bits2f(q[i][w] & 0xffffu);
                    cs8[2 * w + 1] += h16<TI>::bits2f(q[i][w] >> 16);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q[i], rC, lo + row_term(Mi, i), 0, 0);
        } else {
            rows_store(rC, Mi);
        }
        wfence();
    }
    if constexpr (CS) {         // one partial row per 128-row wave block; the launcher folds them
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cs8[j] += __shfl_xor(cs8[j], 8, 64);
            cs8[j] += __shfl_xor(cs8[j], 16, 64);
            cs8[j] += __shfl_xor(cs8[j], 32, 64);
        }
        if (rs_row == 0) {
            const int n = nw + rs_slot * 8;
            float* dst = p.colsum + (size_t)(tm_idx * 2 + wr) * p.N + n;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (n + j < p.N) dst[j] = cs8[j];
        }
    }
}

template <typename TI, typename TO, int ACT, bool CS>
__global__ __launch_bounds__(THREADS2, 2) void gemm2w_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = p.tiles_m * p.tiles_n;
    // tile order: XCD x works through a CONTIGUOUS run of order indices; order = column groups of p.ngroup N-tiles, row-major (m, n) inside a
    // group, so that the ~64 tiles an XCD has in flight share a few A panels and the B panels of one group (its 4-MiB L2)
    const int wg = xcd_remap((int)blockIdx.x, nwg);
    int tm, tn;
    if (p.ngroup <= 0 || p.ngroup >= p.tiles_n) {
        tm = wg / p.tiles_n;
        tn = wg % p.tiles_n;
    } else {
        const int per = p.ngroup * p.tiles_m, ng = (p.tiles_n + p.ngroup - 1) / p.ngroup;
        const int g = min(wg / per, ng - 1), r = wg - g * per;
        const int w = g == ng - 1 ? p.tiles_n - g * p.ngroup : p.ngroup;
        tm = r / w;
        tn = g * p.ngroup + r % w;
    }
    const int m0 = tm * WM, n0 = tn * WN;
    if (p.debug) {      // EXPERIMENT: stagger the two workgroups of a CU in the first round (p.debug = delay in units of 256 cycles; bit 30: by parity)
        const int local = (int)blockIdx.x >> 3;
        const bool second = (p.debug & (1 << 30)) ? (local & 1) : (local >= 32);
        if (local < 64 && second) {
            const long long until = (long long)__builtin_readcyclecounter() + (long long)(p.debug & 0xffff) * 256;
            while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
        }
    }
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    tile2_body<TI, TO, ACT, CS>(p, smem, A + (size_t)m0 * p.lda, B + (size_t)n0 * p.ldb, m0, n0, tm);
}

template <typename TI, int ACT, bool CS>
int launch2w(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    a.tiles_m = (d->M + WM - 1) / WM;
    a.tiles_n = (d->N + WN - 1) / WN;
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2w_kernel<TI, TI, ACT, CS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        return true;
    }();
    (void)attr;
    a.ngroup = a.tiles_n <= 8 ? 0 : 8;
    hipLaunchKernelGGL((gemm2w_kernel<TI, TI, ACT, CS>), dim3(a.tiles_m * a.tiles_n), dim3(THREADS2), LDS2, s, a);
    MOREC_CHECK_LAUNCH();
    if constexpr (CS) return colsum_f32_launch(a.colsum, a.colsum_dst, a.tiles_m * 2, d->N, s);
    return MOREC_OK;
}
}  // namespace

// tuning key "gemm2w" / MOREC_GEMM2W: 0 = automatic (the epilogue-heavy products), 1 = never, 2 = every eligible product
int g_mode2w = -1;
int gemm2w_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    if (g_mode2w < 0) {
        const char* e = getenv("MOREC_GEMM2W");
        g_mode2w = e ? atoi(e) : 0;
    }
    if (g_mode2w == 1) return G8_NOT_TAKEN;
    if (!is_h16(d->in_dtype) || d->out_dtype != d->in_dtype || a.accumulate != 0 || !a.vec_store || d->split_k > 1) return G8_NOT_TAKEN;
    if (d->K % KE || d->K < KE || d->N % 8 || d->M < 1) return G8_NOT_TAKEN;
    if (d->dact != MOREC_ACT_NONE && d->dact != MOREC_DACT_MUL) return G8_NOT_TAKEN;
    if (d->dact == MOREC_DACT_MUL && a.bias) return G8_NOT_TAKEN;
    if (d->act == MOREC_ACT_RELU || (a.aux_out && d->act != MOREC_ACT_GELU)) return G8_NOT_TAKEN;
    if (a.colsum && (d->dact != MOREC_DACT_MUL || d->M < 128)) return G8_NOT_TAKEN;
    const long tiles = (long)((d->M + WM - 1) / WM) * ((d->N + WN - 1) / WN);
    if (tiles < 64) return G8_NOT_TAKEN;
    const int mode = d->dact == MOREC_DACT_MUL ? 5 : d->act == MOREC_ACT_GELU ? 1 : 0;
    if (g_mode2w != 2) {
        // Automatic: where this kernel measured FASTER than gemm8p on random operands (profiles/r06_gemm2w_shapes.txt).  Both kernels hold the part's
        // power limit on random fp16 data (same binary on zero operands: gemm2w 1.43 PFLOP/s, gemm8p 1.18 at N = 3072, K = 768, against 0.93 / 0.96
        // on N(0, 0.5) operands: profiles/r06_gemm_power_probe.txt), so what is left to win is work, not schedule:
        //  (a) plain products whose N fills 256-wide tiles badly (N = 384: three 128-wide tiles instead of two 256-wide ones, a quarter of whose
        //      MFMAs, LDS and DMA traffic are padding): Swin-T stage 3, 1.11 - 1.30 x;
        //  (b) x act' + column sums over fewer rows (Swin-T stage 4, Swin-B stage 3): 1.04 - 1.07 x.
        const long waste = (long)((d->N + 255) / 256) * 256 - d->N;
        const bool narrow = mode == 0 && !a.colsum && waste * 4 >= d->N && d->M >= 8192 && d->K <= 2048;
        const bool dmul = mode == 5 && d->M >= 16384 && ((d->M <= 40000 && d->K <= 768) || (d->M <= 70000 && d->K <= 512));
        if (!narrow && !dmul) return G8_NOT_TAKEN;
    }
    {
        static const int stagger = [] { const char* e = getenv("MOREC_GEMM2W_STAGGER"); return e ? atoi(e) : 0; }();
        a.debug = stagger;
    }
    if (d->in_dtype == MOREC_F16) {
        if (mode == 5) return a.colsum ? launch2w<f16, 5, true>(d, a, s) : launch2w<f16, 5, false>(d, a, s);
        if (mode == 1) return launch2w<f16, 1, false>(d, a, s);
        return launch2w<f16, 0, false>(d, a, s);
    }
    if (mode == 5) return a.colsum ? launch2w<bf16, 5, true>(d, a, s) : launch2w<bf16, 5, false>(d, a, s);
    if (mode == 1) return launch2w<bf16, 1, false>(d, a, s);
    return launch2w<bf16, 0, false>(d, a, s);
}
