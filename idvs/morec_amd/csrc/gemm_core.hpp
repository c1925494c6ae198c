// gemm_core.hpp -- the one MFMA main loop every matmul-shaped kernel of this library shares.
//
// Tile: 128 x 128 output per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave = 4 x 4 MFMA
// 16x16 tiles), K advanced 128 bytes per stage through a double-buffered LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4), XOR-swizzled 16-byte slots, ds_read_b128 fragment reads.
//
// Both operands are K-contiguous ("NT": C[m,n] = sum_k A[m,k] B[n,k]).  One LDS image serves both
// precisions because a 16-byte fragment read is
//   bf16: 8 consecutive k  -> one v_mfma_f32_16x16x32_bf16 (lane l owns k = 8*(l>>4) .. +7)
//   f32 : 4 consecutive k  -> four v_mfma_f32_16x16x4_f32; the e-th takes element e of every lane,
//         i.e. k = 4*(l>>4)+e -- a permutation of k applied identically to A and B, which a dot
//         product does not see.
// The MFMA is issued "swapped" (B fragment as the instruction's A operand) so that each lane ends
// up with 4 CONSECUTIVE n for one m:  acc[mi][ni][r] = C[m0 + wm*64 + mi*16 + (lane&15)]
//                                                     [n0 + wn*64 + ni*16 + (lane>>4)*4 + r]
// which makes the epilogue's global stores 8/16-byte vectors along the contiguous dimension.
#pragma once
#include "common.hpp"

// Tile configuration: WM x WN waves, each owning MI x NI MFMA 16x16 tiles.
//   GemmTile<T, 2>                 128 x 128, 4 waves  (small problems, fused CE / eval tiles)
//   GemmTileCfg<T, 2, 4, 8, 4>     256 x 256, 8 waves  (the encoder GEMMs: twice the MFMA work per staged byte)
//   GemmTileCfg<T, 4, 2, 4, 4, 1>  256 x 128, 8 waves, 64-byte K stages: 48 KiB of LDS and 64 accumulator registers per wave, so TWO
//                                  workgroups fit a CU and one's main loop covers the other's tile turnover
//   GemmTileCfg<T, 2, 2, 2, 2, 2, 4>  64 x 64, 4 waves, a FOUR-stage LDS-DMA ring (gemm_mainloop_ring): the latency-class products of the
//                                  SASRec layers (2 560 rows: 80 tiles of 128 x 128 leave two thirds of the chip idle and expose one
//                                  global-load latency per K stage); 64 KiB of LDS, two to three workgroups per CU
template <typename T, int WM_, int WN_, int MI_, int NI_, int KSUB_ = 2, int NST_ = 2>
struct GemmTileCfg {
    static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_, NST = NST_;
    static constexpr int TM = WM * MI * 16, TN = WN * NI * 16, NWAVES = WM * WN, THREADS = 64 * NWAVES;
    static constexpr int KSUB = KSUB_;                 // MFMA k-steps (64 bytes of K each) per stage: 2 or 1
    static constexpr int KB = 64 * KSUB;               // bytes of K per row per stage
    static constexpr int KE = KB / (int)sizeof(T);     // elements of K per stage
    static constexpr int EPV = 16 / (int)sizeof(T);    // elements per 16-byte vector
    static constexpr int SLOTS = KB / 16;              // 16-byte slots per row (8)
    static constexpr int ROWS_PER_DMA = 64 / SLOTS;    // rows covered by one 1-KiB wave-level LDS-DMA (8)
    static constexpr int ADMA_PER_WAVE = TM / ROWS_PER_DMA / NWAVES;
    static constexpr int BDMA_PER_WAVE = TN / ROWS_PER_DMA / NWAVES;
    static constexpr int A_BYTES = TM * KB, B_BYTES = TN * KB;   // un-padded: an LDS-DMA image is lane-linear
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = NST * STAGE_BYTES;
    static_assert(NST >= 2 && NST <= 8, "stages of the LDS ring");
    static_assert(TM % (ROWS_PER_DMA * NWAVES) == 0 && TN % (ROWS_PER_DMA * NWAVES) == 0, "DMA split");
    static_assert(KSUB == 1 || KSUB == 2, "stage = one or two MFMA k-steps");
    // XOR key that spreads the 16 rows of a fragment read over the 16-byte slots of a row: 8 slots (128-byte rows) -> row & 7;
    // 4 slots (64-byte rows: rows r, r+4, r+8, r+12 start in the same 256-byte bank window) -> (row >> 2) & 3
    __host__ __device__ static constexpr int swz(int row) { return KSUB == 2 ? (row & 7) : ((row >> 2) & 3); }
};
template <typename T, int KSUB>
using GemmTile = GemmTileCfg<T, 2, 2, 4, 4>;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T>
__device__ __forceinline__ void mfma_step(f32x4_t& acc, const uint4& fa_n, const uint4& fb_m);
template <>
__device__ __forceinline__ void mfma_step<bf16>(f32x4_t& acc, const uint4& fn, const uint4& fm) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fn), __builtin_bit_cast(bf16x8_t, fm),
                                                  acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma_step<f16>(f32x4_t& acc, const uint4& fn, const uint4& fm) {
    acc = h16<f16>::mma16(__builtin_bit_cast(bf16x8_t, fn), __builtin_bit_cast(bf16x8_t, fm), acc);
}
template <>
__device__ __forceinline__ void mfma_step<float>(f32x4_t& acc, const uint4& fn, const uint4& fm) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fn.x), __uint_as_float(fm.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fn.y), __uint_as_float(fm.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fn.z), __uint_as_float(fm.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fn.w), __uint_as_float(fm.w), acc, 0, 0, 0);
}

// acc += A[m0.., kbeg:kend] . B[n0.., kbeg:kend]^T.   kbeg/kend in elements, multiples of EPV.
//
// Staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave-instruction, no
// VGPR round trip).  The DMA writes lane-linearly, so rows are exactly 128 B with no padding; the bank
// conflicts a 128-B pitch would give ds_read_b128 are removed by an XOR swizzle of the 16-byte slot index
// with (row & 7), applied to the per-lane SOURCE address on the way in and to the read address on the way out
// (the destination stays linear).  Rows past M / N are clamped to the last valid row (their products land in
// accumulator rows / columns that are never stored); a K tail that does not fill a stage goes through a
// register-staged, zero-filling path.  Double-buffered: the DMA of tile t+1 flies while tile t is multiplied.
template <typename G, typename T>
__device__ __forceinline__ void gemm_mainloop_cfg(const T* __restrict__ A, const T* __restrict__ B, int M, int N, int lda,
                                                  int ldb, int m0, int n0, int kbeg, int kend, char* smem,
                                                  f32x4_t (&acc)[G::MI][G::NI]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int nk = (kend - kbeg + G::KE - 1) / G::KE;
    if (nk <= 0) return;
    const bool tail = ((kend - kbeg) % G::KE) != 0;

    // per-lane source rows of this wave's DMA instructions (clamped) and the swizzled source slot
    const int drow = lane / G::SLOTS;                       // row within the 1-KiB piece
    const int pslot = lane % G::SLOTS;                      // physical slot this lane's 16 B land in
    const T* asrc[G::ADMA_PER_WAVE];
    const T* bsrc[G::BDMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G::ADMA_PER_WAVE; ++i) {
        const int row = (wave * G::ADMA_PER_WAVE + i) * G::ROWS_PER_DMA + drow;
        asrc[i] = A + (size_t)min(m0 + row, M - 1) * lda + (pslot ^ G::swz(row)) * G::EPV;
    }
#pragma unroll
    for (int i = 0; i < G::BDMA_PER_WAVE; ++i) {
        const int row = (wave * G::BDMA_PER_WAVE + i) * G::ROWS_PER_DMA + drow;
        bsrc[i] = B + (size_t)min(n0 + row, N - 1) * ldb + (pslot ^ G::swz(row)) * G::EPV;
    }
    auto dma = [&](int stage, int k0) {
        char* abase = smem + stage * G::STAGE_BYTES + (wave * G::ADMA_PER_WAVE) * 1024;
        char* bbase = smem + stage * G::STAGE_BYTES + G::A_BYTES + (wave * G::BDMA_PER_WAVE) * 1024;
#pragma unroll
        for (int i = 0; i < G::ADMA_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + k0), (lptr_t)(abase + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < G::BDMA_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i] + k0), (lptr_t)(bbase + i * 1024), 16, 0, 0);
    };
    // K tail: plain loads with zero fill, written to the same swizzled image
    auto stage_tail = [&](int stage, int k0) {
        char* base = smem + stage * G::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < (G::TM * G::SLOTS) / G::THREADS; ++i) {
            const int v = tid + G::THREADS * i;
            const int row = v / G::SLOTS, lslot = v % G::SLOTS;
            const int k = k0 + lslot * G::EPV, am = m0 + row;
            const uint4 ra = (k < kend && am < M) ? *reinterpret_cast<const uint4*>(A + (size_t)am * lda + k) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(base + row * G::KB + ((lslot ^ G::swz(row)) * 16)) = ra;
        }
#pragma unroll
        for (int i = 0; i < (G::TN * G::SLOTS) / G::THREADS; ++i) {
            const int v = tid + G::THREADS * i;
            const int row = v / G::SLOTS, lslot = v % G::SLOTS;
            const int k = k0 + lslot * G::EPV, bn = n0 + row;
            const uint4 rb = (k < kend && bn < N) ? *reinterpret_cast<const uint4*>(B + (size_t)bn * ldb + k) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(base + G::A_BYTES + row * G::KB + ((lslot ^ G::swz(row)) * 16)) = rb;
        }
    };
    auto stage = [&](int st, int t) {
        if (tail && t == nk - 1) stage_tail(st, kbeg + t * G::KE);
        else dma(st, kbeg + t * G::KE);
    };

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 15;                             // fragment row within a 16-row block (== row & 15)
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) stage(cur ^ 1, t + 1);
        const char* as = smem + cur * G::STAGE_BYTES + (wm * G::MI * 16 + frow) * G::KB;
        const char* bs = smem + cur * G::STAGE_BYTES + G::A_BYTES + (wn * G::NI * 16 + frow) * G::KB;
#pragma unroll
        for (int ks = 0; ks < G::KSUB; ++ks) {
            const int phys = ((ks * 4 + (lane >> 4)) ^ G::swz(frow)) * 16;
            uint4 fa[G::MI], fb[G::NI];
#pragma unroll
            for (int i = 0; i < G::MI; ++i) fa[i] = *reinterpret_cast<const uint4*>(as + i * 16 * G::KB + phys);
#pragma unroll
            for (int i = 0; i < G::NI; ++i) fb[i] = *reinterpret_cast<const uint4*>(bs + i * 16 * G::KB + phys);
#pragma unroll
            for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni) mfma_step<T>(acc[mi][ni], fb[ni], fa[mi]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}


template <int N>
__device__ __forceinline__ void gemm_vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The same product on an NST-deep LDS ring (G::NST > 2): the DMAs of stages t+1 .. t+NST-2 are in flight while stage t is multiplied,
// the wait is a COUNTED vmcnt (each stage is the same number of LDS-DMA instructions per wave, and they retire in order), one barrier per
// stage: after it every wave's pieces of stage t have landed and every wave has finished reading stage t-1, whose slot is refilled.
// For the short products (K = 512: eight stages) this exposes ONE global-load latency instead of one per stage.
template <typename G, typename T>
__device__ __forceinline__ void gemm_mainloop_ring(const T* __restrict__ A, const T* __restrict__ B, int M, int N, int lda,
                                                   int ldb, int m0, int n0, int kbeg, int kend, char* smem,
                                                   f32x4_t (&acc)[G::MI][G::NI]) {
    constexpr int NST = G::NST;
    constexpr int DPS = G::ADMA_PER_WAVE + G::BDMA_PER_WAVE;      // LDS-DMA instructions per wave per stage
    static_assert((NST - 2) * DPS <= 56, "vmcnt range");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int nk = (kend - kbeg + G::KE - 1) / G::KE;
    if (nk <= 0) return;
    const bool tail = ((kend - kbeg) % G::KE) != 0;
    const int last_dma = tail ? nk - 2 : nk - 1;            // last stage that is staged by DMA (the K tail goes through registers)

    const int drow = lane / G::SLOTS, pslot = lane % G::SLOTS;
    const T* asrc[G::ADMA_PER_WAVE];
    const T* bsrc[G::BDMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G::ADMA_PER_WAVE; ++i) {
        const int row = (wave * G::ADMA_PER_WAVE + i) * G::ROWS_PER_DMA + drow;
        asrc[i] = A + (size_t)min(m0 + row, M - 1) * lda + (pslot ^ G::swz(row)) * G::EPV;
    }
#pragma unroll
    for (int i = 0; i < G::BDMA_PER_WAVE; ++i) {
        const int row = (wave * G::BDMA_PER_WAVE + i) * G::ROWS_PER_DMA + drow;
        bsrc[i] = B + (size_t)min(n0 + row, N - 1) * ldb + (pslot ^ G::swz(row)) * G::EPV;
    }
    auto dma = [&](int slot, int k0) {
        char* abase = smem + slot * G::STAGE_BYTES + (wave * G::ADMA_PER_WAVE) * 1024;
        char* bbase = smem + slot * G::STAGE_BYTES + G::A_BYTES + (wave * G::BDMA_PER_WAVE) * 1024;
#pragma unroll
        for (int i = 0; i < G::ADMA_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + k0), (lptr_t)(abase + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < G::BDMA_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i] + k0), (lptr_t)(bbase + i * 1024), 16, 0, 0);
    };
    auto stage_tail = [&](int slot, int k0) {
        char* base = smem + slot * G::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < (G::TM * G::SLOTS) / G::THREADS; ++i) {
            const int v = tid + G::THREADS * i;
            const int row = v / G::SLOTS, lslot = v % G::SLOTS;
            const int k = k0 + lslot * G::EPV, am = m0 + row;
            const uint4 ra = (k < kend && am < M) ? *reinterpret_cast<const uint4*>(A + (size_t)am * lda + k) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(base + row * G::KB + ((lslot ^ G::swz(row)) * 16)) = ra;
        }
#pragma unroll
        for (int i = 0; i < (G::TN * G::SLOTS) / G::THREADS; ++i) {
            const int v = tid + G::THREADS * i;
            const int row = v / G::SLOTS, lslot = v % G::SLOTS;
            const int k = k0 + lslot * G::EPV, bn = n0 + row;
            const uint4 rb = (k < kend && bn < N) ? *reinterpret_cast<const uint4*>(B + (size_t)bn * ldb + k) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(base + G::A_BYTES + row * G::KB + ((lslot ^ G::swz(row)) * 16)) = rb;
        }
    };
    auto stage = [&](int slot, int t) {
        if (t > last_dma) stage_tail(slot, kbeg + t * G::KE);
        else dma(slot, kbeg + t * G::KE);
    };

#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < nk) stage(i, i);
    const int frow = lane & 15;
    int slot = 0, fill = NST - 1;
    for (int t = 0; t < nk; ++t) {
        // stage t has landed when at most the DMAs of the NST-2 younger stages are outstanding; near the end (fewer younger stages, or the
        // register-staged tail among them) everything is waited for -- those loads were issued at least one full stage earlier
        if (t + NST - 2 <= last_dma) gemm_vm_wait<(NST - 2) * DPS>();
        else gemm_vm_wait<0>();
        __syncthreads();
        if (t + NST - 1 < nk) stage(fill, t + NST - 1);
        const char* as = smem + slot * G::STAGE_BYTES + (wm * G::MI * 16 + frow) * G::KB;
        const char* bs = smem + slot * G::STAGE_BYTES + G::A_BYTES + (wn * G::NI * 16 + frow) * G::KB;
#pragma unroll
        for (int ks = 0; ks < G::KSUB; ++ks) {
            const int phys = ((ks * 4 + (lane >> 4)) ^ G::swz(frow)) * 16;
            uint4 fa[G::MI], fb[G::NI];
#pragma unroll
            for (int i = 0; i < G::MI; ++i) fa[i] = *reinterpret_cast<const uint4*>(as + i * 16 * G::KB + phys);
#pragma unroll
            for (int i = 0; i < G::NI; ++i) fb[i] = *reinterpret_cast<const uint4*>(bs + i * 16 * G::KB + phys);
#pragma unroll
            for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni) mfma_step<T>(acc[mi][ni], fb[ni], fa[mi]);
        }
        slot = slot + 1 == NST ? 0 : slot + 1;
        fill = fill + 1 == NST ? 0 : fill + 1;
    }
    __syncthreads();      // the epilogue reuses the ring
}

template <typename T, int KSUB>
__device__ __forceinline__ void gemm_mainloop(const T* __restrict__ A, const T* __restrict__ B, int M, int N, int lda,
                                              int ldb, int m0, int n0, int kbeg, int kend, char* smem,
                                              f32x4_t (&acc)[4][4]) {
    gemm_mainloop_cfg<GemmTile<T, KSUB>, T>(A, B, M, N, lda, ldb, m0, n0, kbeg, kend, smem, acc);
}

// coordinates of accumulator element (mi, ni) of this lane: m (one row), n (first of 4 consecutive columns)
template <typename G>
__device__ __forceinline__ int acc_row_cfg(int m0, int mi) {
    return m0 + ((threadIdx.x >> 6) / G::WN) * (G::MI * 16) + mi * 16 + (threadIdx.x & 15);
}
template <typename G>
__device__ __forceinline__ int acc_col_cfg(int n0, int ni) {
    return n0 + ((threadIdx.x >> 6) % G::WN) * (G::NI * 16) + ni * 16 + ((threadIdx.x & 63) >> 4) * 4;
}
__device__ __forceinline__ int acc_row(int m0, int mi) { return acc_row_cfg<GemmTile<float, 2>>(m0, mi); }
__device__ __forceinline__ int acc_col(int n0, int ni) { return acc_col_cfg<GemmTile<float, 2>>(n0, ni); }
