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

template <typename T, int KSUB>
struct GemmTile {
    static constexpr int TM = 128, TN = 128, THREADS = 256;
    static constexpr int KB = 64 * KSUB;               // bytes of K per row per stage
    static constexpr int KE = KB / (int)sizeof(T);     // elements of K per stage
    static constexpr int EPV = 16 / (int)sizeof(T);    // elements per 16-byte vector
    static constexpr int SLOTS = KB / 16;              // 16-byte slots per row (8 at KSUB = 2)
    static constexpr int ROWS_PER_DMA = 64 / SLOTS;    // rows covered by one 1-KiB wave-level LDS-DMA
    static constexpr int DMA_PER_OP = 128 / ROWS_PER_DMA;        // wave-instructions per operand per stage
    static constexpr int DMA_PER_WAVE = DMA_PER_OP / 4;
    static constexpr int OP_BYTES = 128 * KB;          // un-padded: the LDS image of an LDS-DMA is lane-linear
    static constexpr int STAGE_BYTES = 2 * OP_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static_assert(KSUB == 2, "the XOR swizzle below assumes 8 slots (128-byte rows)");
};

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
__device__ __forceinline__ void mfma_step<float>(f32x4_t& acc, const uint4& fn, const uint4& fm) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fn.x), __uint_as_float(fm.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fn.y), __uint_as_float(fm.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fn.z), __uint_as_float(fm.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fn.w), __uint_as_float(fm.w), acc, 0, 0, 0);
}

// XCD-aware workgroup -> tile mapping: the dispatcher round-robins consecutive workgroup ids over
// the 8 XCDs (private L2 each); remap so that each XCD walks a CONTIGUOUS run of tiles, i.e. the
// tiles that share an A row-panel hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
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
template <typename T, int KSUB>
__device__ __forceinline__ void gemm_mainloop(const T* __restrict__ A, const T* __restrict__ B, int M, int N, int lda,
                                              int ldb, int m0, int n0, int kbeg, int kend, char* smem,
                                              f32x4_t (&acc)[4][4]) {
    using G = GemmTile<T, KSUB>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = (kend - kbeg + G::KE - 1) / G::KE;
    if (nk <= 0) return;
    const bool tail = ((kend - kbeg) % G::KE) != 0;

    // per-lane source rows of this wave's DMA instructions (clamped) and the swizzled source slot
    const int drow = lane / G::SLOTS;                       // row within the 1-KiB piece
    const int pslot = lane % G::SLOTS;                      // physical slot this lane's 16 B land in
    const T* asrc[G::DMA_PER_WAVE];
    const T* bsrc[G::DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G::DMA_PER_WAVE; ++i) {
        const int row = (wave * G::DMA_PER_WAVE + i) * G::ROWS_PER_DMA + drow;
        const int lslot = pslot ^ (row & 7);                // logical 16-byte column of the tile row
        asrc[i] = A + (size_t)min(m0 + row, M - 1) * lda + lslot * G::EPV;
        bsrc[i] = B + (size_t)min(n0 + row, N - 1) * ldb + lslot * G::EPV;
    }
    auto dma = [&](int stage, int k0) {
        char* base = smem + stage * G::STAGE_BYTES + (wave * G::DMA_PER_WAVE) * 1024;
#pragma unroll
        for (int i = 0; i < G::DMA_PER_WAVE; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + k0), (lptr_t)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i] + k0), (lptr_t)(base + G::OP_BYTES + i * 1024), 16, 0, 0);
        }
    };
    // K tail: plain loads with zero fill, written to the same swizzled image
    auto stage_tail = [&](int stage, int k0) {
        char* base = smem + stage * G::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < (128 * G::SLOTS) / G::THREADS; ++i) {
            const int v = tid + G::THREADS * i;
            const int row = v / G::SLOTS, lslot = v % G::SLOTS;
            const int k = k0 + lslot * G::EPV;
            const bool kin = k < kend;
            const int am = m0 + row, bn = n0 + row;
            const uint4 ra = (kin && am < M) ? *reinterpret_cast<const uint4*>(A + (size_t)am * lda + k) : make_uint4(0, 0, 0, 0);
            const uint4 rb = (kin && bn < N) ? *reinterpret_cast<const uint4*>(B + (size_t)bn * ldb + k) : make_uint4(0, 0, 0, 0);
            const int off = row * G::KB + ((lslot ^ (row & 7)) * 16);
            *reinterpret_cast<uint4*>(base + off) = ra;
            *reinterpret_cast<uint4*>(base + G::OP_BYTES + off) = rb;
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
        const char* as = smem + cur * G::STAGE_BYTES + (wm * 64 + frow) * G::KB;
        const char* bs = smem + cur * G::STAGE_BYTES + G::OP_BYTES + (wn * 64 + frow) * G::KB;
#pragma unroll
        for (int ks = 0; ks < KSUB; ++ks) {
            const int phys = ((ks * 4 + (lane >> 4)) ^ (frow & 7)) * 16;
            uint4 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const uint4*>(as + i * 16 * G::KB + phys);
                fb[i] = *reinterpret_cast<const uint4*>(bs + i * 16 * G::KB + phys);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) mfma_step<T>(acc[mi][ni], fb[ni], fa[mi]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// coordinates of accumulator element (mi, ni) of this lane: m (one row), n (first of 4 consecutive columns)
__device__ __forceinline__ int acc_row(int m0, int mi) {
    return m0 + ((threadIdx.x >> 6) >> 1) * 64 + mi * 16 + (threadIdx.x & 15);
}
__device__ __forceinline__ int acc_col(int n0, int ni) {
    return n0 + ((threadIdx.x >> 6) & 1) * 64 + ni * 16 + ((threadIdx.x & 63) >> 4) * 4;
}
