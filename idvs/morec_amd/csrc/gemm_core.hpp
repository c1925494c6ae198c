// gemm_core.hpp -- the one MFMA main loop every matmul-shaped kernel of this library shares.
//
// Tile: 128 x 128 output per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave = 4 x 4 MFMA
// 16x16 tiles), K advanced 64*KSUB bytes per stage through a double-buffered LDS ring, global ->
// register -> LDS staging with 16-byte vectors, ds_read_b128 fragment reads.
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
    static constexpr int PITCH = KB + 16;              // LDS row pitch (bytes): odd multiple of 16 -> conflict-light b128 reads
    static constexpr int VEC_PER_ROW = KB / 16;
    static constexpr int LOADS = 128 * VEC_PER_ROW / THREADS;  // 16-byte vectors per thread per operand per stage
    static constexpr int OP_BYTES = 128 * PITCH;
    static constexpr int STAGE_BYTES = 2 * OP_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
};

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
template <typename T, int KSUB>
__device__ __forceinline__ void gemm_mainloop(const T* __restrict__ A, const T* __restrict__ B, int M, int N, int lda,
                                              int ldb, int m0, int n0, int kbeg, int kend, char* smem,
                                              f32x4_t (&acc)[4][4]) {
    using G = GemmTile<T, KSUB>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    uint4 ra[G::LOADS], rb[G::LOADS];
    int lrow[G::LOADS], lkv[G::LOADS];
#pragma unroll
    for (int i = 0; i < G::LOADS; ++i) {
        const int v = tid + G::THREADS * i;
        lrow[i] = v / G::VEC_PER_ROW;
        lkv[i] = v % G::VEC_PER_ROW;
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < G::LOADS; ++i) {
            const int k = k0 + lkv[i] * G::EPV;
            const bool kin = k < kend;
            const int am = m0 + lrow[i], bn = n0 + lrow[i];
            ra[i] = (kin && am < M) ? *reinterpret_cast<const uint4*>(A + (size_t)am * lda + k) : make_uint4(0, 0, 0, 0);
            rb[i] = (kin && bn < N) ? *reinterpret_cast<const uint4*>(B + (size_t)bn * ldb + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int stage) {
        char* base = smem + stage * G::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < G::LOADS; ++i) {
            const int off = lrow[i] * G::PITCH + lkv[i] * 16;
            *reinterpret_cast<uint4*>(base + off) = ra[i];
            *reinterpret_cast<uint4*>(base + G::OP_BYTES + off) = rb[i];
        }
    };

    const int nk = (kend - kbeg + G::KE - 1) / G::KE;
    if (nk <= 0) return;
    gload(kbeg);
    lstore(0);
    __syncthreads();
    const int frag_off = (lane & 15) * G::PITCH + (lane >> 4) * 16;
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) gload(kbeg + (t + 1) * G::KE);
        const char* as = smem + cur * G::STAGE_BYTES + (wm * 64) * G::PITCH + frag_off;
        const char* bs = smem + cur * G::STAGE_BYTES + G::OP_BYTES + (wn * 64) * G::PITCH + frag_off;
#pragma unroll
        for (int ks = 0; ks < KSUB; ++ks) {
            uint4 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const uint4*>(as + i * 16 * G::PITCH + ks * 64);
                fb[i] = *reinterpret_cast<const uint4*>(bs + i * 16 * G::PITCH + ks * 64);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) mfma_step<T>(acc[mi][ni], fb[ni], fa[mi]);
        }
        if (t + 1 < nk) lstore(cur ^ 1);
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
