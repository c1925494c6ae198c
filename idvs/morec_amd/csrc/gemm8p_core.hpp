// gemm8p_core.hpp -- the eight-phase 256 x 256 main loop shared by the encoder GEMMs (gemm8p.hip) and the scoring kernels
// (inbatch_ce8p.hip): LDS geometry, LDS-DMA context of a tile, prologue, the four-phase K-tile and the loop around it.
// The schedule itself is described at the top of gemm8p.hip.
#pragma once
#include "gemm_core.hpp"

namespace g8 {
typedef f32x16c_t f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

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
// AUX: cache policy of the loads (0 = default; 2 = nt: a panel that is streamed through once should not push the re-used one out of L2)
template <int AUX = 0>
__device__ __forceinline__ void dma2(__amdgpu_buffer_rsrc_t rs, uint32_t o0, uint32_t o1, int kbyte, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, o0, kbyte, 0, AUX);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + 1024), 16, o1, kbyte, 0, AUX);
}
struct Ctx;
// The same for a K-tile that may be the LAST one of a K that is not a multiple of 64: `m` = all ones when it is (wave-uniform),
// and the lanes whose 16-byte slot lies past K carry 0x80000000 in c.pz[]: the offset leaves the descriptor's range and the DMA
// writes zeros.  One v_and_or_b32 per instruction; with K % 64 == 0 pz is 0 and nothing changes.
template <int AUX = 0>
__device__ __forceinline__ void dma2z(__amdgpu_buffer_rsrc_t rs, uint32_t o0, uint32_t o1, const uint32_t (&pz)[2], uint32_t m, int kbyte, char* dst) {
    dma2<AUX>(rs, o0 | (pz[0] & m), o1 | (pz[1] & m), kbyte, dst);
}

struct Ctx {
    __amdgpu_buffer_rsrc_t ra, rb;               // descriptors of A / B, based at the tile's first row
    uint32_t a1[2], a2[2], b1[2], b2[2];         // per-lane byte offsets of the wave's two pieces of each half-tile
    int dA1, dA2, dB1, dB2;                      // wave-uniform LDS offsets (within a buffer) of those pieces
    int aoff, boff;                              // LDS offsets (within a buffer) of the wave's first A row / first B row
    int loff[4];                                 // per-lane fragment offset of MFMA k-step ks: row (lane & 31), swizzled slot
    uint32_t pz[2];                              // 0x80000000 where this lane's slot of piece j lies past K in the LAST K-tile, else 0
};

__device__ __forceinline__ uint4 lds16(const char* p) { return *reinterpret_cast<const uint4*>(p); }

// ZERO: the first K-tile of an output tile starts its accumulators from the MFMA's inline-constant 0 operand instead of 128
// v_mov per lane ahead of the loop.
// NBW = 32-row blocks this WAVE owns (4: the full 128-row wave tile; 3 / 2: wave row 1 of a 224- / 192-row tile, gemm8p.hip "tile
// height").  Compile time: a run-time branch around MFMAs makes the accumulators loop-carried through phi nodes and hipcc spills ~150 of
// them (measured: 576-704 bytes of scratch per lane, launches 7x slower); the caller branches ONCE, outside the main loop, between two
// instantiations.  Barriers, DMA and the fragment reads are those of the full tile; the accumulators of absent blocks are never read.
template <typename T16, int M0, int NQ, bool ZERO, int NBW = 4>
__device__ __forceinline__ void mfma_quadrant(f32x16_t (&acc)[4][2], const uint4 (&fa)[2][4], const uint4 (&fb)[4]) {
    if constexpr (M0 >= NBW) return;
    constexpr int NMI = (M0 + 1 < NBW) ? 2 : 1;
#ifndef G8_NO_SETPRIO      // (experiment builds only: scripts/shadow_build.sh)
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {  // operand-swapped: the lane ends up owning ONE m and runs of 4 consecutive n
            f32x16_t cin = acc[M0 + mi][NQ];
            if constexpr (ZERO) {
                if (ks == 0) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) cin[v] = 0.f;
                }
            }
            acc[M0 + mi][NQ] = h16<T16>::mma32(__builtin_bit_cast(bf16x8_t, fb[ks]), __builtin_bit_cast(bf16x8_t, fa[mi][ks]), cin);
        }
#ifndef G8_NO_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// vmcnt left in flight after a phase's issue: 8 in the steady state (the refills of the last four phases); the last two
// K-tiles issue fewer, so fewer may be left.  REM = K-tiles after this one, capped at 2 (compile time: no branches in the loop).
// SLACK: VMEM operations YOUNGER than the tile's prologue and OLDER than its in-loop refills that may also stay in flight -- the
// global stores of the previous tile's epilogue.  Vector memory operations complete in execution order, so "at most 8 + SLACK
// outstanding" still means "everything up to the 8 newest refills has landed" as long as SLACK does not exceed their number.
template <int REM, int W1, int W0, int SLACK = 0>
__device__ __forceinline__ void vm_wait_tail() {
    vm_wait<(REM >= 2 ? 8 + SLACK : (REM == 1 ? W1 : W0))>();
}

#ifdef G8_EXP_SHADOW
// EXPERIMENT (not in the product build): G8_EXP_SHADOW units of 4 GELU + derivative evaluations on dummy registers in every read / DMA
// segment -- what VALU work in the partner wave's MFMA shadow costs the main loop.
#define G8_GS_PARAM , float (&gs)[4]
#define G8_GS_ARG , gs
#ifndef G8_EXP_SHADOW_MASK
#define G8_EXP_SHADOW_MASK 15      // bit p: the units run in phase p of every K-tile
#endif
#define G8_SHADOW() G8_SHADOW_P(0)
#define G8_SHADOW_P(ph_)                                              \
    do {                                                              \
        if (!((G8_EXP_SHADOW_MASK >> (ph_)) & 1)) break;              \
        pin();                                                        \
        _Pragma("unroll") for (int u_ = 0; u_ < G8_EXP_SHADOW; ++u_) { \
            float d_[4];                                              \
            gelu4_with_deriv(gs, d_);                                 \
            _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) gs[r_] = gs[r_] * 0.5f + d_[r_]; \
        }                                                             \
        pin();                                                        \
    } while (0)
#else
#define G8_GS_PARAM
#define G8_GS_ARG
#define G8_SHADOW() do {} while (0)
#define G8_SHADOW_P(ph_) do {} while (0)
#endif
// One K-tile out of the buffer at byte offset `cb` (0 or BUF_BYTES); kb = byte offset of this K-tile within a row.
// last2 (REM == 2 only): K-tile t + 2, refilled in phases 2 / 3, is the last one of K.
// BAUX: cache policy of the B-panel loads (see dma2)
template <typename T16, int REM, int SLACK = 0, bool ZERO = false, int BAUX = 0, int NBW = 4>
__device__ __forceinline__ void ktile(char* smem, const Ctx& c, int cb, int kb, f32x16_t (&acc)[4][2], bool last2 G8_GS_PARAM) {
    const uint32_t m1 = REM == 1 ? 0xffffffffu : 0u;           // K-tile t + 1 is the last one exactly when REM == 1
    const uint32_t m2 = last2 ? 0xffffffffu : 0u;
    static_assert(SLACK == 0 || REM == 2, "slack only on a steady K-tile");
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
    if constexpr (REM >= 1) dma2z<BAUX>(c.rb, c.b2[0], c.b2[1], c.pz, m1, kb + KB, oth + OP_BYTES + c.dB2);
    G8_SHADOW_P(0);
    pin();
    vm_wait_tail<REM, 8, 2, SLACK>();
    bar();
    mfma_quadrant<T16, 0, 0, ZERO, NBW>(acc, fa, fb0);
    bar();
    // ---- phase 1: B-second fragments; refill A-second of t+1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb1[ks] = lds16(smem + adb[ks] + 32 * KB);
    if constexpr (REM >= 1) dma2z(c.ra, c.a2[0], c.a2[1], c.pz, m1, kb + KB, oth + c.dA2);
    G8_SHADOW_P(1);
    pin();
    vm_wait_tail<REM, 8, 0, SLACK>();
    bar();
    mfma_quadrant<T16, 0, 1, ZERO, NBW>(acc, fa, fb1);
    bar();
    // ---- phase 2: A-second fragments; refill A-first of t+2 (this buffer)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[mi][ks] = lds16(smem + ada[ks] + (64 + mi * 32) * KB);
    if constexpr (REM >= 2) dma2z(c.ra, c.a1[0], c.a1[1], c.pz, m2, kb + 2 * KB, cur + c.dA1);
    G8_SHADOW_P(2);
    pin();
    vm_wait_tail<REM, 6, 0, SLACK>();
    bar();
    mfma_quadrant<T16, 2, 1, ZERO, NBW>(acc, fa, fb1);
    bar();
    // ---- phase 3: nothing to read (B-first is still in registers); refill B-first of t+2
    if constexpr (REM >= 2) dma2z<BAUX>(c.rb, c.b1[0], c.b1[1], c.pz, m2, kb + 2 * KB, cur + OP_BYTES + c.dB1);
    G8_SHADOW_P(3);
    pin();
    vm_wait_tail<REM, 4, 0, SLACK>();
    bar();
    mfma_quadrant<T16, 2, 0, ZERO, NBW>(acc, fa, fb0);
    bar();
}

// ---- DMA / fragment context of one tile.  At, Bt = first row of the tile's A / B panel; rows_a, rows_b = rows that exist from
// there on (M - m0, N - n0; rows past them are clamped to the last valid one).  `tid` is an opaque copy of threadIdx.x: the
// context is recomputed per tile (a few dozen integer ops) instead of being kept alive across the epilogue.
__device__ __forceinline__ void make_ctx(Ctx& c, int tid, const bf16* At, const bf16* Bt, int rows_a, int rows_b, int lda, int ldb, int krem) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // DMA geometry: piece j of a half-tile = 8 rows; lane -> row (lane >> 3) of the piece, physical slot lane & 7
    const int ra = wr * 128 + wc * 16;                       // this wave's 16 rows of A-first (A-second: + 64)
    const int rb = (wave >> 1) * 64 + (wave & 1) * 16;       // this wave's 16 rows of B-first (B-second: + 32)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rl = j * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((rl >> 1) & 7);       // (row >> 1) & 7 with row = 16-aligned base + rl
        c.a1[j] = (uint32_t)min(ra + rl, rows_a - 1) * (uint32_t)(lda * 2) + slot * 16;
        c.a2[j] = (uint32_t)min(ra + 64 + rl, rows_a - 1) * (uint32_t)(lda * 2) + slot * 16;
        c.b1[j] = (uint32_t)min(rb + rl, rows_b - 1) * (uint32_t)(ldb * 2) + slot * 16;
        c.b2[j] = (uint32_t)min(rb + 32 + rl, rows_b - 1) * (uint32_t)(ldb * 2) + slot * 16;
        c.pz[j] = slot * 8 >= krem ? 0x80000000u : 0u;          // krem = elements of K in the unit's last K-tile (64: all of it)
    }
    // descriptors: raw (stride 0), extent = the rows of this tile that exist (every offset above stays inside it)
    const long abytes = (long)min(256, rows_a) * lda * 2, bbytes = (long)min(256, rows_b) * ldb * 2;
    c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)At, 0, (int)min(abytes, 0x7fffffffL), 0x00020000);
    c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bt, 0, (int)min(bbytes, 0x7fffffffL), 0x00020000);
    c.dA1 = ra * KB; c.dA2 = (ra + 64) * KB; c.dB1 = rb * KB; c.dB2 = (rb + 32) * KB;
    c.aoff = wr * 128 * KB;
    c.boff = wc * 64 * KB;
    const int r5 = lane & 31, fr = (r5 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) c.loff[ks] = r5 * KB + (((2 * ks + (lane >> 5)) ^ fr) << 4);
}

// Prologue of a tile: K-tile 0 entirely, the first halves of K-tile 1 (12 LDS-DMA instructions per lane).  LDS must be free
// of readers: called before the first tile and, for the NEXT tile, right after a main loop (every wave is past its last barrier)
// -- i.e. ahead of the finished tile's epilogue, whose slices live outside the two K-tile buffers.
template <int BAUX = 0>
__device__ __forceinline__ void issue_prologue(const Ctx& c, char* smem, int nk) {
    const uint32_t m = nk == 2 ? 0xffffffffu : 0u;
    dma2(c.ra, c.a1[0], c.a1[1], 0, smem + c.dA1);
    dma2<BAUX>(c.rb, c.b1[0], c.b1[1], 0, smem + OP_BYTES + c.dB1);
    dma2<BAUX>(c.rb, c.b2[0], c.b2[1], 0, smem + OP_BYTES + c.dB2);
    dma2(c.ra, c.a2[0], c.a2[1], 0, smem + c.dA2);
    dma2z(c.ra, c.a1[0], c.a1[1], c.pz, m, KB, smem + BUF_BYTES + c.dA1);
    dma2z<BAUX>(c.rb, c.b1[0], c.b1[1], c.pz, m, KB, smem + BUF_BYTES + OP_BYTES + c.dB1);
    pin();
}

// acc += A-panel . B-panel^T over K (K % 64 == 0, K >= 128), prologue already issued.  On return every DMA of this tile has
// landed and every wave has passed the last barrier: the K-tile buffers are free.
// The first wait is `vmcnt(8)` whatever else the wave has in flight: it bounds the number of PENDING loads by 8, and loads
// retire in order among themselves, so the four oldest prologue pieces have landed even with younger stores outstanding.
#define G8_MSTAMP(i)                                                                    \
    do {                                                                                \
        if (st && threadIdx.x == 0) st[(i)] = __builtin_readcyclecounter();              \
    } while (0)
template <typename T16, int SLACK, int BAUX = 0, int NBW = 4>
__device__ __forceinline__ void mainloop8p_s(const Ctx& c, int wr, int nk, char* smem, f32x16_t (&acc)[4][2], unsigned long long* st) {
    G8_MSTAMP(8);
    vm_wait<8 + SLACK>();    // A-first, B-first of K-tile 0 (this wave's pieces)
    G8_MSTAMP(9);
    bar();                   // ... everybody's
    if (wr == 1) bar();      // waves 4-7 run one barrier behind waves 0-3 from here on
    G8_MSTAMP(10);
#ifdef G8_EXP_SHADOW
    float gs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) gs[r] = (float)(c.loff[r] & 255) * 0.01f;
#endif
    int cb = 0;
    int t = 0;
    if (nk >= 3) {     // first K-tile: accumulators start from zero; the previous epilogue's stores drain under it
        ktile<T16, 2, SLACK, true, BAUX, NBW>(smem, c, cb, 0, acc, nk == 3 G8_GS_ARG);
        cb ^= BUF_BYTES;
        t = 1;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    }
    G8_MSTAMP(11);
    for (; t < nk - 2; ++t) {
        ktile<T16, 2, 0, false, BAUX, NBW>(smem, c, cb, t * KB, acc, t + 3 == nk G8_GS_ARG);
        cb ^= BUF_BYTES;
        if (t == 1) G8_MSTAMP(12);
    }
    G8_MSTAMP(13);
    ktile<T16, 1, 0, false, BAUX, NBW>(smem, c, cb, t * KB, acc, false G8_GS_ARG);
    ktile<T16, 0, 0, false, BAUX, NBW>(smem, c, cb ^ BUF_BYTES, (t + 1) * KB, acc, false G8_GS_ARG);
#ifdef G8_EXP_SHADOW
    asm volatile("" ::"v"(gs[0]), "v"(gs[1]), "v"(gs[2]), "v"(gs[3]));
#endif
    G8_MSTAMP(14);
    if (wr == 0) bar();      // waves 0-3 catch the trailing barrier of waves 4-7
    G8_MSTAMP(15);
}
// `younger`: VMEM operations this wave has issued since the tile's prologue (only a LOWER bound matters: see vm_wait_tail)
template <typename T16, int BAUX = 0>
__device__ __forceinline__ void mainloop8p(const Ctx& c, int wr, int nk, int younger, char* smem, f32x16_t (&acc)[4][2],
                                           unsigned long long* st) {
    if (younger >= 32) mainloop8p_s<T16, 32, BAUX>(c, wr, nk, smem, acc, st);
    else if (younger >= 16) mainloop8p_s<T16, 16, BAUX>(c, wr, nk, smem, acc, st);
    else mainloop8p_s<T16, 0, BAUX>(c, wr, nk, smem, acc, st);
}
}  // namespace g8
