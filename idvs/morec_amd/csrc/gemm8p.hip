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
#include <mutex>
#include "gemm8p_core.hpp"
#include "gemm_args.hpp"
#include "ce_args.hpp"

namespace {
using namespace g8;
// -----------------------------------------------------------------------------------------------------------------------
// Persistent kernel: workgroup b walks the tiles b, b + gridDim.x, ... (gridDim.x = one workgroup per CU); the prologue DMA of
// the next tile is issued BEFORE the epilogue of the finished one, so the first fill of the pipeline (7.4 k cycles when it
// was exposed: profiles/r02_gemm8p_stamps.txt), the workgroup turnover and most of the store drain overlap the epilogue.
//
// Epilogue: wave-private.  acc[Mi][Ni][4 g + r] = C[m0 + wr*128 + Mi*32 + (lane & 31)][n0 + wc*64 + Ni*32 + 8 g + 4 (lane >> 5) + r].
// Every 32-row block goes through the wave's own 4-KiB LDS slice ([32 rows][128 B], 16-byte slots XOR-swizzled with row & 7;
// above the K-tile buffers) so that all global traffic (the stores, and the loads of the activation-derivative operand) is
// 16-byte lanes along rows: full 128-byte lines per row per instruction.  128 B = 64 bf16 columns (one pass per block) or
// 32 fp32 columns (two passes).
//
// tile_body's four panel pointers are `__restrict__` for the sake of hipcc's s_waitcnt insertion, not of the optimiser: inlining
// a function with noalias arguments tags the LDS-DMA instructions (based on the panels) with alias scopes and every other LDS
// access with "does not alias them".  Without the tags the compiler puts `s_waitcnt vmcnt(0)` in front of every ds_read that
// follows an LDS-DMA -- a full drain of the DMA queue three times per K-tile.  (Nothing is written through the panels, so
// overlapping A / B tiles behind different restrict pointers are fine.)  The hand-counted vmcnt waits + barriers order the
// DMA data for the reads.
// -----------------------------------------------------------------------------------------------------------------------
constexpr int SLICE = 4096;
constexpr int LDS_TOTAL = LDS_BYTES + 8 * SLICE;       // 160 KiB: the whole LDS of a CU

#define G8_STAMPW(i, w)                                                                                  \
    do {                                                                                                 \
        if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)(w) * 16 + (i)] = __builtin_readcyclecounter(); \
    } while (0)

// One unit of a workgroup's work list: an output tile over K-tiles [k0, k0 + nkt).  tail >= 0: one HALF (chunk 0 / 1) of the
// K range of tail tile number `tail` (see gemm8p_kernel); the two halves meet through p.tail_ws / p.tail_cnt.
struct TileXY {
    int wg, m0, n0;
    int k0, nkt, tail, chunk;
    int krem;      // elements of K inside the unit's last K-tile (64 unless the unit ends at a K that is not a multiple of 64)
};

// TMR: rows of a tile -- 256, or 224 / 192 (wave row 1 owns three / two 32-row blocks instead of four): "tile height", below, at launch8p.
template <typename TI, typename TO, int ACT, bool CS, int TMR = 256>
__device__ __forceinline__ void tile_body(const GemmArgs& p, char* smem, const bf16* __restrict__ Acur, const bf16* __restrict__ Bcur,
                                          const bf16* __restrict__ Anext, const bf16* __restrict__ Bnext, const TileXY cur, const TileXY nxt,
                                          const bool first) {
    constexpr bool PART = TMR != TM;
    constexpr int tmr = TMR;                                  // rows of a tile
    constexpr int nb1 = (TMR - 128) / 32;                     // 32-row blocks of wave row 1 (wave row 0 always owns four)
    constexpr int ES = (int)sizeof(TO);
    constexpr int EPV = 16 / ES;                 // elements per 16-byte vector
    constexpr int PC = 128 / ES;                 // columns per pass: 64 (bf16) / 32 (f32)
    constexpr int NPASS = 64 / PC;               // passes per 32-row block: 1 / 2
    constexpr int NG = PC / 8;                   // 4-element accumulator groups of this lane per pass: 8 / 4
    struct Vecs { u32x4_t q[4]; };
    const int m0 = cur.m0, n0 = cur.n0;
    G8_STAMPW(0, cur.wg);
    f32x16_t acc[4][2];
    float bias_l;
    {
        int tid_m = threadIdx.x;
        asm volatile("" : "+v"(tid_m));
        {   // any valid address when there is no bias (discarded in the epilogue): no branch around a load
            const float* bp = p.bias ? p.bias : reinterpret_cast<const float*>(p.B);
            bias_l = bp[min(n0 + ((tid_m >> 6) & 3) * 64 + (tid_m & 63), p.N - 1)];
        }
        Ctx c;
        make_ctx(c, tid_m, Acur, Bcur, min(p.M - m0, tmr), p.N - n0, p.lda, p.ldb, cur.krem);
        if (first) issue_prologue(c, smem, cur.nkt);      // later tiles: issued by the previous tile's body, ahead of its epilogue
        // global stores of the previous tile's epilogue (issued behind this tile's prologue): 4 per pass and output
        constexpr int NST = 16 * (int)sizeof(TO) / 2;
        const int younger = (first || (p.debug & 3) || (p.debug & 64)) ? 0 : (p.aux_out ? 2 * NST : NST);
        const int wr_m = __builtin_amdgcn_readfirstlane(tid_m >> 8);
        unsigned long long* st_m = p.stamps ? p.stamps + (size_t)cur.wg * 16 : nullptr;
        if constexpr (!PART) {
            mainloop8p<TI>(c, wr_m, cur.nkt, younger, smem, acc, st_m);
        } else {      // ONE branch, outside the loop: wave row 0 runs the full wave tile, wave row 1 the instantiation without its absent blocks
            (void)younger;
            if (wr_m == 0) mainloop8p_s<TI, 0, 0, 4>(c, 0, cur.nkt, smem, acc, st_m);
            else mainloop8p_s<TI, 0, 0, nb1>(c, 1, cur.nkt, smem, acc, st_m);
        }
    }
    G8_STAMPW(1, cur.wg);
    if (cur.tail >= 0) {
        // ---- tail split: this workgroup holds the sums over PART of K.  Both parts park their accumulators in p.tail_ws -- lane-linear,
        // 32 x [512 lanes x 16 B]: every store / load instruction moves one contiguous 8 KiB -- and take a ticket; the SECOND arrival
        // adds the other part to its registers and runs the epilogue, the first one is done.  Nobody ever waits for another
        // workgroup (no co-residency assumption: two such grids sharing a GPU cannot deadlock each other), and a + b does not depend
        // on who arrives last: results are bit-identical from run to run.  The exchange uses agent-scope (sc1) stores and loads,
        // which go through to the memory side of the per-XCD L2s: "store complete" (vmcnt 0) is "visible to the other XCDs".  (A
        // release fence instead -- __threadfence: buffer_wbl2 -- writes back EVERY dirty line of the XCD's L2, i.e. the output
        // tiles of the rounds before: 60 us per launch.)  The part with the shorter K range normally arrives first, so the
        // longer one finds the data waiting.
        int tid_t = threadIdx.x;
        asm volatile("" : "+v"(tid_t));
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.tail_ws + (size_t)(cur.tail * 2 + cur.chunk) * (TM * TN)), 0,
                                                                            TM * TN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[i][j][4 * g + e]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rw, tid_t * 16, ((i * 2 + j) * 4 + g) * (THREADS * 16), 16);
                }
        vm_wait<0>();
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem + LDS_BYTES);      // first word of wave 0's (idle) epilogue slice
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(p.tail_cnt + cur.tail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(*flag);
        __syncthreads();
        if (ticket == 0) return;
        if (threadIdx.x == 0) __hip_atomic_store(p.tail_cnt + cur.tail, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.tail_ws + (size_t)(cur.tail * 2 + (cur.chunk ^ 1)) * (TM * TN)), 0,
                                                                            TM * TN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // 8 x 16 B per lane in flight per round trip (16 would spill the accumulators)
            u32x4_t v[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) v[j][g] = __builtin_amdgcn_raw_buffer_load_b128(rr, tid_t * 16, ((i * 2 + j) * 4 + g) * (THREADS * 16), 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += __uint_as_float(v[j][g][e]);
            pin();
        }
    }
    // ---- this tile's epilogue state.  Lane constants come from an OPAQUE copy of the thread id so that they are recomputed
    // here (a dozen integer ops) instead of being kept alive -- i.e. spilled -- across the main loop.
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane = tid_e & 63, r5 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid_e >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    char* ws = smem + LDS_BYTES + wave * SLICE;
    TO* C = reinterpret_cast<TO*>(p.C);
    TO* aux = reinterpret_cast<TO*>(p.aux_out);
    const TO* din = reinterpret_cast<const TO*>(p.dact_in);
    auto wfence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // row side of the slice: vector i of this lane = row (lane >> 3) + 8 i, 16-byte slot lane & 7
    const int rs_row = lane >> 3, rs_slot = lane & 7;
    const int rs_off = rs_row * 128 + ((rs_slot ^ (rs_row & 7)) << 4);    // + i * 1024 (row + 8 keeps row & 7)
    // accumulator side: group q of a pass = columns 8 q + 4 h .. + 3 of the pass (bf16: 8 bytes, slot q; f32: 16 bytes, slot 2 q + h)
    auto cell = [&](int q) {
        const int slot = ES == 2 ? q : 2 * q + h;
        return reinterpret_cast<TO*>(ws + r5 * 128 + ((slot ^ (r5 & 7)) << 4) + (ES == 2 ? 8 * h : 0));
    };
    const int cur_tm = m0 / tmr;
    const int mw = m0 + wr * 128, nw = n0 + wc * 64;
    const int m_end = min(p.M, m0 + tmr);                     // first row past this tile
    const int nblk = (PART && wr) ? nb1 : 4;                  // this wave's 32-row blocks
    // Row side of the global traffic through buffer descriptors based at the tile's first row: ONE per-lane offset register
    // for every access of the tile (+ a wave-uniform term), and the hardware range check drops rows >= M (the extent is
    // the tile's valid rows); lanes whose 16-byte column lies past N carry an out-of-range offset instead of a branch.
    const long tile_bytes = (long)min(tmr, p.M - m0) * p.ldc * ES;
    const int ext = (int)min(tile_bytes, 0x7fffffffL);
    const size_t tile_off = (size_t)m0 * p.ldc;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(C + tile_off), 0, ext, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)(((ACT >= 3) ? din : C) + tile_off), 0, ext, 0x00020000);
    auto lane_off = [&](int ps) {    // byte offset of (row rs_row, this lane's column of pass ps) from (tile row 0, column 0)
        const int n = nw + ps * PC + rs_slot * EPV;
        return n < p.N ? (uint32_t)((rs_row * p.ldc + n) * ES) : 0x80000000u;
    };
    auto row_term = [&](int Mi, int i) { return (uint32_t)((wr * 128 + Mi * 32 + 8 * i) * p.ldc * ES); };   // wave-uniform
    auto rows_store = [&](const __amdgpu_buffer_rsrc_t& rs, int Mi, uint32_t lo) {
        u32x4_t q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const u32x4_t*>(ws + rs_off + i * 1024);
        if (!(p.debug & 1)) {
            const int cp = (p.debug >> 4) & 3;      // experiment: cache policy of the output stores
            if (cp == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q[i], rs, lo + row_term(Mi, i), 0, 0);
            } else if (cp == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q[i], rs, lo + row_term(Mi, i), 0, 2);
            } else if (cp == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q[i], rs, lo + row_term(Mi, i), 0, 16);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q[i], rs, lo + row_term(Mi, i), 0, 18);
            }
        }
    };
    auto rows_fetch = [&](const __amdgpu_buffer_rsrc_t& rs, int Mi, uint32_t lo) {     // rows >= M / columns >= N read as zero
        Vecs r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r.q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lo + row_term(Mi, i), 0, 0);
        return r;
    };
    auto rows_put = [&](const Vecs r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(ws + rs_off + i * 1024) = r.q[i];
    };
    uint32_t lo[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) lo[ps] = lane_off(ps);

    // ---- bias: one float per lane (column nw + lane), loaded AHEAD of the main loop (bias_l, above: its round trip hides under
    // the K-tiles and costs one register there), spread to the lanes' 8 column groups through the wave's slice.  (A load issued
    // here instead would be waited for in front of the first pass: 1.4 k cycles per tile, profiles/r02_gemm8p_stamps.txt.)
    // (the activation-derivative epilogues never carry a bias -- dX = (dY W) * act' -- and need the 32 registers for the second
    // prefetch set of their operand: the launcher refuses bias + dact for this kernel)
    float4 bv[2][4];
    if constexpr (ACT < 3) {
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
    // the activation-derivative operand, fetched TWO passes ahead (two register sets): a pass is ~1.2 k cycles of work, a global
    // load queued behind the previous pass's output stores returns after 2-4 k (profiles/r02_gemm8p_stamps.txt: 5 k cycles per
    // block with one pass of lead)
    Vecs dq[2] = {Vecs(), Vecs()};
    if constexpr (ACT >= 3) {
        dq[0] = rows_fetch(rD, 0, lo[0]);
        dq[1] = rows_fetch(rD, 1 / NPASS, lo[1 % NPASS]);
    }
    pin();

    // ---- next tile: its prologue flies under this tile's epilogue.  Issued UNCONDITIONALLY (after the last tile it re-reads that
    // tile's panels into buffers nobody reads; the kernel drains it before exiting): behind a branch, hipcc's s_waitcnt for the
    // bias loads above has to assume the path without the twelve younger DMA pieces and drains them too.
    {
        int tid_n = threadIdx.x;
        asm volatile("" : "+v"(tid_n));
        Ctx cn;
        make_ctx(cn, tid_n, Anext, Bnext, min(p.M - nxt.m0, tmr), p.N - nxt.n0, p.lda, p.ldb, nxt.krem);
        issue_prologue(cn, smem, nxt.nkt);
    }
    if (p.debug & 2) {      // ablation: no epilogue (the store keeps the accumulators alive)
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.C)[0] = acc[1][1][3] + acc[2][0][7] + acc[3][1][15] + bv[0][0].x;
        return;
    }
    if (p.stamps) { asm volatile("" :: "v"(bv[0][0].x), "v"(bv[1][3].w)); G8_STAMPW(2, cur.wg); }
    [[maybe_unused]] float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int Mi = 0; Mi < 4; ++Mi) {
        if constexpr (PART) {
            if (Mi >= nblk) break;      // (wave-uniform) blocks past the tile's rows: never computed
        }
        [[maybe_unused]] const bool row_ok = (mw + Mi * 32 + r5) < m_end;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            pin();      // nothing of this pass may be computed ahead of the previous one (128 fresh values on top of the accumulators)
            if constexpr (ACT >= 3) {
                constexpr int LAST = 4 * NPASS - 1;
                const int it = Mi * NPASS + ps, nx = it + 2;
                rows_put(dq[it & 1]);
                if (nx <= LAST) dq[it & 1] = rows_fetch(rD, nx / NPASS, lo[nx % NPASS]);
                wfence();
            }
            if (aux) {      // second output first (wave-uniform branch): the pre-activation, or act'(pre) with aux_deriv
                if constexpr (ACT == 1 || ACT == 2) {
                    // one evaluation of the activation serves both outputs: the results are parked as packed bf16 (16 registers)
                    // while the slice carries the second output to its rows
                    static_assert(ES == 2, "activation epilogues are instantiated for bf16 outputs");
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
                        if (p.aux_deriv) {
                            if constexpr (ACT == 1) {
                                gelu4_with_deriv(v, d);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) { d[r] = v[r] > 0.f ? 1.f : 0.f; v[r] = fmaxf(v[r], 0.f); }
                            }
                        } else {
                            if constexpr (ACT == 1) {
                                gelu4(v);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                            }
                        }
                        io<TO>::store4(cell(q), d);
                        park[q] = make_uint2(h16<TI>::pack2(v[0], v[1]), h16<TI>::pack2(v[2], v[3]));
                        if (q & 1) pin();      // keeps the scheduler from computing every group ahead of the first store (VGPR pressure)
                    }
                    wfence();
                    rows_store(__builtin_amdgcn_make_buffer_rsrc((void*)(aux + tile_off), 0, ext, 0x00020000), Mi, lo[ps]);
                    wfence();
#pragma unroll
                    for (int q = 0; q < NG; ++q) *reinterpret_cast<uint2*>(cell(q)) = park[q];
                    wfence();
                    rows_store(rC, Mi, lo[ps]);
                    wfence();
                    continue;
                } else {
#pragma unroll
                    for (int q = 0; q < NG; ++q) {
                        const int Ni = ES == 2 ? q / 4 : ps, g = ES == 2 ? q % 4 : q;
                        const float4 b = bv[Ni][g];
                        float pre[4];
                        pre[0] = fmaf(acc[Mi][Ni][4 * g + 0], p.alpha, b.x);
                        pre[1] = fmaf(acc[Mi][Ni][4 * g + 1], p.alpha, b.y);
                        pre[2] = fmaf(acc[Mi][Ni][4 * g + 2], p.alpha, b.z);
                        pre[3] = fmaf(acc[Mi][Ni][4 * g + 3], p.alpha, b.w);
                        io<TO>::store4(cell(q), pre);
                        if (q & 1) pin();
                    }
                    wfence();
                    rows_store(__builtin_amdgcn_make_buffer_rsrc((void*)(aux + tile_off), 0, ext, 0x00020000), Mi, lo[ps]);
                    wfence();
                }
            }
            // a lane's cells are its own: the derivative operand is read (all groups first: one LDS round trip, not eight dependent
            // ones) and replaced in place
            [[maybe_unused]] uint32_t uraw[NG][ES == 2 ? 2 : 4];
            if constexpr (ACT >= 3) {
#pragma unroll
                for (int q = 0; q < NG; ++q) {
                    if constexpr (ES == 2) {
                        const uint2 t = *reinterpret_cast<const uint2*>(cell(q));
                        uraw[q][0] = t.x; uraw[q][1] = t.y;
                    } else {
                        const uint4 t = *reinterpret_cast<const uint4*>(cell(q));
                        uraw[q][0] = t.x; uraw[q][1] = t.y; uraw[q][2] = t.z; uraw[q][3] = t.w;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const int Ni = ES == 2 ? q / 4 : ps, g = ES == 2 ? q % 4 : q;
                const float4 b = bv[Ni][g];
                float v[4];
                v[0] = fmaf(acc[Mi][Ni][4 * g + 0], p.alpha, b.x);
                v[1] = fmaf(acc[Mi][Ni][4 * g + 1], p.alpha, b.y);
                v[2] = fmaf(acc[Mi][Ni][4 * g + 2], p.alpha, b.z);
                v[3] = fmaf(acc[Mi][Ni][4 * g + 3], p.alpha, b.w);
                if constexpr (ACT == 1) {
                    gelu4(v);
                } else if constexpr (ACT == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if constexpr (ACT >= 3) {
                    float u[4];
                    if constexpr (ES == 2) {
                        u[0] = h16<TI>::bits2f(uraw[q][0] & 0xffffu); u[1] = h16<TI>::bits2f(uraw[q][0] >> 16);
                        u[2] = h16<TI>::bits2f(uraw[q][1] & 0xffffu); u[3] = h16<TI>::bits2f(uraw[q][1] >> 16);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) u[r] = __uint_as_float(uraw[q][r]);
                    }
                    if constexpr (ACT == 3) {
                        dgelu4_mul(v, u);
                    } else if constexpr (ACT == 4) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (u[r] > 0.f) ? v[r] : 0.f;
                    } else {        // MOREC_DACT_MUL: the operand already holds act'(pre)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] *= u[r];
                    }
                }
                if constexpr (CS) {     // rows past M hold copies of row M - 1 (clamped loads): keep them out of the column sums
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = row_ok ? v[r] : 0.f;
                }
                io<TO>::store4(cell(q), v);
                if (ACT != 5 && (q & 1)) pin();
            }
            wfence();
            if constexpr (CS) {
                // column sums of the block AS STORED (rounded to TO), taken from the very vectors the global stores carry: this lane's
                // four 16-byte pieces = rows (lane >> 3) + 8 i, columns 8 (lane & 7) .. + 7 of the block -- 8 partial column sums per
                // lane over 16 rows of the wave's 128, folded over the 8 lanes that share a column group once per tile.  (The first
                // form read the staged block back column by column: 32 two-byte LDS reads per pass and lane, a third of this
                // epilogue's time.)
                static_assert(!CS || ES == 2, "fused column sums: bf16 output only");
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
                if (!(p.debug & 1)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q[i], rC, lo[ps] + row_term(Mi, i), 0, 0);
                }
            } else {
                rows_store(rC, Mi, lo[ps]);
            }
            wfence();
        }
        G8_STAMPW(3 + Mi, cur.wg);
    }
    G8_STAMPW(7, cur.wg);
    if constexpr (CS) {         // one partial row per 128-row wave block; the launcher folds them
#pragma unroll
        for (int j = 0; j < 8; ++j) {      // fold over the 8 lanes (rows lane >> 3) that hold the same 8 columns
            cs8[j] += __shfl_xor(cs8[j], 8, 64);
            cs8[j] += __shfl_xor(cs8[j], 16, 64);
            cs8[j] += __shfl_xor(cs8[j], 32, 64);
        }
        if (rs_row == 0) {
            const int n = nw + rs_slot * 8;
            float* dst = p.colsum + (size_t)(cur_tm * 2 + wr) * p.N + n;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (n + j < p.N) dst[j] = cs8[j];
        }
    }
}

template <typename TI, typename TO, int ACT, bool CS, int TMR = 256>
__global__ __launch_bounds__(THREADS) void gemm8p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = p.tiles_m * p.tiles_n;
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    const int Keff = (p.debug & 4) ? 128 : p.K;
    const int nk = (Keff + KE - 1) / KE;
    const int krem_all = Keff - (nk - 1) * KE;
    // Work list of workgroup b (G = gridDim.x): F = nwg / G full rounds of tiles b, b + G, ...; then the R = nwg % G tail tiles.
    // When at most half of the workgroups would have a tail tile (2 R <= G) and K is long (>= 24 K-tiles: below that the exchange
    // costs what the split saves), every tail tile is cut in two K ranges taken by workgroups r and r + R: the last, partly
    // filled round costs ~0.54 of a tile time + the exchange instead of a whole one (600 tiles on 256 CUs: 2.6 instead of 3).
    const int G = gridDim.x, F = nwg / G, R = nwg - F * G;
    const bool split = p.tail_ws != nullptr && F >= 1 && R >= 1 && 2 * R <= G && nk >= 24;
    const int nk0 = (nk + p.tail_bias) / 2;   // part 0's share: a little more than half, so that part 1's data is normally waiting when part 0 arrives; flat optimum 2..6
    const int n_units = F + ((int)blockIdx.x < (split ? 2 * R : R) ? 1 : 0);
    auto unit_at = [&](int i) {
        TileXY t;
        int vb = (int)blockIdx.x + i * G;
        t.k0 = 0; t.nkt = nk; t.tail = -1; t.chunk = 0; t.krem = krem_all;
        if (i >= F && split) {
            t.chunk = (int)blockIdx.x >= R ? 1 : 0;
            t.tail = (int)blockIdx.x - t.chunk * R;
            vb = F * G + t.tail;
            t.k0 = t.chunk ? nk0 : 0;
            t.nkt = t.chunk ? nk - nk0 : nk0;
            t.krem = t.chunk ? krem_all : KE;
        }
        // Tile order: XCD x walks a CONTIGUOUS run of order indices (xcd_remap).  Plain order = row-major over (m, n): the
        // 32 tiles an XCD works on at a time are 32 / tiles_n M-panels x ALL N-panels, i.e. the whole of B streams through its
        // 4-MiB L2 once per round (N = 3072, K = 768: 4.7 MB -- nothing survives to the next round: profiles/r02e_gemm_pmc.json,
        // 573 MB fetched for 89 MB of operands).  With column groups of p.ngroup N-tiles (order: group, m, n within the group) the
        // XCD's block is (32 / ngroup) x ngroup panels -- fewest distinct panels per round when it is about square -- and an XCD
        // only ever touches the B panels of one or two groups.
        t.wg = xcd_remap(vb, nwg);
        int tm, tn;
        if (p.ngroup <= 0 || p.ngroup >= p.tiles_n) {
            tm = t.wg / p.tiles_n;
            tn = t.wg % p.tiles_n;
        } else {
            const int per = p.ngroup * p.tiles_m, ng = (p.tiles_n + p.ngroup - 1) / p.ngroup;
            const int g = min(t.wg / per, ng - 1), r = t.wg - g * per;
            const int w = g == ng - 1 ? p.tiles_n - g * p.ngroup : p.ngroup;
            tm = r / w;
            tn = g * p.ngroup + r % w;
        }
        t.m0 = tm * TMR;
        t.n0 = tn * TN;
        return t;
    };
    TileXY cur = unit_at(0);
    if (p.debug >> 8) {     // experiment: de-synchronise the workgroups' store bursts with a staggered start (G groups over one tile time)
        const int GS = (p.debug >> 8) & 0xff;
        const long long T = (long long)nk * 2300 + 9000;
        const long long until = (long long)__builtin_readcyclecounter() + T * ((blockIdx.x >> 3) % GS) / GS;
        while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(16);
    }
    for (int i = 0; i < n_units; ++i) {
        const bool more = i + 1 < n_units;
        const TileXY nxt = more ? unit_at(i + 1) : cur;
        tile_body<TI, TO, ACT, CS, TMR>(p, smem, A + (size_t)cur.m0 * p.lda + cur.k0 * KE, B + (size_t)cur.n0 * p.ldb + cur.k0 * KE,
                               A + (size_t)nxt.m0 * p.lda + nxt.k0 * KE, B + (size_t)nxt.n0 * p.ldb + nxt.k0 * KE, cur, nxt, i == 0);
        cur = nxt;
    }
    vm_wait<0>();      // the trailing prologue must not land in LDS that already belongs to another workgroup
}

// Tail-split scratch: per stream (launches on one stream are ordered, so they can share it; two streams must not), sized for the
// largest split (n_cu / 2 tail tiles x 2 parts x 256 KiB = 64 MiB at 256 CUs) + the arrival counters, allocated on first use
// and kept.  More than 8 streams: the ninth runs without the split.
static void tail_workspace(hipStream_t s, int n_cu, GemmArgs& a) {
    struct Slot { hipStream_t s; float* ws; int* cnt; int dev; };
    static Slot slots[8];
    static int n_slots = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < n_slots; ++i)
        if (slots[i].s == s && slots[i].dev == dev) { a.tail_ws = slots[i].ws; a.tail_cnt = slots[i].cnt; return; }
    if (n_slots == 8) return;
    const size_t ws_bytes = (size_t)n_cu * TM * TN * sizeof(float), cnt_bytes = (size_t)n_cu * sizeof(int);
    char* base = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&base), ws_bytes + cnt_bytes) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipMemset(base + ws_bytes, 0, cnt_bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(base); return; }
    slots[n_slots] = Slot{s, reinterpret_cast<float*>(base), reinterpret_cast<int*>(base + ws_bytes), dev};
    a.tail_ws = slots[n_slots].ws;
    a.tail_cnt = slots[n_slots].cnt;
    ++n_slots;
}

int g_reserve_cus = 0;      // tuning key "gemm8p_reserve_cus"
int gemm8p_cu_count() {     // CUs the persistent grid uses right now (device count rounded to the 8 XCDs, minus the reserved ones)
    static const int n_dev = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n &= ~7;
        return n < 8 ? 8 : n;
    }();
    const int n = n_dev - (g_reserve_cus & ~7);
    return n < 8 ? 8 : n;
}
// tuning key "gemm8p_ngroup": 0 = row-major tile order, -1 = automatic column groups (default: FETCH_SIZE per launch -20 % on the QKV
// projection, -24 % on FFN-up + GELU, -23 % on its backward at equal launch times; profiles/r03_gemm8p_tile_order_pmc.txt), n = groups of n N-tiles
int g_ngroup = -1;

// ---- tile height.  A launch is ceil(tiles / CUs) ROUNDS of tiles; with 256-row tiles the encoder's N = 768 products at ~55 k token rows
// are 645 tiles = 2.52 rounds, i.e. THREE rounds of tile time for 2.52 rounds of work (16 % of the launch is idle CUs), N = 3072 is 10.08
// rounds -> eleven.  The number of rounds is an integer either way, but the tile's height need not be 256: with 224-row tiles (wave row 1
// owns three 32-row blocks instead of four; its fourth block's MFMAs are branched around, everything else -- barriers, DMA, LDS image --
// is the 256-row schedule) the same N = 768 product is 738 tiles = 2.88 rounds of 7/8-size tiles: three rounds x 0.875.  pick_tmr takes
// the height in {256, 224, 192} with the smallest rounds x (K-tiles x height / 256 + fixed per-tile cost in K-tile units), 256 on ties.
// The result is the same numbers bit for bit (a tile boundary does not enter any sum).  MOREC_GEMM8P_TMR=0 / tuning key "gemm8p_tmr" 0: off.
int g_tmr_mode = -1;      // -1: read MOREC_GEMM8P_TMR on first use; 0: always 256; 1: automatic; 224 / 192: forced where an instantiation exists
static int pick_tmr(int M, int tiles_n, int nk, int n_cu) {
    if (g_tmr_mode < 0) {
        const char* e = getenv("MOREC_GEMM8P_TMR");
        g_tmr_mode = e ? atoi(e) : 1;
    }
    if (g_tmr_mode == 0) return TM;
    if (g_tmr_mode == 224 || g_tmr_mode == 192) return g_tmr_mode;
    const double fixed = 4.0;      // prologue + first barrier + epilogue of a tile, in K-tile times (profiles/r03_gemm8p_stamps.txt: ~8.4 k of 2.25 k cycles)
    int best = TM;
    double best_cost = 0.0;
    for (int t : {256, 224, 192}) {
        const long tiles = (long)((M + t - 1) / t) * tiles_n;
        if (tiles <= n_cu && t != TM) continue;                                        // a single round: nothing to balance
        const double cost = (double)((tiles + n_cu - 1) / n_cu) * ((double)nk * t / 256.0 + fixed);
        if (t == TM || cost < best_cost * 0.985) { if (t == TM || cost < best_cost) { best = t; best_cost = cost; } }
    }
    return best;
}

template <typename TI, typename TO, int ACT, bool CS, int TMR = 256>
int launch8p(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    constexpr bool PART = TMR != TM;
    const int tmr = TMR;
    a.tmr = TMR;
    a.tiles_m = (d->M + tmr - 1) / tmr;
    a.tiles_n = (d->N + TN - 1) / TN;
    static const int n_cu_dev = [] {      // thread-safe one-time set-up (function-local static)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<TI, TO, ACT, CS, TMR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_TOTAL);
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n &= ~7;      // a multiple of the 8 XCDs: workgroup b and its later tiles b + k * grid stay on XCD b % 8
        return n < 8 ? 8 : n;
    }();
    // morec_tuning_set("gemm8p_reserve_cus", r): leave r CUs (rounded to a multiple of 8) out of the persistent grid -- room for the
    // workgroups of a concurrent RCCL ring kernel when the step runs data-parallel (a persistent 160-KiB-LDS workgroup per CU
    // otherwise owns the whole chip until its launch ends)
    int n_cu = n_cu_dev - (g_reserve_cus & ~7);
    if (n_cu < 8) n_cu = 8;
    const int nwg = a.tiles_m * a.tiles_n;
    {   // column groups of the tile order: automatic = groups of about 6 N-tiles (a 32-CU XCD then works on a ~5 x 6 block)
        const int ng = (a.tiles_n + 5) / 6;
        a.ngroup = g_ngroup < 0 ? (a.tiles_n + ng - 1) / ng : g_ngroup;
    }
    a.tail_ws = nullptr;
    a.tail_cnt = nullptr;
    // Tail split: OPT-IN (tuning key "gemm8p_tail_split" / MOREC_GEMM8P_TAIL_SPLIT=1).  It is as accurate as the unsplit sum (same
    // error against an exact product) but rounds 0.02 % of the outputs the other way, and a 12-layer bf16 encoder at random init
    // turns that into +-1e-2 on the step-0 loss (tests/test_bench_mode_parity_gpu.py); the default keeps the summation order of
    // the two-buffer kernels (bit-identical outputs) for 0.35 % of the step time.
    if (!PART && a.tail_split && nwg > n_cu && nwg % n_cu && 2 * (nwg % n_cu) <= n_cu && d->K >= 24 * KE && !(a.debug & 128)) tail_workspace(s, n_cu, a);
    hipLaunchKernelGGL((gemm8p_kernel<TI, TO, ACT, CS, TMR>), dim3(nwg < n_cu ? nwg : n_cu), dim3(THREADS), LDS_TOTAL, s, a);
    MOREC_CHECK_LAUNCH();
    if constexpr (CS) return colsum_f32_launch(a.colsum, a.colsum_dst, a.tiles_m * 2, d->N, s);
    return MOREC_OK;
}
}  // namespace

// 0: automatic (eligible large problems), 1: never, 2: every eligible problem regardless of size
static int g_mode8p = -1, g_debug8p = 0, g_tail_bias = 2, g_tail_split = 0;
extern int g_mode2w;      // gemm2w.hip
extern int g_mode_small;  // gemm_small.hip
static unsigned long long g_stamps = 0;
extern "C" int morec_tuning_set(const char* key, int value) {
    if (!key) return MOREC_E_ARG;
    if (!strcmp(key, "deterministic")) { morec_set_deterministic(value); return MOREC_OK; }
    if (!strcmp(key, "gemm8p")) { g_mode8p = value; return MOREC_OK; }
    if (!strcmp(key, "gemm2w")) { g_mode2w = value; return MOREC_OK; }
    if (!strcmp(key, "gemm_small")) { g_mode_small = value; return MOREC_OK; }
    if (!strcmp(key, "gemm8p_debug")) { g_debug8p = value; return MOREC_OK; }
    if (!strcmp(key, "gemm8p_tail_split")) { g_tail_split = value != 0; return MOREC_OK; }
    if (!strcmp(key, "ce8p")) { g_ce8p_mode = value; return MOREC_OK; }
    if (!strcmp(key, "gemm_skinny")) { g_skinny_mode = value; return MOREC_OK; }
    if (!strcmp(key, "gemm8p_ngroup")) { g_ngroup = value; return MOREC_OK; }
    if (!strcmp(key, "gemm8p_tmr")) { g_tmr_mode = (value == 0 || value == 1 || value == 224 || value == 192) ? value : 1; return MOREC_OK; }
    if (!strcmp(key, "gemm8p_reserve_cus")) { g_reserve_cus = value < 0 ? 0 : value > 128 ? 128 : value; return MOREC_OK; }
    if (!strcmp(key, "gemm8p_tail_bias")) { g_tail_bias = value < 0 ? 0 : value > 16 ? 16 : value; return MOREC_OK; }
    // device buffer (16 x 8 bytes per workgroup) that receives s_memtime stamps of wave 0: address in two halves
    if (!strcmp(key, "gemm8p_stamps_lo")) { g_stamps = (g_stamps & 0xffffffff00000000ull) | (unsigned)value; return MOREC_OK; }
    if (!strcmp(key, "gemm8p_stamps_hi")) { g_stamps = (g_stamps & 0xffffffffull) | ((unsigned long long)(unsigned)value << 32); return MOREC_OK; }
    return MOREC_E_UNSUPPORTED;
}

int gemm8p_mode() {
    if (g_mode8p < 0) {
        const char* e = getenv("MOREC_GEMM8P");
        g_mode8p = e ? atoi(e) : 0;
        if (const char* d = getenv("MOREC_GEMM8P_DEBUG")) g_debug8p = atoi(d);     // ablation bits for whole-step A/B runs
        if (const char* t = getenv("MOREC_GEMM8P_TAIL_SPLIT")) g_tail_split = atoi(t) != 0;
        if (const char* n = getenv("MOREC_GEMM8P_NGROUP")) g_ngroup = atoi(n);
        if (const char* c = getenv("MOREC_CE8P")) g_ce8p_mode = atoi(c);
        if (const char* r = getenv("MOREC_GEMM8P_RESERVE_CUS")) { const int v = atoi(r); g_reserve_cus = v < 0 ? 0 : v > 128 ? 128 : v; }
    }
    return g_mode8p;
}

template <typename TI>
static int dispatch8p(const morec_gemm_desc* d, GemmArgs& a, int mode, hipStream_t s) {
    // tile height (see pick_tmr): the PART instantiations exist for the three kernels of the encoder step -- plain, FFN-up + GELU (+ act'),
    // x act' + column sums -- with 16-bit outputs
    a.tmr = TM;
    if (d->out_dtype == h16<TI>::dtype && ((mode == 0 && !a.colsum) || (mode == 1 && !a.colsum) || (mode == 5 && a.colsum))) {
        const int t = pick_tmr(d->M, (d->N + TN - 1) / TN, (d->K + KE - 1) / KE, gemm8p_cu_count());
        if (t == 224) {
            if (mode == 0) return launch8p<TI, TI, 0, false, 224>(d, a, s);
            if (mode == 1) return launch8p<TI, TI, 1, false, 224>(d, a, s);
            return launch8p<TI, TI, 5, true, 224>(d, a, s);
        }
        if (t == 192) {
            if (mode == 0) return launch8p<TI, TI, 0, false, 192>(d, a, s);
            if (mode == 1) return launch8p<TI, TI, 1, false, 192>(d, a, s);
            return launch8p<TI, TI, 5, true, 192>(d, a, s);
        }
    }
    if (d->out_dtype == MOREC_F32) {
        if (mode != 0 || a.colsum) return G8_NOT_TAKEN;
        return launch8p<TI, float, 0, false>(d, a, s);
    }
    if (d->out_dtype != h16<TI>::dtype) return G8_NOT_TAKEN;
    if (a.colsum) {
        if (mode == 3) return launch8p<TI, TI, 3, true>(d, a, s);
        if (mode == 4) return launch8p<TI, TI, 4, true>(d, a, s);
        if (mode == 5) return launch8p<TI, TI, 5, true>(d, a, s);
        return G8_NOT_TAKEN;
    }
    switch (mode) {
        case 1: return launch8p<TI, TI, 1, false>(d, a, s);
        case 2: return launch8p<TI, TI, 2, false>(d, a, s);
        case 3: return launch8p<TI, TI, 3, false>(d, a, s);
        case 4: return launch8p<TI, TI, 4, false>(d, a, s);
        case 5: return launch8p<TI, TI, 5, false>(d, a, s);
        default: return launch8p<TI, TI, 0, false>(d, a, s);
    }
}

int gemm8p_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    (void)gemm8p_mode();
    if (g_mode8p == 1) return G8_NOT_TAKEN;
    if (!is_h16(d->in_dtype) || a.accumulate != 0 || !a.vec_store || d->split_k > 1) return G8_NOT_TAKEN;
    if (d->K % 8 || d->K <= KE || d->N < 64 || d->M < 1) return G8_NOT_TAKEN;      // 16-byte slots; at least two K-tiles (the last may be partial)
    const long tiles = (long)((d->M + TM - 1) / TM) * ((d->N + TN - 1) / TN);
    if (a.colsum && d->M < 128) return G8_NOT_TAKEN;      // partial-row workspace is sized per 64 rows
    // automatic: enough tiles to fill the 256 CUs, and at most a quarter of the tile columns past N (N = 192, 384, 576 of the Swin
    // stages: faster here than in the two-buffer kernel; N = 96 is not -- profiles/r02_swin_gemm_shapes_modes.txt)
    // (long-K problems already from 160 tiles: the scoring backward's dE = dl^T P, 168 tiles of 40 K-tiles each)
    if (g_mode8p != 2 && (tiles < (d->K >= 2048 ? 160 : 192) || ((d->N + TN - 1) / TN) * TN * 100L > d->N * 134L)) return G8_NOT_TAKEN;
    if (d->dact != MOREC_ACT_NONE && a.bias) return G8_NOT_TAKEN;      // no bias register set in the derivative epilogues
    const int mode = d->dact == MOREC_ACT_GELU ? 3 : d->dact == MOREC_ACT_RELU ? 4 : d->dact == MOREC_DACT_MUL ? 5
                     : d->act == MOREC_ACT_GELU ? 1 : d->act == MOREC_ACT_RELU ? 2 : 0;
    a.debug = g_debug8p;
    a.tail_bias = g_tail_bias;
    a.tail_split = g_tail_split;
    a.stamps = reinterpret_cast<unsigned long long*>(g_stamps);
    if (d->in_dtype == MOREC_F16) return dispatch8p<f16>(d, a, mode, s);
    return dispatch8p<bf16>(d, a, mode, s);
}
