// gemm_tn8p.hip -- weight-gradient GEMM  C[n, k] += sum_m DY[m, n] * X[m, k]  (dW = dY^T X of every nn.Linear on the path) on the
// eight-phase schedule of gemm8p.hip: 256 (n) x 256 (k) fp32 output tile, contraction advanced 64 TOKENS per stage, eight waves
// as 2 (n) x 4 (k) with 128 x 64 outputs each on v_mfma_f32_32x32x16_bf16, a stage cut in four phases
//      { ds_read_b64_tr_b16 sub-tile | one 16-KiB LDS-DMA half-tile | counted vmcnt }  s_barrier  { 8 MFMA }  s_barrier
// with the two wave rows one barrier apart and the DMA queue never drained (see gemm8p.hip for the protocol and its proof
// obligations: refill two phases after the last read, read one phase after the wait + barrier that retire a refill).
// Replaces (for the large split-m launches of the backward pass) the two-buffer loop of gemm_tn.hip.
//
// Both operands are contracted over their ROW index, so the MFMA fragments are COLUMNS of the row-major global tiles.  The
// tiles are staged token-major as they lie in memory and transposed by the LDS read (ds_read_b64_tr_b16: a 16-lane group
// fetches a [4 tokens x 16 columns] block, lane c receives column c's four tokens).
//
// LDS: two 64-KiB buffers, each [DY: 32 KiB][X: 32 KiB]; an operand part = two halves of [64 tokens][256 B], a half holding
// the 128 columns that ONE phase reads:  DY-first = columns wr*128 + [0, 64) of both wave rows (phase 0), DY-second = the other
// 64 of each (phase 2), X-first / X-second = columns wc*64 + [0, 32) / [32, 64) of the four wave columns (phases 0 / 1).
// 32-byte column block b of token row t is stored at block b ^ 2 (t & 3): the four token rows x two blocks a 32-lane group
// of a transposing read touches then cover all 64 banks once.  (LDS-DMA writes lane-linearly: the permutation is applied to
// the per-lane SOURCE column and to the read address.)
// Tokens past M must contribute ZERO (they are summed): their lanes carry an out-of-range buffer offset.
#include <stdlib.h>
#include "gemm_core.hpp"
#include "gemm_args.hpp"

int gemm8p_mode();      // gemm8p.hip: 0 automatic, 1 never, 2 wherever eligible (tuning key "gemm8p")

namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

constexpr int TT = 64;                            // tokens per stage
constexpr int ROWB = 256;                         // bytes per token row of a half (128 columns)
constexpr int HALF_BYTES = TT * ROWB;             // 16 KiB
constexpr int OP_BYTES = 2 * HALF_BYTES;          // 32 KiB
constexpr int BUF_BYTES = 2 * OP_BYTES;           // 64 KiB
constexpr int LDS_MAIN = 2 * BUF_BYTES;           // 128 KiB
constexpr int SLICE = 4096;
constexpr int LDS_TOTAL = LDS_MAIN + 8 * SLICE;   // + the epilogue's per-wave slices
constexpr int THREADS = 512;

struct TnArgs8 {
    const bf16* DY;
    const bf16* X;
    float* out;        // slab base ([zs][N][K]) or C itself when there is one chunk
    int M, N, K, ldy, ldx, ldo, mchunk, tiles_n, tiles_k, zs;
    size_t slab_stride;
};

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
template <int REM, int W1, int W0>
__device__ __forceinline__ void vm_wait_tail() {
    vm_wait<(REM >= 2 ? 8 : (REM == 1 ? W1 : W0))>();
}

struct Ctx {
    __amdgpu_buffer_rsrc_t ry, rx;        // descriptors of DY / X, based at the chunk's first token row and the tile's first column
    uint32_t y1[2], y2[2], x1[2], x2[2];  // per-lane byte offsets (token row within the stage, source column) of the wave's two pieces
    int tok[2];                           // token (within the stage) of this lane in piece j: rows past the end are poisoned per stage
    int dpiece;                           // LDS offset of the wave's first piece within a half
    int ay[2], ax;                        // per-lane fragment base offsets: DY column group 0 / 1 of this wave, X column group
    uint32_t vmask;                       // SKIP kernels: the wave's output blocks inside the matrix (mfma_quadrant)
};

// two pieces (4 token rows each) of one half-tile; `left` = tokens that exist from the stage's first row on
__device__ __forceinline__ void dma2(__amdgpu_buffer_rsrc_t rs, const uint32_t (&o)[2], const int (&tok)[2], int left, int soff, char* dst) {
    const uint32_t o0 = tok[0] < left ? o[0] : 0x80000000u;
    const uint32_t o1 = tok[1] < left ? o[1] : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, o0, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + 1024), 16, o1, soff, 0, 0);
}

// fragment of k-step ks (16 tokens) from a half: two transposing reads (tokens 8h + 0..3 and 8h + 4..7 of the step)
__device__ __forceinline__ bf16x8_t tr_frag(const char* smem, int addr, int ks) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + addr + (16 * ks) * ROWB));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + addr + (16 * ks + 4) * ROWB));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// SKIP (outputs narrower than the tile: Swin's 96 ... 576-wide weights): vm = wave-uniform mask of the wave's 32 x 32 output blocks that lie
// inside the matrix (bits 0-3: n-blocks, bits 4-5: k-blocks); a block outside it costs no MFMA.  A [96, 384] weight gradient fills 36 of
// the 128 blocks of its two 256 x 256 tiles: unskipped, the launch was bound by MFMAs on padding (578 GFLOP for 163), not by its 2.1 GB.
template <typename T16, int N0, int KQ, bool SKIP>
__device__ __forceinline__ void mfma_quadrant(f32x16_t (&acc)[4][2], const bf16x8_t (&fy)[2][4], const bf16x8_t (&fx)[4], uint32_t vm) {
    if constexpr (SKIP) {
        if (!(vm & (16u << KQ))) return;
        const bool a = (vm >> N0) & 1u, b = (vm >> (N0 + 1)) & 1u;
        if (!(a && b)) {
            if (!(a || b)) return;
            __builtin_amdgcn_s_setprio(1);
            if (a) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[N0][KQ] = h16<T16>::mma32(fx[ks], fy[0][ks], acc[N0][KQ]);
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[N0 + 1][KQ] = h16<T16>::mma32(fx[ks], fy[1][ks], acc[N0 + 1][KQ]);
            }
            __builtin_amdgcn_s_setprio(0);
            return;
        }
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)      // X fragment as the instruction's A operand: the lane owns ONE n and runs of 4 consecutive k
            acc[N0 + ni][KQ] = h16<T16>::mma32(fx[ks], fy[ni][ks], acc[N0 + ni][KQ]);
    __builtin_amdgcn_s_setprio(0);
}

// One stage (64 tokens) out of the buffer at byte offset cb.  mo = byte offset (rows x pitch) of this stage's first token in DY
// (mo * ldx / ldy for X is passed separately); left = tokens of the chunk from this stage's first row on.
template <typename T16, int REM, bool SKIP>
__device__ __forceinline__ void stage(char* smem, const Ctx& c, int cb, int soy, int sox, int dy_step, int dx_step, int left,
                                      f32x16_t (&acc)[4][2]) {
    char* cur = smem + cb;
    char* oth = smem + (cb ^ BUF_BYTES);
    bf16x8_t fy[2][4], fx0[4], fx1[4];
    const int ay0 = cb + c.ay[0], ay1 = cb + c.ay[1], ax = cb + OP_BYTES + c.ax;
    // ---- phase 0: X-first + DY-first fragments; refill X-second of stage t+1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fx0[ks] = tr_frag(smem, ax, ks);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fy[0][ks] = tr_frag(smem, ay0, ks);
        fy[1][ks] = tr_frag(smem, ay1, ks);
    }
    if constexpr (REM >= 1) dma2(c.rx, c.x2, c.tok, left - TT, sox + dx_step, oth + OP_BYTES + HALF_BYTES + c.dpiece);
    pin();
    vm_wait_tail<REM, 8, 2>();
    bar();
    mfma_quadrant<T16, 0, 0, SKIP>(acc, fy, fx0, c.vmask);
    bar();
    // ---- phase 1: X-second fragments; refill DY-second of t+1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fx1[ks] = tr_frag(smem, ax + HALF_BYTES, ks);
    if constexpr (REM >= 1) dma2(c.ry, c.y2, c.tok, left - TT, soy + dy_step, oth + HALF_BYTES + c.dpiece);
    pin();
    vm_wait_tail<REM, 8, 0>();
    bar();
    mfma_quadrant<T16, 0, 1, SKIP>(acc, fy, fx1, c.vmask);
    bar();
    // ---- phase 2: DY-second fragments; refill DY-first of t+2 (this buffer)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fy[0][ks] = tr_frag(smem, ay0 + HALF_BYTES, ks);
        fy[1][ks] = tr_frag(smem, ay1 + HALF_BYTES, ks);
    }
    if constexpr (REM >= 2) dma2(c.ry, c.y1, c.tok, left - 2 * TT, soy + 2 * dy_step, cur + c.dpiece);
    pin();
    vm_wait_tail<REM, 6, 0>();
    bar();
    mfma_quadrant<T16, 2, 1, SKIP>(acc, fy, fx1, c.vmask);
    bar();
    // ---- phase 3: nothing to read (X-first is still in registers); refill X-first of t+2
    if constexpr (REM >= 2) dma2(c.rx, c.x1, c.tok, left - 2 * TT, sox + 2 * dx_step, cur + OP_BYTES + c.dpiece);
    pin();
    vm_wait_tail<REM, 4, 0>();
    bar();
    mfma_quadrant<T16, 2, 0, SKIP>(acc, fy, fx0, c.vmask);
    bar();
}

// The panel pointers are __restrict__ for hipcc's s_waitcnt insertion (see gemm8p.hip::tile_body): it tags the LDS-DMA
// instructions with alias scopes and every ds_read with "does not alias them"; untagged, each ds_read after an LDS-DMA gets
// a full `s_waitcnt vmcnt(0)`.
template <typename T16, bool SKIP>
__device__ __forceinline__ void tn_body(const TnArgs8& p, char* smem, const bf16* __restrict__ DYt, const bf16* __restrict__ Xt, int n0, int k0,
                                        int mbeg, int mend, float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nt = (mend - mbeg + TT - 1) / TT;          // >= 2 (launcher)
    Ctx c;
    {
        // DMA geometry: piece j of a half = 4 token rows; lane -> row (lane >> 4) of the piece, physical 16-byte chunk lane & 15
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = 8 * wave + 4 * j + (lane >> 4);                 // token row within the stage
            const int lc = (lane & 15) ^ (4 * (t & 3));                   // logical chunk stored at this physical chunk
            const int ycol = (lc < 8 ? lc * 8 : 128 + (lc - 8) * 8);      // DY-first source column (DY-second: + 64)
            const int xcol = (lc >> 2) * 64 + (lc & 3) * 8;               // X-first source column (X-second: + 32)
            c.tok[j] = t;
            // columns past N / K belong to output columns that are never stored: clamp them into the row
            c.y1[j] = (uint32_t)(t * p.ldy + min(n0 + ycol, p.N - 8) - n0) * 2u;
            c.y2[j] = (uint32_t)(t * p.ldy + min(n0 + ycol + 64, p.N - 8) - n0) * 2u;
            c.x1[j] = (uint32_t)(t * p.ldx + min(k0 + xcol, p.K - 8) - k0) * 2u;
            c.x2[j] = (uint32_t)(t * p.ldx + min(k0 + xcol + 32, p.K - 8) - k0) * 2u;
        }
        const long ybytes = (long)(p.M - mbeg) * p.ldy * 2, xbytes = (long)(p.M - mbeg) * p.ldx * 2;
        c.ry = __builtin_amdgcn_make_buffer_rsrc((void*)DYt, 0, (int)min(ybytes, 0x7fffffffL), 0x00020000);
        c.rx = __builtin_amdgcn_make_buffer_rsrc((void*)Xt, 0, (int)min(xbytes, 0x7fffffffL), 0x00020000);
        c.dpiece = 8 * wave * ROWB;
        // fragment addressing: lane -> column (lane & 15) of 16-column block cgrp = (lane >> 4) & 1 of its 32-column group,
        // token sub-row rr = (lane & 15) >> 2 ... of the 4-token block; (lane & 3) * 8 bytes = its 4-column slice of the block row
        const int rr = (lane & 15) >> 2, cgrp = (lane >> 4) & 1, h = lane >> 5;
        const int rowoff = (8 * h + rr) * ROWB + 8 * (lane & 3);
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) c.ay[cg] = rowoff + (((wr * 4 + cg * 2 + cgrp) ^ (2 * rr)) << 5);
        c.ax = rowoff + (((wc * 2 + cgrp) ^ (2 * rr)) << 5);
        uint32_t vm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) vm |= (n0 + wr * 128 + i * 32 < p.N) ? (1u << i) : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j) vm |= (k0 + wc * 64 + j * 32 < p.K) ? (16u << j) : 0u;
        c.vmask = __builtin_amdgcn_readfirstlane(vm);
    }
    const int dy_step = TT * p.ldy * 2, dx_step = TT * p.ldx * 2;
    const int left0 = mend - mbeg;
    // prologue: stage 0 entirely, the first halves of stage 1
    dma2(c.ry, c.y1, c.tok, left0, 0, smem + c.dpiece);
    dma2(c.rx, c.x1, c.tok, left0, 0, smem + OP_BYTES + c.dpiece);
    dma2(c.rx, c.x2, c.tok, left0, 0, smem + OP_BYTES + HALF_BYTES + c.dpiece);
    dma2(c.ry, c.y2, c.tok, left0, 0, smem + HALF_BYTES + c.dpiece);
    dma2(c.ry, c.y1, c.tok, left0 - TT, dy_step, smem + BUF_BYTES + c.dpiece);
    dma2(c.rx, c.x1, c.tok, left0 - TT, dx_step, smem + BUF_BYTES + OP_BYTES + c.dpiece);
    pin();

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    vm_wait<8>();
    bar();
    if (wr == 1) bar();
    int cb = 0, t = 0;
    for (; t < nt - 2; ++t) {
        stage<T16, 2, SKIP>(smem, c, cb, t * dy_step, t * dx_step, dy_step, dx_step, left0 - t * TT, acc);
        cb ^= BUF_BYTES;
    }
    stage<T16, 1, SKIP>(smem, c, cb, t * dy_step, t * dx_step, dy_step, dx_step, left0 - t * TT, acc);
    stage<T16, 0, SKIP>(smem, c, cb ^ BUF_BYTES, (t + 1) * dy_step, (t + 1) * dx_step, dy_step, dx_step, left0 - (t + 1) * TT, acc);
    if (wr == 0) bar();

    // ---- epilogue: acc[Ni][Ki][4 g + r] = C[n0 + wr*128 + Ni*32 + (lane & 31)][k0 + wc*64 + Ki*32 + 8 g + 4 (lane >> 5) + r], fp32.
    // Each 32 x 32 block goes through the wave's 4-KiB LDS slice (16-byte slots XOR-swizzled with row & 7) so that the global
    // stores are 16-byte lanes along rows (128 contiguous bytes per row per instruction).
    const int r5 = lane & 31, hh = lane >> 5;
    char* ws = smem + LDS_MAIN + wave * SLICE;
    const int rs_row = lane >> 3, rs_slot = lane & 7;
    const int rs_off = rs_row * 128 + ((rs_slot ^ (rs_row & 7)) << 4);
    const int nrows = min(256, p.N - n0);
    const long obytes = (long)nrows * p.ldo * 4;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (size_t)n0 * p.ldo), 0, (int)min(obytes, 0x7fffffffL), 0x00020000);
    auto wfence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
#pragma unroll
    for (int Ni = 0; Ni < 4; ++Ni) {
#pragma unroll
        for (int Ki = 0; Ki < 2; ++Ki) {
            pin();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int slot = 2 * g + hh;
                *reinterpret_cast<float4*>(ws + r5 * 128 + ((slot ^ (r5 & 7)) << 4)) =
                    make_float4(acc[Ni][Ki][4 * g], acc[Ni][Ki][4 * g + 1], acc[Ni][Ki][4 * g + 2], acc[Ni][Ki][4 * g + 3]);
            }
            wfence();
            u32x4_t q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const u32x4_t*>(ws + rs_off + i * 1024);
            const int kcol = k0 + wc * 64 + Ki * 32 + rs_slot * 4;
            const uint32_t lo = kcol < p.K ? (uint32_t)((rs_row * p.ldo + kcol) * 4) : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(q[i], ro, lo + (uint32_t)((wr * 128 + Ni * 32 + 8 * i) * p.ldo * 4), 0, 0);
            wfence();
        }
    }
}

template <typename T16, bool SKIP>
__global__ __launch_bounds__(THREADS) void gemm_tn8p_kernel(TnArgs8 p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Work items in (token chunk, n-tile, k-tile) order, k fastest, dealt to the XCDs in CONTIGUOUS runs (xcd_remap over the whole
    // launch): the tiles of one chunk -- which share its DY panel (across k-tiles) and its X panel (across n-tiles) and walk the
    // tokens in lockstep -- then sit behind ONE L2.  With the chunk on the grid's z axis they were spread over all eight XCDs and
    // every L2 fetched its own copy of the panels: FETCH_SIZE 778 MB per launch against 364 MB now (= the operands;
    // profiles/r02e_gemm_pmc.json).  The launch time did not change (the re-fetches were served by the memory-side cache).
    const int tiles = p.tiles_n * p.tiles_k;
    const int item = xcd_remap(blockIdx.x, tiles * p.zs);
    const int z = item / tiles, wg = item - z * tiles;
    const int n0 = (wg / p.tiles_k) * 256, k0 = (wg % p.tiles_k) * 256;
    const int mbeg = z * p.mchunk, mend = min(p.M, mbeg + p.mchunk);
    tn_body<T16, SKIP>(p, smem, p.DY + (size_t)mbeg * p.ldy + n0, p.X + (size_t)mbeg * p.ldx + k0, n0, k0, mbeg, mend,
            p.out + (size_t)z * p.slab_stride);
}
// MOREC_TN8P_SKIP=0: never use the block-skipping variant (A/B aid)
bool skip_env() {
    static const bool on = [] { const char* e = getenv("MOREC_TN8P_SKIP"); return !e || atoi(e) != 0; }();
    return on;
}
}  // namespace

// Runs the launch on the eight-phase kernel when it is eligible: returns MOREC_OK / an error, or G8_NOT_TAKEN.
// `out` = slab workspace ([zs][N][K], fp32) when zs > 1, else C (plain stores: accumulate must be 0).
int gemm_tn8p_try_launch(const bf16* DY, const bf16* X, float* out, size_t slab_stride, int M, int N, int K, int ldy, int ldx, int ldo,
                         int mchunk, int zs, int dtype, hipStream_t s) {
    const int mode = gemm8p_mode();
    if (mode == 1) return G8_NOT_TAKEN;
    if (N % 8 || K % 8 || ldo % 4) return G8_NOT_TAKEN;
    if (mchunk % TT || mchunk < 2 * TT) return G8_NOT_TAKEN;
    if (M - (zs - 1) * (long)mchunk < 2 * TT && zs > 1) return G8_NOT_TAKEN;     // the last chunk needs two stages as well
    if (zs == 1 && M < 2 * TT) return G8_NOT_TAKEN;
    const int tiles_n = (N + 255) / 256, tiles_k = (K + 255) / 256;
    if (mode != 2) {
        // automatic: enough work per workgroup.  (No tile-fill rule: on every Swin-T / BERT weight-gradient shape measured --
        // down to 96 x 96 outputs, 86 % of the tile outside the matrix -- this kernel is at least as fast as the two-buffer one,
        // which is HBM-bound there as well: profiles/r02_swin_gemm_shapes_modes.txt.)
        if (mchunk < 8 * TT) return G8_NOT_TAKEN;
    }
    if ((long)M * ldy * 2 >= 0x7fffffffL || (long)M * ldx * 2 >= 0x7fffffffL) return G8_NOT_TAKEN;   // 32-bit stage offsets
    TnArgs8 a;
    a.DY = DY; a.X = X; a.out = out; a.M = M; a.N = N; a.K = K; a.ldy = ldy; a.ldx = ldx; a.ldo = ldo; a.mchunk = mchunk;
    a.tiles_n = tiles_n; a.tiles_k = tiles_k; a.zs = zs; a.slab_stride = slab_stride;
    if (!by_h16(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn8p_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);   // thread-safe one-time set-up
            static const hipError_t attr_rc2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn8p_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
            (void)attr_rc; (void)attr_rc2;
            // a 32-wide block row or column of some tile lies outside the matrix: the variant that skips those blocks' MFMAs
            if (skip_env() && ((N % 256 != 0 && N % 256 <= 224) || (K % 256 != 0 && K % 256 <= 224)))
                hipLaunchKernelGGL((gemm_tn8p_kernel<T, true>), dim3(tiles_n * tiles_k * zs), dim3(THREADS), LDS_TOTAL, s, a);
            else
                hipLaunchKernelGGL((gemm_tn8p_kernel<T, false>), dim3(tiles_n * tiles_k * zs), dim3(THREADS), LDS_TOTAL, s, a);
        }))
        return G8_NOT_TAKEN;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
