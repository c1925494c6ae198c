// gemm4w.hip -- EXPERIMENT, not part of the library build (it was compiled as idvs/morec_amd/csrc/gemm4w.hip behind gemm8p_try_launch for
// the linear epilogue; gemm_args.hpp declared gemm4w_try_launch).  The architecture of the vendor library's hand-written kernel
// (Custom_Cijk_..._MT256x256x64_MI16x16x1: 256 threads, 128 x 128 outputs per wave, LDS-DMA into two 64-KiB K-tile buffers, 32 ds_read_b128 +
// 16 DMA per wave and K-tile): ONE wave per SIMD on v_mfma_f32_16x16x32_bf16, 8 x 8 accumulator blocks of 16 x 16 (256 AGPRs), two barriers
// per K-tile, every LDS read / DMA issued in the shadow of the wave's own MFMAs (sched_group_barrier: two MFMA, one other).
// Correct on every shape of scripts/gemm8p_check.py.  Measured (1 x MI355X, power-limited at 1.38 kW):
//      main-loop-dominated 4096 x 4096 x 8192:  this kernel 1.31 PFLOP/s (1.27 with v_mfma_f32_32x32x16_bf16 and 4 x 4 blocks of 32 x 32),
//                                               gemm8p 1.25-1.34, vendor library 1.45-1.57
//      the step's shapes (M = 51200):           qkv 188 us (gemm8p 170), fc2 255 (214), d_qkv 194 (164): the eight-wave kernel hides
//                                               prologue and epilogue far better (two wave rows, counted waits under the next tile).
// So neither the wave tile (a third less LDS traffic), nor the barrier count (2 vs 8), nor the MFMA shape explains the vendor kernel's
// 15 % -- what is left is its hand-scheduled instruction stream.  Kept as the starting point for that work.
//
// Pipeline (K-tile t lives in buffer t & 1; an iteration = one 32-deep MFMA k-step = 64 MFMA per wave):
//      even iteration 2t    : read the second k-step's fragments | 32 MFMA | lgkmcnt(0), s_barrier: buffer dead -> issue K-tile t + 2 | 32 MFMA
//      odd  iteration 2t + 1: 32 MFMA | vmcnt(16): K-tile t + 1 landed, s_barrier | read its first k-step's fragments | 32 MFMA
#include <stdlib.h>
#include "gemm_core.hpp"
#include "gemm_args.hpp"

namespace {
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr int TM = 256, TN = 256, KE = 64;       // tile; K elements per K-tile
constexpr int KB = 2 * KE;                       // bytes of K per row per K-tile
constexpr int OPB = 256 * KB;                    // one operand of one K-tile: 32 KiB
constexpr int BUFB = 2 * OPB;                    // 64 KiB
constexpr int LDS_RING = 2 * BUFB;               // two K-tile buffers: 128 KiB
constexpr int SLICE = 4096;                      // epilogue: one [32 rows][128 B] slice per wave
constexpr int LDS_TOTAL = LDS_RING + 4 * SLICE;
constexpr int THREADS = 256;
constexpr int GROUP = 16;                        // DMA instructions per wave per K-tile

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lgkm_wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void bar() {
    pin();
    __builtin_amdgcn_s_barrier();
    pin();
}
__device__ __forceinline__ uint4 lds16(const char* p) { return *reinterpret_cast<const uint4*>(p); }

struct Ctx4 {
    __amdgpu_buffer_rsrc_t ra, rb;   // descriptors of A / B, based at the tile's first row
    uint32_t va[8], vb[8];           // per-lane source byte offsets of the wave's eight 8-row pieces of A / of B
    int dpa, dpb;                    // wave-uniform LDS offsets (within a buffer) of the wave's first A / B piece
    int la[2], lb[2];                // per-lane fragment offsets (within a buffer) of the 32-deep MFMA k-step 0 / 1: A rows of wave row wr, B rows of wave column wc
};

__device__ __forceinline__ void make_ctx4(Ctx4& c, int tid, const bf16* At, const bf16* Bt, int rows_a, int rows_b, int lda, int ldb) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // DMA geometry (as gemm8p.hip): a piece = 8 rows x 128 B = one instruction; lane -> row (lane >> 3) of the piece, physical slot lane & 7;
    // this wave owns rows [64 wave, 64 wave + 64) of both operands
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int rl = j * 8 + (lane >> 3);
        const int row = wave * 64 + rl;
        const int slot = (lane & 7) ^ ((rl >> 1) & 7);       // (row >> 1) & 7: the wave's base is a multiple of 16 rows
        c.va[j] = (uint32_t)min(row, rows_a - 1) * (uint32_t)(lda * 2) + slot * 16;
        c.vb[j] = (uint32_t)min(row, rows_b - 1) * (uint32_t)(ldb * 2) + slot * 16;
    }
    const long abytes = (long)min(256, rows_a) * lda * 2, bbytes = (long)min(256, rows_b) * ldb * 2;
    c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)At, 0, (int)min(abytes, 0x7fffffffL), 0x00020000);
    c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bt, 0, (int)min(bbytes, 0x7fffffffL), 0x00020000);
    c.dpa = wave * 64 * KB;
    c.dpb = OPB + wave * 64 * KB;
    const int c15 = lane & 15, g = lane >> 4, fr = (c15 >> 1) & 7;      // row blocks start at multiples of 16: (row >> 1) & 7 = (c15 >> 1) & 7
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ps = (4 * ks + g) ^ fr;
        c.la[ks] = (wr * 128 + c15) * KB + (ps << 4);
        c.lb[ks] = OPB + (wc * 128 + c15) * KB + (ps << 4);
    }
}

// the wave's sixteen pieces of one K-tile; kbyte = byte offset of the K-tile's first column within a row
__device__ __forceinline__ void issue_ktile(const Ctx4& c, char* buf, int kbyte) {
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(c.ra, (lptr_t)(buf + c.dpa + j * 1024), 16, c.va[j], kbyte, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rb, (lptr_t)(buf + c.dpb + j * 1024), 16, c.vb[j], kbyte, 0, 0);
}

struct Frags {
    uint4 a[8], b[8];     // one 32-deep k-step: A rows mi * 16 + (lane & 15), B rows ni * 16 + (lane & 15); lane group g holds k = 8 g .. 8 g + 7
};
__device__ __forceinline__ void read_frags(Frags& f, const char* buf, const Ctx4& c, int ks) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f.a[i] = lds16(buf + c.la[ks] + i * (16 * KB));
#pragma unroll
    for (int i = 0; i < 8; ++i) f.b[i] = lds16(buf + c.lb[ks] + i * (16 * KB));
}
// 32 MFMA: accumulator blocks mi = M0 .. M0 + 3 (all eight ni).  B fragment as the instruction's first operand: the lane ends up
// owning ONE m (mi * 16 + (lane & 15)) and runs of 4 consecutive n (ni * 16 + 4 (lane >> 4) + r).
template <bool ZERO, int M0>
__device__ __forceinline__ void mfma32(f32x4_t (&acc)[8][8], const Frags& f) {
#pragma unroll
    for (int mi = M0; mi < M0 + 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            f32x4_t cin = acc[mi][ni];
            if constexpr (ZERO) cin = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f.b[ni]), __builtin_bit_cast(bf16x8_t, f.a[mi]), cin, 0, 0, 0);
        }
}

// One wave per SIMD has no partner whose MFMAs cover its LDS / DMA issue: every other instruction has to be issued in the shadow
// of an MFMA of the SAME wave (32 cycles each).  sched_group_barrier pins the interleave: one MFMA, one LDS read (or one
// LDS-DMA), sixteen times.
template <int OTHER>     // 0x100: DS read, 0x020: VMEM read
__device__ __forceinline__ void interleave16() {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(OTHER, 1, 0);
    }
}

template <bool ZERO, bool ISSUE>
__device__ __forceinline__ void iter_even(char* cur, const Ctx4& c, int kb_issue, f32x4_t (&acc)[8][8], const Frags& fc, Frags& fn) {
    read_frags(fn, cur, c, 1);
    mfma32<ZERO, 0>(acc, fc);
    interleave16<0x100>();
    pin();
    lgkm_wait0();            // this wave's reads of the buffer are done
    bar();                   // ... everybody's
    if constexpr (ISSUE) issue_ktile(c, cur, kb_issue);
    mfma32<ZERO, 4>(acc, fc);
    if constexpr (ISSUE) interleave16<0x020>();
    pin();
}
// Odd iteration: second k-step of the K-tile (fc = its fragments, read during the even iteration).  WAIT: the vmcnt that retires the
// NEXT K-tile (in buffer `oth`), whose first k-step's fragments are read behind the barrier.
template <int WAIT, bool NEXT>
__device__ __forceinline__ void iter_odd(char* oth, const Ctx4& c, f32x4_t (&acc)[8][8], const Frags& fc, Frags& fn) {
    mfma32<false, 0>(acc, fc);
    if constexpr (NEXT) {
        pin();
        vm_wait<WAIT>();
        bar();
        read_frags(fn, oth, c, 0);
    }
    mfma32<false, 4>(acc, fc);
    if constexpr (NEXT) interleave16<0x100>();
    pin();
}

// acc = A-panel . B-panel^T over nk K-tiles (nk >= 3), K-tiles 0 and 1 already requested.
__device__ __forceinline__ void mainloop4w(const Ctx4& c, int nk, char* smem, f32x4_t (&acc)[8][8]) {
    Frags f0, f1;
    vm_wait<GROUP>();          // K-tile 0 (younger stores of the previous epilogue only make this stricter)
    bar();
    read_frags(f0, smem, c, 0);
    char* cur = smem;
    char* oth = smem + BUFB;
    iter_even<true, true>(cur, c, 2 * KB, acc, f0, f1);
    iter_odd<GROUP, true>(oth, c, acc, f1, f0);
    int t = 1;
    for (; t < nk - 2; ++t) {
        char* x = cur; cur = oth; oth = x;
        iter_even<false, true>(cur, c, (t + 2) * KB, acc, f0, f1);
        iter_odd<GROUP, true>(oth, c, acc, f1, f0);
    }
    {   // K-tile nk - 2: nothing left to request; K-tile nk - 1 is the only group in flight
        char* x = cur; cur = oth; oth = x;
        iter_even<false, false>(cur, c, 0, acc, f0, f1);
        iter_odd<0, true>(oth, c, acc, f1, f0);
    }
    {
        char* x = cur; cur = oth; oth = x;
        iter_even<false, false>(cur, c, 0, acc, f0, f1);
        iter_odd<0, false>(oth, c, acc, f1, f0);
    }
    bar();                     // every wave is done reading (iter_even's wait + this barrier): the buffers are free for the next tile's prologue
}

struct Tile4 {
    int wg, m0, n0;
};

template <typename TO>
__device__ __forceinline__ void tile_body4(const GemmArgs& p, char* smem, int nk, const bf16* __restrict__ Acur, const bf16* __restrict__ Bcur,
                                           const bf16* __restrict__ Anext, const bf16* __restrict__ Bnext, const Tile4 cur, const Tile4 nxt,
                                           const bool first) {
    static_assert(sizeof(TO) == 2, "bf16 outputs");
    const int m0 = cur.m0, n0 = cur.n0;
    f32x4_t acc[8][8];
    float bias_l[2];
    {
        int tid_m = threadIdx.x;
        asm volatile("" : "+v"(tid_m));
        {   // any valid address when there is no bias (discarded in the epilogue): no branch around a load
            const float* bp = p.bias ? p.bias : reinterpret_cast<const float*>(p.B);
            const int nb = n0 + ((tid_m >> 6) & 1) * 128 + (tid_m & 63);
            bias_l[0] = bp[min(nb, p.N - 1)];
            bias_l[1] = bp[min(nb + 64, p.N - 1)];
        }
        Ctx4 c;
        make_ctx4(c, tid_m, Acur, Bcur, p.M - m0, p.N - n0, p.lda, p.ldb);
        if (first) {
            issue_ktile(c, smem, 0);
            issue_ktile(c, smem + BUFB, KB);
        }
        mainloop4w(c, nk, smem, acc);
    }
    // ---- epilogue (wave-private; see gemm8p.hip): acc[mi][ni][r] = C[m0 + wr*128 + mi*16 + (lane & 15)][n0 + wc*128 + ni*16 + 4 (lane >> 4) + r].
    // A pass moves 32 rows (two mi) x 64 columns (four ni) through the wave's [32 rows][128 B] slice.
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane = tid_e & 63, c15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid_e >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    char* ws = smem + LDS_RING + wave * SLICE;
    TO* C = reinterpret_cast<TO*>(p.C);
    auto wfence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int rs_row = lane >> 3, rs_slot = lane & 7;
    const int rs_off = rs_row * 128 + ((rs_slot ^ (rs_row & 7)) << 4);
    // row `row` of the slice, columns 16 nq + 4 g .. + 3 of the pass: 16-byte slot 2 nq + (g >> 1), second half of the slot for odd g
    auto cell = [&](int row, int nq) { return reinterpret_cast<TO*>(ws + row * 128 + (((2 * nq + (g >> 1)) ^ (row & 7)) << 4) + 8 * (g & 1)); };
    const int nw = n0 + wc * 128;
    const long tile_bytes = (long)min(256, p.M - m0) * p.ldc * 2;
    const int ext = (int)min(tile_bytes, 0x7fffffffL);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(C + (size_t)m0 * p.ldc), 0, ext, 0x00020000);
    uint32_t lo[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int n = nw + hf * 64 + rs_slot * 8;
        lo[hf] = n < p.N ? (uint32_t)((rs_row * p.ldc + n) * 2) : 0x80000000u;
    }
    // bias of the lane's 8 column groups through the slice (fetched ahead of the main loop)
    float4 bv[8];
    {
        reinterpret_cast<float*>(ws)[lane] = p.bias ? bias_l[0] : 0.f;
        reinterpret_cast<float*>(ws)[64 + lane] = p.bias ? bias_l[1] : 0.f;
        wfence();
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) bv[ni] = *reinterpret_cast<const float4*>(ws + (ni * 16 + 4 * g) * 4);
        wfence();
    }
    pin();
    {   // next tile: its first two K-tiles fly under this tile's epilogue (issued unconditionally: see gemm8p.hip)
        int tid_n = threadIdx.x;
        asm volatile("" : "+v"(tid_n));
        Ctx4 cn;
        make_ctx4(cn, tid_n, Anext, Bnext, p.M - nxt.m0, p.N - nxt.n0, p.lda, p.ldb);
        issue_ktile(cn, smem, 0);
        issue_ktile(cn, smem + BUFB, KB);
    }
#pragma unroll
    for (int mp = 0; mp < 4; ++mp)          // pairs of 16-row blocks
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            pin();
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int nq = 0; nq < 4; ++nq) {
                    const int mi = 2 * mp + mh, ni = 4 * hf + nq;
                    const float4 b = bv[ni];
                    float v[4];
                    v[0] = fmaf(acc[mi][ni][0], p.alpha, b.x);
                    v[1] = fmaf(acc[mi][ni][1], p.alpha, b.y);
                    v[2] = fmaf(acc[mi][ni][2], p.alpha, b.z);
                    v[3] = fmaf(acc[mi][ni][3], p.alpha, b.w);
                    io<TO>::store4(cell(mh * 16 + c15, nq), v);
                }
            wfence();
            u32x4_t qv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const u32x4_t*>(ws + rs_off + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(qv[i], rC, lo[hf] + (uint32_t)((wr * 128 + mp * 32 + 8 * i) * p.ldc * 2), 0, 0);
            wfence();
        }
}

template <typename TO>
__global__ __launch_bounds__(THREADS, 1) void gemm4w_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = p.tiles_m * p.tiles_n;
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    const int nk = p.K / KE;
    auto tile_at = [&](int vb) {
        Tile4 t;
        t.wg = xcd_remap(vb, nwg);
        t.m0 = (t.wg / p.tiles_n) * TM;
        t.n0 = (t.wg % p.tiles_n) * TN;
        return t;
    };
    int vb = blockIdx.x;
    Tile4 cur = tile_at(vb);
    bool first = true;
    while (true) {
        vb += gridDim.x;
        const bool more = vb < nwg;
        const Tile4 nxt = more ? tile_at(vb) : cur;
        tile_body4<TO>(p, smem, nk, A + (size_t)cur.m0 * p.lda, B + (size_t)cur.n0 * p.ldb, A + (size_t)nxt.m0 * p.lda,
                       B + (size_t)nxt.n0 * p.ldb, cur, nxt, first);
        if (!more) break;
        cur = nxt;
        first = false;
    }
    vm_wait<0>();      // the trailing prologue must not land in LDS that already belongs to another workgroup
}
}  // namespace

// bf16 in / bf16 out, linear epilogue (alpha, bias).  Returns G8_NOT_TAKEN when the problem is not eligible.
int gemm4w_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("MOREC_GEMM4W"); enabled = e ? atoi(e) : 1; }
    if (!enabled) return G8_NOT_TAKEN;
    if (d->in_dtype != MOREC_BF16 || d->out_dtype != MOREC_BF16 || a.accumulate != 0 || !a.vec_store || d->split_k > 1) return G8_NOT_TAKEN;
    if (d->act != MOREC_ACT_NONE || d->dact != MOREC_ACT_NONE || a.aux_out || a.colsum) return G8_NOT_TAKEN;
    if (d->K % KE || d->K < 4 * KE || d->N < 64) return G8_NOT_TAKEN;
    a.tiles_m = (d->M + TM - 1) / TM;
    a.tiles_n = (d->N + TN - 1) / TN;
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        n_cu &= ~7;
        if (n_cu < 8) n_cu = 8;
        attr_set = true;
    }
    const int nwg = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL((gemm4w_kernel<bf16>), dim3(nwg < n_cu ? nwg : n_cu), dim3(THREADS), LDS_TOTAL, s, a);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
