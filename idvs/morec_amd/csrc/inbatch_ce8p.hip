// inbatch_ce8p.hip -- the scoring kernels on the eight-phase 256 x 256 main loop (gemm8p_core.hpp), bf16.
//
// Reference arithmetic: T/model/model.py:32-33,45-67 (see inbatch_ce.hip, whose 128 x 128 kernels keep serving fp32, small and
// ragged problems).  What changes here is the SHAPE of the computation, for the pooled-negative sizes (Nr = B S rows against
// Nc = world x B (S + 1) columns, D = 512 ... 2048):
//   * the logit tile is 256 rows x 256 columns in v_mfma_f32_32x32x16_bf16 accumulators, both operand panels staged through LDS by
//     LDS-DMA (the encoder GEMMs' main loop: persistent workgroups, next tile's prologue in flight under the epilogue);
//   * tiles are walked COLUMN-PANEL-major inside an XCD's contiguous run: an XCD touches its share of E once and the (small) P
//     panels from its L2, instead of re-reading the E panel once per 128-row block;
//   * masking needs no LDS tables and no barriers: one byte per (user, column) -- member of the user's S + 1 ids / padding column /
//     column past the pool -- and the log-popularity of every column are laid out by a prep launch IN LANE ORDER of the accumulator
//     tile, so a lane fetches its 32 cells of a row with two 16-byte loads;
//   * backward: the same tile, then dlogit^T is written (bf16 [Nc][ldr]); dE = dl^T P is ONE NT GEMM on the eight-phase kernel
//     (fp32 straight from the accumulators: no split over rows, no slab fold), dP = dl E the transposing TN GEMM over column chunks.
#include "gemm8p_core.hpp"
#include "ce_args.hpp"

namespace {
using namespace g8;
constexpr float MASKED_LOGIT = -1e4f;
constexpr int SLICE = 4096;
constexpr int LDS_TOTAL = LDS_BYTES + 8 * SLICE;
// The E panels stream through an XCD once (column-panel-major order); P (2.6 MB at B = 128, D = 512) is re-read from L2 by every tile.
// Non-temporal loads for E keep them from pushing P out of the 4-MiB L2: forward FETCH_SIZE 43.5 -> 26.9 MiB raw per launch at the pooled
// size (2.2x the operand bytes instead of 3.6x).  The backward streams 110 MB of dl^T through the same L2 and fetched slightly MORE with
// them (56.8 -> 63.0 MiB): default policy there.
template <bool BWD>
constexpr int E_AUX = BWD ? 0 : 2;

// The tile works in the BASE-2 domain: x2 = (acc - log pop) * log2(e) is one fma per cell (the table holds log2(e) * log pop) and
// 2^x is the native v_exp_f32; partial maxima / sums leave the kernel in that domain and ce_combine converts (lse = ln 2 * (max2 +
// log2 sum)).  A masked cell is the VALUE -1e4 inside the softmax (as in the reference): exp2((-1e4 - lse) log2 e) == 0 in fp32.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASKED2 = MASKED_LOGIT * LOG2E;

// cell flags of the (user, column) table: non-zero = the cell is overwritten with -1e4 (model.py:50-63)
constexpr uint32_t F_MEMBER = 1, F_INVALID = 2;

// Position of column c (within its 64-column wave chunk) in LANE ORDER: the lane with h = lane >> 5 owns columns
// 32 Ni + 8 g + 4 h + r (Ni < 2, g < 4, r < 4) and reads them as 32 consecutive cells [h][Ni][g][r].
__host__ __device__ inline int lane_order(int c64) {
    return ((c64 >> 2) & 1) * 32 + (c64 >> 5) * 16 + ((c64 >> 3) & 3) * 4 + (c64 & 3);
}

// ---- prep: tab[u][Ncp] flags and lpp[Ncp] = log2(e) * log-popularity, both in lane order per 64-column chunk (Ncp = tiles_n * 256)
template <typename T16>
__device__ __forceinline__ void ce8p_pos_rows(const bf16* __restrict__ P, const bf16* __restrict__ E, const float* __restrict__ col_logpop,
                                              const uint8_t* __restrict__ col_valid, float* __restrict__ pos, int u, int j, int S, int D, int col_offset);

// blockIdx.y = user.  Blocks x < prep_blocks build the flag table / log-pop row; the blocks behind them compute the user's positive
// logits (four rows each): one launch for the two independent pieces of bookkeeping.
template <typename T16>
__global__ __launch_bounds__(256) void ce8p_prep_kernel(const int32_t* __restrict__ row_ids, const int32_t* __restrict__ col_ids,
                                                        const float* __restrict__ col_logpop, const uint8_t* __restrict__ col_valid,
                                                        uint8_t* __restrict__ tab, float* __restrict__ lpp, int B, int S1, int Nc, int Ncp,
                                                        int prep_blocks, const bf16* __restrict__ P, const bf16* __restrict__ E,
                                                        float* __restrict__ pos, int D, int col_offset) {
    extern __shared__ int32_t s_uid[];                     // this user's S + 1 slot ids
    const int u = blockIdx.y;
    if ((int)blockIdx.x >= prep_blocks) {
        const int j = ((int)blockIdx.x - prep_blocks) * 4 + (int)(threadIdx.x >> 6);
        if (j < S1 - 1) ce8p_pos_rows<T16>(P, E, col_logpop, col_valid, pos, u, j, S1 - 1, D, col_offset);
        return;
    }
    for (int i = threadIdx.x; i < S1; i += blockDim.x) s_uid[i] = row_ids[u * S1 + i];
    __syncthreads();
    // one thread per group of 4 consecutive columns (= 4 consecutive cells in lane order; Nc % 4 == 0: a group is all inside the pool
    // or all past it): 16-byte loads of the ids / log-pop, one 4-byte table store
    for (int c = 4 * (blockIdx.x * blockDim.x + threadIdx.x); c < Ncp; c += 4 * prep_blocks * blockDim.x) {
        uint32_t f4 = 0;
        float4 lp = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);   // past the pool: x2 = fma(acc, log2 e, -inf) = -inf (weight 0, never the maximum)
        if (c < Nc) {
            const int4 id = *reinterpret_cast<const int4*>(col_ids + c);
            const uint32_t cv = *reinterpret_cast<const uint32_t*>(col_valid + c);
            bool h0 = false, h1 = false, h2 = false, h3 = false;
            for (int k = 0; k < S1; ++k) {
                const int32_t v = s_uid[k];
                h0 |= v == id.x; h1 |= v == id.y; h2 |= v == id.z; h3 |= v == id.w;
            }
            f4 = (h0 ? F_MEMBER : 0u) | (h1 ? F_MEMBER << 8 : 0u) | (h2 ? F_MEMBER << 16 : 0u) | (h3 ? F_MEMBER << 24 : 0u);
            f4 |= ((cv & 0xffu) ? 0u : F_INVALID) | ((cv & 0xff00u) ? 0u : F_INVALID << 8) | ((cv & 0xff0000u) ? 0u : F_INVALID << 16) |
                  ((cv & 0xff000000u) ? 0u : F_INVALID << 24);
            if (u == 0) {
                const float4 l = *reinterpret_cast<const float4*>(col_logpop + c);
                lp = make_float4(l.x * LOG2E, l.y * LOG2E, l.z * LOG2E, l.w * LOG2E);
            }
        }
        const int pos = (c & ~63) + lane_order(c & 63);
        *reinterpret_cast<uint32_t*>(tab + (size_t)u * Ncp + pos) = f4;
        if (u == 0) *reinterpret_cast<float4*>(lpp + pos) = lp;
    }
}

// The positive logit of every row, pos[m] = P[m] . E[label(m)] - log pop (model.py:45-50; -1e4 when the label's column is a padding
// slot, :51-52 -- only on rows that are dropped anyway): one wavefront per row.  Separate from the tile kernel so that its inner loop
// carries no "is this my label" select per cell; the forward needs it for loss = lse - pos, the backward for the one cell per row
// whose gradient is softmax - 1.  Runs in the tail blocks of ce8p_prep_kernel.
template <typename T16>
__device__ __forceinline__ void ce8p_pos_rows(const bf16* __restrict__ P, const bf16* __restrict__ E, const float* __restrict__ col_logpop,
                                              const uint8_t* __restrict__ col_valid, float* __restrict__ pos, int u, int j, int S, int D, int col_offset) {
    const int lane = threadIdx.x & 63;
    const int row = u * S + j;
    const int lab = col_offset + u * (S + 1) + j + 1;
    const bf16* p = P + (size_t)row * D;
    const bf16* e = E + (size_t)lab * D;
    float acc = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
        const uint4 a = *reinterpret_cast<const uint4*>(p + c), b = *reinterpret_cast<const uint4*>(e + c);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc = fmaf(h16<T16>::bits2f(aw[k] & 0xffffu), h16<T16>::bits2f(bw[k] & 0xffffu), acc);
            acc = fmaf(h16<T16>::bits2f(aw[k] >> 16), h16<T16>::bits2f(bw[k] >> 16), acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) pos[row] = col_valid[lab] ? acc - col_logpop[lab] : MASKED_LOGIT;
}

struct Tile8 { int m0, n0; };

// Base-2 masked logits of the lane's row of block Mi: x[k], k = Ni * 16 + g * 4 + r  <->  column nw + 32 Ni + 8 g + 4 h + r.
// c0 | c1: the 32 flag bytes of (user of the row, this lane's columns); lp2: log2(e) * log-popularity of those columns (+inf past the
// pool); labk: local index of the row's positive (-1: not among this lane's columns) -- its MEMBER flag is cleared (model.py:61-62:
// the positive is taken out of the reject mask; a padding-slot positive stays masked, :51-52).  MASKED: value of an overwritten cell.
template <bool BWD>
__device__ __forceinline__ void logits2_row(float (&x)[32], const f32x16_t (&a)[2], const uint4 c0, const uint4 c1, const float4 (&lp2)[8], int labk) {
    uint32_t cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const int lq = labk >> 2;
    const uint32_t keep = ~(F_MEMBER << (8 * (labk & 3)));
#pragma unroll
    for (int q = 0; q < 8; ++q) cw[q] = (q == lq) ? (cw[q] & keep) : cw[q];
    // backward: an overwritten cell has weight 0 AND no gradient -> -inf does both without a second select
    constexpr float MASKED = BWD ? -INFINITY : MASKED2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float lq4[4] = {lp2[q].x, lp2[q].y, lp2[q].z, lp2[q].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float fb = (float)((cw[q] >> (8 * r)) & 0xffu);                  // v_cvt_f32_ubyte<r>
            const float t = fmaf(a[q >> 2][(q & 3) * 4 + r], LOG2E, -lq4[r]);
            x[q * 4 + r] = fb > 0.f ? MASKED : t;
        }
    }
}

template <typename T16, bool BWD>
__device__ __forceinline__ void ce_tile(const Ce8Args& p, char* smem, const bf16* __restrict__ Pc, const bf16* __restrict__ Ec,
                                        const bf16* __restrict__ Pn, const bf16* __restrict__ En, const Tile8 cur, const Tile8 nxt,
                                        const bool first, const int nk, const int krem) {
    f32x16_t acc[4][2];
    {
        int tid_m = threadIdx.x;
        asm volatile("" : "+v"(tid_m));
        Ctx c;
        make_ctx(c, tid_m, Pc, Ec, p.Nr - cur.m0, p.Nc - cur.n0, p.D, p.D, krem);
        if (first) issue_prologue<E_AUX<BWD>>(c, smem, nk);
        mainloop8p<T16, E_AUX<BWD>>(c, __builtin_amdgcn_readfirstlane(tid_m >> 8), nk, 0, smem, acc, nullptr);
    }
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane = tid_e & 63, r5 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid_e >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nw = cur.n0 + wc * 64;                          // first column of this wave's 64-column chunk
    const int S1 = p.S + 1;
    // ---- everything the epilogue reads from global memory, requested BEFORE the next tile's prologue DMA (vector memory operations
    // retire in order: waiting for these never waits for the DMA behind them)
    float4 lp[8];
    {
        const float4* lpv = reinterpret_cast<const float4*>(p.lpp + nw + h * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) lp[q] = lpv[q];
    }
    uint4 cells[4][2];
    int mrow[4], labk[4];
    float rw[4], rlse2[4], rpos2[4];
#pragma unroll
    for (int Mi = 0; Mi < 4; ++Mi) {
        const int m = cur.m0 + wr * 128 + Mi * 32 + r5;
        const int mc = min(m, p.Nr - 1);                       // rows past Nr: any valid row (nothing of them is stored)
        const int u = mc / p.S, j = mc - u * p.S;
        mrow[Mi] = m;
        const uint4* cv = reinterpret_cast<const uint4*>(p.tab + (size_t)u * p.Ncp + nw + h * 32);
        cells[Mi][0] = cv[0];
        cells[Mi][1] = cv[1];
        const int rel = p.col_offset + u * S1 + j + 1 - nw;    // label column (model.py:45-48) relative to the chunk
        labk[Mi] = (rel >= 0 && rel < 64 && ((rel >> 2) & 1) == h) ? ((rel >> 5) * 16 + ((rel >> 3) & 3) * 4 + (rel & 3)) : -1;
        if constexpr (BWD) {
            rlse2[Mi] = p.row_lse[mc] * LOG2E;
            rpos2[Mi] = p.pos[mc];
            rw[Mi] = (m < p.Nr && p.row_valid[mc]) ? 1.f : 0.f;
        }
    }
    pin();
    {   // next tile's prologue flies under this epilogue (unconditional: see gemm8p.hip)
        int tid_n = threadIdx.x;
        asm volatile("" : "+v"(tid_n));
        Ctx cn;
        make_ctx(cn, tid_n, Pn, En, p.Nr - nxt.m0, p.Nc - nxt.n0, p.D, p.D, krem);
        issue_prologue<E_AUX<BWD>>(cn, smem, nk);
    }
    if constexpr (!BWD) {
        // per (row, 64-column chunk of this wave) maximum and sum; the four chunks of a row (the four wave columns) are merged through
        // LDS so that ONE partial per (row, 256-column tile) leaves the workgroup: K2 = tiles_n entries per row for ce_combine
        float2* red = reinterpret_cast<float2*>(smem + LDS_BYTES);        // [4 wave columns][256 rows] (the epilogue slices' space)
#pragma unroll
        for (int Mi = 0; Mi < 4; ++Mi) {
            float x[32];
            logits2_row<false>(x, acc[Mi], cells[Mi][0], cells[Mi][1], lp, labk[Mi]);
            float mx = x[0];
#pragma unroll
            for (int k = 1; k < 32; ++k) mx = fmaxf(mx, x[k]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float base = mx > -INFINITY ? mx : 0.f;      // a chunk wholly past the pool: every term is 2^-inf = 0
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) sm += __builtin_amdgcn_exp2f(x[k] - base);
            sm += __shfl_xor(sm, 32, 64);
            if (h == 0) red[wc * 256 + wr * 128 + Mi * 32 + r5] = make_float2(mx, sm);
            pin();
        }
        bar();      // all eight waves (they are aligned here: mainloop8p ends with the wave rows re-joined)
        if (tid_e < 256) {
            const int m = cur.m0 + tid_e;
            const float2 p0 = red[tid_e], p1 = red[256 + tid_e], p2 = red[512 + tid_e], p3 = red[768 + tid_e];
            const float mx = fmaxf(fmaxf(p0.x, p1.x), fmaxf(p2.x, p3.x));
            const float base = mx > -INFINITY ? mx : 0.f;
            const float sm = p0.y * __builtin_amdgcn_exp2f(p0.x - base) + p1.y * __builtin_amdgcn_exp2f(p1.x - base) +
                             p2.y * __builtin_amdgcn_exp2f(p2.x - base) + p3.y * __builtin_amdgcn_exp2f(p3.x - base);
            if (m < p.Nr) {
                const int kcol = cur.n0 >> 8;
                p.pmax[(size_t)m * p.K2 + kcol] = mx;
                p.psum[(size_t)m * p.K2 + kcol] = sm;
            }
        }
        // (the next write to `red` is a whole main loop -- dozens of barriers -- away: no second barrier)
    } else {
        // dlogit = g (softmax - onehot) on the cells of valid rows that were not overwritten, 0 elsewhere (model.py:65-67 backward;
        // an overwritten cell receives no gradient: index_put semantics), written TRANSPOSED: dlt[c][m].  Each 32-row block goes
        // through the wave's 4-KiB slice as [64 columns][32 rows] bf16 so that the global stores are 16-byte lanes along m (64-byte
        // row segments).  The one cell per row that carries the "- 1" is patched in the slice by the lane that owns it, from pos[m].
        const float g = p.gscale * (p.gscale_dev ? *p.gscale_dev : 1.0f);
        char* ws = smem + LDS_BYTES + wave * SLICE;
        unsigned short* ws16 = reinterpret_cast<unsigned short*>(ws);
        bf16* dlt = p.dlt;
        auto wfence = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
#pragma unroll
        for (int Mi = 0; Mi < 4; ++Mi) {
            float x[32];
            logits2_row<true>(x, acc[Mi], cells[Mi][0], cells[Mi][1], lp, labk[Mi]);
            const float w = rw[Mi] * g, lse2 = rlse2[Mi];
#pragma unroll
            for (int k = 0; k < 32; k += 2) {      // pairs k, k + 1 = consecutive columns, same row: two 2-byte cells of different LDS rows
                const float d0 = w * __builtin_amdgcn_exp2f(x[k] - lse2), d1 = w * __builtin_amdgcn_exp2f(x[k + 1] - lse2);
                const uint32_t pk = h16<T16>::pack2(d0, d1);
                const int cl = (k >> 4) * 32 + ((k >> 2) & 3) * 8 + 4 * h + (k & 3);     // column within the chunk
                ws16[cl * 32 + r5] = (unsigned short)(pk & 0xffffu);
                ws16[(cl + 1) * 32 + r5] = (unsigned short)(pk >> 16);
            }
            if (labk[Mi] >= 0) {                   // the positive: g (softmax - 1), or nothing at all when it sits on a padding slot (rows dropped anyway)
                const int k = labk[Mi];
                const int cl = (k >> 4) * 32 + ((k >> 2) & 3) * 8 + 4 * h + (k & 3);
                const float pos = rpos2[Mi];
                const float d = pos == MASKED_LOGIT ? 0.f : w * (__builtin_amdgcn_exp2f(pos * LOG2E - lse2) - 1.f);
                ws16[cl * 32 + r5] = h16<T16>::bits(d);
            }
            wfence();
            // read back [64 columns][64 B]: lane -> column (lane >> 2) + 16 i, 16-byte slot lane & 3 (8 rows m)
            const int mb = cur.m0 + wr * 128 + Mi * 32 + (lane & 3) * 8;             // first of this lane's 8 rows
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cl = (lane >> 2) + 16 * i;
                const uint4 q = *reinterpret_cast<const uint4*>(ws + cl * 64 + (lane & 3) * 16);
                const int c = nw + cl;
                if (c < p.Nc && mb < p.ldr) *reinterpret_cast<uint4*>(dlt + (size_t)c * p.ldr + mb) = q;
            }
            wfence();
        }
    }
}

template <typename T16, bool BWD>
__global__ __launch_bounds__(THREADS) void ce8p_kernel(Ce8Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int nk = (p.D + KE - 1) / KE, krem = p.D - (nk - 1) * KE;
    const int G = gridDim.x;
    const int n_units = (nwg - (int)blockIdx.x + G - 1) / G;
    auto unit_at = [&](int i) {
        // XCD x walks a contiguous run of order indices; order = column panel major (tn, then tm): an XCD's run covers a few
        // column panels x ALL row panels -- E fetched once per XCD, the small P re-read from its L2
        const int o = xcd_remap((int)blockIdx.x + i * G, nwg);
        Tile8 t;
        t.n0 = (o / p.tiles_m) * TN;
        t.m0 = (o % p.tiles_m) * TM;
        return t;
    };
    Tile8 cur = unit_at(0);
    for (int i = 0; i < n_units; ++i) {
        const Tile8 nxt = i + 1 < n_units ? unit_at(i + 1) : cur;
        ce_tile<T16, BWD>(p, smem, p.P + (size_t)cur.m0 * p.D, p.E + (size_t)cur.n0 * p.D, p.P + (size_t)nxt.m0 * p.D, p.E + (size_t)nxt.n0 * p.D,
                     cur, nxt, i == 0, nk, krem);
        cur = nxt;
    }
    vm_wait<0>();      // the trailing prologue must not land in LDS that already belongs to another workgroup
}

int n_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        v &= ~7;
        return v < 8 ? 8 : v;
    }();
    return n;
}
}  // namespace

int g_ce8p_mode = 0;

bool ce8p_eligible(const morec_ce_desc* d) {
    const long Nr = (long)d->B * d->S;
    if (g_ce8p_mode == 1) return false;
    if (!is_h16(d->dtype) || d->D % 8 || d->D <= KE || d->Nc % 8 || Nr % 8) return false;
    if ((long)d->Nc * d->D * 2 >= 0x7fffffffL || Nr * d->D * 2 >= 0x7fffffffL) return false;      // 32-bit DMA offsets within a panel run
    // enough 256 x 256 tiles to give most CUs one (below that the 128 x 128 kernels fill the chip better)
    if (g_ce8p_mode == 2) return true;
    return ((Nr + 255) / 256) * (((long)d->Nc + 255) / 256) >= 96;
}

void ce8p_layout(const morec_ce_desc* d, Ce8Layout& L) {
    const size_t Nr = (size_t)d->B * d->S, Nc = d->Nc, D = d->D;
    L.tiles_m = (int)((Nr + 255) / 256);
    L.tiles_n = (int)((Nc + 255) / 256);
    L.Ncp = L.tiles_n * 256;
    L.K2 = L.tiles_n;
    L.ldr = (int)((Nr + 63) & ~(size_t)63);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    L.off_tab = o; o += up((size_t)d->B * L.Ncp);
    L.off_lpp = o; o += up((size_t)L.Ncp * 4);
    L.off_pos = o; o += up(Nr * 4);
    L.off_pmax = o; o += up(Nr * L.K2 * 4);
    L.off_psum = o; o += up(Nr * L.K2 * 4);
    L.off_part = o; o += up(((Nr + 3) / 4) * 4);
    L.fwd_bytes = o;
    o = L.off_pmax;                                            // backward reuses the space behind the tables
    L.off_dlt = o; o += up(Nc * L.ldr * 2);
    L.off_pt = o; o += up(D * L.ldr * 2);
    L.off_dp32 = o; o += up(Nr * D * 4);
    L.off_de32 = o; o += up(Nc * D * 4);
    L.tn_split = 1;
    {   // dP = dl E: transposing GEMM over column chunks, ~one workgroup per CU, at least 512 columns per chunk
        const long tiles = (long)((Nr + 255) / 256) * ((D + 255) / 256);
        long s = (256 + tiles - 1) / tiles;
        const long max_s = (long)Nc / 512 > 0 ? (long)Nc / 512 : 1;
        L.tn_split = (int)(s > max_s ? max_s : s);
        if (L.tn_split < 1) L.tn_split = 1;
    }
    L.off_slabs = o; o += up((size_t)L.tn_split * Nr * D * 4);
    L.bwd_bytes = o;
}

static int ce8p_prep(const morec_ce_desc* d, const Ce8Layout& L, char* ws, const void* P, const void* E, const int32_t* row_ids, const int32_t* col_ids,
                     const float* col_logpop, const uint8_t* col_valid, hipStream_t s) {
    const int S1 = d->S + 1;
    const int prep_blocks = (L.Ncp / 4 + 255) / 256;
    dim3 grid(prep_blocks + (d->S + 3) / 4, d->B);
    by_h16(d->dtype, [&](auto* t) {
        using T = MOREC_TAG_T(t);
        hipLaunchKernelGGL(ce8p_prep_kernel<T>, grid, dim3(256), S1 * sizeof(int32_t), s, row_ids, col_ids, col_logpop, col_valid,
                       reinterpret_cast<uint8_t*>(ws + L.off_tab), reinterpret_cast<float*>(ws + L.off_lpp), d->B, S1, d->Nc, L.Ncp, prep_blocks,
                       reinterpret_cast<const bf16*>(P), reinterpret_cast<const bf16*>(E), reinterpret_cast<float*>(ws + L.off_pos), d->D,
                       d->col_offset);
    });
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

static void ce8p_fill(const morec_ce_desc* d, const Ce8Layout& L, char* ws, const void* P, const void* E, const uint8_t* row_valid, Ce8Args& a) {
    a.P = reinterpret_cast<const bf16*>(P); a.E = reinterpret_cast<const bf16*>(E);
    a.tab = reinterpret_cast<const uint8_t*>(ws + L.off_tab); a.lpp = reinterpret_cast<const float*>(ws + L.off_lpp);
    a.row_valid = row_valid;
    a.pos = reinterpret_cast<float*>(ws + L.off_pos);
    a.B = d->B; a.S = d->S; a.D = d->D; a.Nr = d->B * d->S; a.Nc = d->Nc; a.col_offset = d->col_offset;
    a.K2 = L.K2; a.Ncp = L.Ncp; a.ldr = L.ldr; a.tiles_m = L.tiles_m; a.tiles_n = L.tiles_n;
}

template <typename T16, bool BWD>
static int ce8p_launch(const Ce8Args& a, hipStream_t s) {
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&ce8p_kernel<T16, BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    (void)attr_rc;
    const int nwg = a.tiles_m * a.tiles_n, ncu = n_cus();
    hipLaunchKernelGGL((ce8p_kernel<T16, BWD>), dim3(nwg < ncu ? nwg : ncu), dim3(THREADS), LDS_TOTAL, s, a);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// forward: per-(row, 64-column) softmax partials (BASE-2 domain) + the positive logit (natural) into the workspace; the caller runs
// ce_combine over the K2 partials with log2_domain = 1
int ce8p_fwd(const morec_ce_desc* d, const void* P, const void* E, const int32_t* row_ids, const int32_t* col_ids, const float* col_logpop,
             const uint8_t* col_valid, const uint8_t* row_valid, void* workspace, float** pmax, float** psum, float** pos, float** part, int* K2, hipStream_t s) {
    Ce8Layout L;
    ce8p_layout(d, L);
    char* ws = reinterpret_cast<char*>(workspace);
    int rc = ce8p_prep(d, L, ws, P, E, row_ids, col_ids, col_logpop, col_valid, s);
    if (rc) return rc;
    Ce8Args a{};
    ce8p_fill(d, L, ws, P, E, row_valid, a);
    a.pmax = reinterpret_cast<float*>(ws + L.off_pmax); a.psum = reinterpret_cast<float*>(ws + L.off_psum);
    *pmax = a.pmax; *psum = a.psum; *pos = a.pos; *K2 = L.K2;
    *part = reinterpret_cast<float*>(ws + L.off_part);
    return d->dtype == MOREC_F16 ? ce8p_launch<f16, false>(a, s) : ce8p_launch<bf16, false>(a, s);
}

int ce8p_bwd(const morec_ce_desc* d, const void* P, const void* E, const int32_t* row_ids, const int32_t* col_ids, const float* col_logpop,
             const uint8_t* col_valid, const uint8_t* row_valid, const float* row_lse, const float* gscale_dev, float gscale, void* dP, void* dE,
             void* workspace, hipStream_t s) {
    Ce8Layout L;
    ce8p_layout(d, L);
    char* ws = reinterpret_cast<char*>(workspace);
    const int Nr = d->B * d->S, Nc = d->Nc, D = d->D;
    int rc = d->ws_from_fwd ? MOREC_OK : ce8p_prep(d, L, ws, P, E, row_ids, col_ids, col_logpop, col_valid, s);   // tables + positive logits: the forward's, or rebuilt
    if (rc) return rc;
    Ce8Args a{};
    ce8p_fill(d, L, ws, P, E, row_valid, a);
    a.row_lse = row_lse; a.gscale_dev = gscale_dev; a.gscale = gscale;
    a.dlt = reinterpret_cast<bf16*>(ws + L.off_dlt);
    // (every column m < ldr of every row c < Nc is written by some tile -- tiles_m * 256 >= ldr -- with zeros for m >= Nr)
    rc = d->dtype == MOREC_F16 ? ce8p_launch<f16, true>(a, s) : ce8p_launch<bf16, true>(a, s);
    if (rc) return rc;
    void* stream = reinterpret_cast<void*>(s);
    // dE[Nc, D] = dlt[Nc, Nr] . Pt[D, Nr]^T: one NT GEMM, fp32 (handed out as it is when the caller reduces it over ranks)
    bf16* Pt = reinterpret_cast<bf16*>(ws + L.off_pt);
    // (the pad columns of Pt / dlt are never read: K = Nr with pitch ldr, a partial last K-tile is zero-filled by the GEMM)
    rc = morec_transpose(P, Pt, Nr, D, D, L.ldr, d->dtype, d->dtype, stream);
    if (rc) return rc;
    morec_gemm_desc g{};
    g.in_dtype = d->dtype; g.alpha = 1.0f; g.split_k = 1;
    g.M = Nc; g.N = D; g.K = Nr; g.lda = L.ldr; g.ldb = L.ldr; g.ldc = D;
    g.out_dtype = d->dE_fp32 ? MOREC_F32 : d->dtype;
    rc = morec_gemm_nt(&g, a.dlt, Pt, dE, nullptr, nullptr, nullptr, stream);
    if (rc) return rc;
    // dP[Nr, D] = sum_c dlt[c, r] E[c, d]: the transposing GEMM, contraction over the Nc columns cut into tn_split chunks
    if (L.tn_split > 1)      // the slab fold rounds the sum straight into dP: no zero-fill, no fp32 copy, no conversion pass
        return gemm_tn_launch(a.dlt, E, dP, d->dtype, Nc, Nr, D, L.ldr, D, D, d->dtype, L.tn_split, 0, reinterpret_cast<float*>(ws + L.off_slabs),
                              stream);
    float* dP32 = reinterpret_cast<float*>(ws + L.off_dp32);
    rc = morec_gemm_tn(a.dlt, E, dP32, Nc, Nr, D, L.ldr, D, D, d->dtype, 1, 0, nullptr, stream);
    if (rc) return rc;
    return morec_cast(dP32, dP, (size_t)Nr * D, MOREC_F32, d->dtype, stream);
}
