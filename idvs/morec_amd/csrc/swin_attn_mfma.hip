// swin_attn_mfma.hip -- 16-bit (bf16 / f16) Swin window attention (7 x 7 = 49 tokens, 32-wide heads) on the matrix cores.
//
// Same contract as the exact-fp32 / generic kernels in swin.hip (SwinAttention, modeling_swin.py:401-468; window
// partition / cyclic shift / reverse folded into row addressing).  One wavefront per (window, head), four wavefronts
// per workgroup, 49 tokens padded to a 64 x 64 score tile = 4 x 4 MFMA 16x16 tiles:
//   * S^T = K Q^T and dP^T = V dO^T: v_mfma_f32_16x16x32_bf16 whose operands are 16-byte rows of q / k / v / dctx read
//     straight from global memory (head width 32 = one MFMA k-step, so no LDS staging for these);
//     a lane ends up with S[query i = 16 ti + (lane & 15)][key j = 16 tj + 4 (lane >> 4) + r]: a softmax row lives in
//     4 lanes x 16 registers, row max / sum / delta are two shuffles (xor 16, 32);
//   * products that contract over keys (P V, dS K) take P / dS from those registers as the B operand (the k-slot <-> key
//     permutation {32 s + 4 g + e, 32 s + 16 + 4 g + e} is the one ds_read_b64_tr_b16 produces, so nothing is shuffled)
//     and V^T / K^T as hardware-transposed reads of a row-major LDS tile;
//   * products that contract over queries (P^T dO, dS^T Q) read BOTH operands with transposing LDS reads: P / dS go
//     through a row-major bf16 LDS tile (pitch 136 B: conflict-free writes and transposed reads).
// Outputs are produced transposed (O^T, dQ^T, dK^T, dV^T) so that a lane owns 4 consecutive head columns of one token
// row: 8-byte global stores.
#include <algorithm>
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

namespace {
constexpr int DH = 32;
constexpr int WS = 7, NT = WS * WS;
constexpr int TROW = 64;              // bytes per row of a [64 x 32] bf16 operand tile
constexpr int TILE = 64 * TROW;       // 4 KiB
constexpr int PROW = 136;             // bytes per row of the [64 x 64] bf16 P / dS tile (17 x 8: odd multiple of 8 B)
constexpr int PTILE = 64 * PROW;
constexpr int BP = 68;                // floats per row of the shared [NT x 64] bias / dbias tiles
#ifndef SWIN_BWD_NTI
#define SWIN_BWD_NTI 2                // query blocks the one-wave-per-SIMD backward runs through the softmax together (1 | 2 | 4)
#endif
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

struct SwinMArgs {
    const bf16* qkv;
    const float* bias_t;   // [heads][j][i]
    bf16* ctx;
    const bf16* dctx;
    bf16* dqkv;
    float* dbias_t;        // [heads][j][i]
    int n_img, H, W, shift, heads;
    float scale;
    int n_win_total, wpw;  // windows per wavefront
    int gx;                // window groups (4 wavefronts x wpw windows each)
    unsigned qkv_bytes;    // size of qkv / dqkv in bytes (bwd: buffer-descriptor extent, < 4 GiB)
    float* csum;           // bwd, optional: [gx * 4 wavefronts][3 C] fp32 column sums of the wavefront's dq / dk / dv rows
    float* dbias_part;     // bwd, deterministic mode: [heads][gx][NT * NT] per-workgroup dbias tiles (the launcher folds them in order) instead of atomics
};

// T16 = bf16 | f16: the storage type of q / k / v / ctx and their gradients (the operand bits go to the MFMA as they are; only the
// instruction and the fp32 -> 16-bit packs differ)
template <typename T16>
__device__ __forceinline__ f32x4_t mfma(const uint4& a_rows, const uint4& b_rows, f32x4_t acc) {
    // acc[r] += sum_k A[4 (lane >> 4) + r][k] * B[lane & 15][k]; both operands given as "row (lane & 15), 8 k of chunk lane >> 4"
    return h16<T16>::mma16(__builtin_bit_cast(bf16x8_t, a_rows), __builtin_bit_cast(bf16x8_t, b_rows), acc);
}

// transposed fragment of a row-major LDS tile: lane (c, g) receives column col0 + c of rows {32 s + 4 g + e} (e = 0..3) and
// {32 s + 16 + 4 g + e} (elements 4..7)
__device__ __forceinline__ uint4 frag_tr(const char* tile, int row_bytes, int col0, int s) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const char* p0 = tile + (32 * s + 4 * g + (c >> 2)) * row_bytes + (col0 + 4 * (c & 3)) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 16 * row_bytes));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(uint4, v);
}

// ---- [64 x 32] 16-bit operand tiles (K, V, dO, Q): 64-byte rows, the four 16-byte chunks of row r stored at chunk ^ ((r >> 1) & 3).  With plain
// rows the eight lanes of a ds_write_b128 service group (rows c = 0..7, one chunk) sit on TWO 4-bank groups (4-way conflict: 32 LDS cycles per
// store instead of 13) and the 32 lanes of a transposing read (rows R and R + 4 share their banks) are 2-way; with the swizzle both are conflict-free
// (round 6; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the backward was 47 %: profiles/r05_swin_attn_pmc.txt).
// byte offset of (row c + 16 k, chunk g4) for the lane that owns it: tile_wr_off() + k * 1024
__device__ __forceinline__ int tile_wr_off() {
    const int lane = threadIdx.x & 63, c = lane & 15, g4 = lane >> 4;
    return c * TROW + 16 * (g4 ^ ((c >> 1) & 3));
}
// The transposed fragment with the 16 head columns of block pair td spread as 4 groups of 4: lane c receives column 4 td + 8 (c & 3) + e of rows
// {32 s + 4 g + e} (e = 0..3) and {32 s + 16 + 4 g + e}.  As the A operand of an MFMA this makes output row 4 g + r of block td the head column
// 8 g + 4 td + r, i.e. the two blocks td = 0, 1 of a lane are 8 CONSECUTIVE head columns of its token row: one 16-byte store per (token, tensor)
// instead of two 8-byte ones (the 8-byte pieces made these kernels store-issue bound: 2.4 - 3.3 TB/s, profiles/r03_swin_tiny_kernel_stats.csv).
__device__ __forceinline__ uint4 frag_tr_spread(const char* tile, int row_bytes, int td, int s) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    // the lane reads 8 bytes of row R = 32 s + 4 g + (c >> 2), logical chunk c & 3; (R >> 1) & 3 = (2 g + (c >> 3)) & 3 for R and for R + 16
    const char* p0 = tile + (32 * s + 4 * g + (c >> 2)) * row_bytes + 16 * ((c & 3) ^ ((2 * g + (c >> 3)) & 3)) + 8 * td;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 16 * row_bytes));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(uint4, v);
}

struct LaneGeom {
    int row[4];        // natural row of token t = (lane & 15) + 16 k (clamped for padded tokens)
    bool valid[4];
    bool edge_r, edge_c;   // window on the last window row / column of a shifted layer: region mask applies
};

// per-lane constants of the shift-region mask (modeling_swin.py:584-607).  Only windows on the last window row / column
// contain more than one region; inside them the region of a token is decided by (wy >= WS - shift) / (wx >= WS - shift).
struct MaskBits {
    uint32_t ai, bi;   // bit ti: token i = c + 16 ti
    uint32_t aj, bj;   // bit 4 tj + r: token j = 16 tj + 4 g + r
};

__device__ __forceinline__ MaskBits make_mask_bits(int shift) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    MaskBits m{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = c + 16 * t, wy = i / WS, wx = i - wy * WS;
        if (i < NT && wy >= WS - shift) m.ai |= 1u << t;
        if (i < NT && wx >= WS - shift) m.bi |= 1u << t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r, jy = j / WS, jx = j - jy * WS;
            if (j < NT && jy >= WS - shift) m.aj |= 1u << (4 * t + r);
            if (j < NT && jx >= WS - shift) m.bj |= 1u << (4 * t + r);
        }
    }
    return m;
}

// Workgroup -> (head, window group).  A head's q / k / v / ctx slices are 64 bytes of a token row -- half a 128-byte line -- so the
// heads of one window group must run on the SAME XCD (one L2) at about the same time, or every line is fetched from / merged
// in HBM once per head.  Workgroups are dealt to the 8 XCDs round-robin: ids congruent mod 8 share an XCD, and within one
// XCD consecutive ids walk the heads of one window group before moving to the next group.
struct WgMap { int head, bx; };
__device__ __forceinline__ WgMap wg_map(const SwinMArgs& a) {
    const int lin = blockIdx.x, xcd = lin & 7, t = lin >> 3;
    WgMap m;
    m.head = t % a.heads;
    m.bx = (t / a.heads) * 8 + xcd;
    if (m.bx >= a.gx) m.bx = -1;
    return m;
}

__device__ __forceinline__ LaneGeom window_geom(const SwinMArgs& a, int g) {
    const int c = threadIdx.x & 15;
    const int nWx = a.W / WS, nWy = a.H / WS, nW = nWx * nWy;
    const int img = g / nW, wi = g - img * nW;
    const int wr = wi / nWx, wc = wi - wr * nWx;
    LaneGeom G;
    G.edge_r = a.shift > 0 && wr == nWy - 1;
    G.edge_c = a.shift > 0 && wc == nWx - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = c + 16 * k;
        G.valid[k] = t < NT;
        const int tt = G.valid[k] ? t : 0;
        const int wy = tt / WS, wx = tt - wy * WS;
        int y = wr * WS + wy + a.shift, x = wc * WS + wx + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        G.row[k] = (img * a.H + y) * a.W + x;
    }
    return G;
}

// Accumulator element r of key block tj holds key 16 tj + 4 (lane >> 4) + r: with NT = 49, elements r = 1..3 of block 3 are padded keys in EVERY
// lane (probability exactly 0, whatever the scores) -- the softmax, its backward and the bias gradient skip them at compile time.
__device__ __forceinline__ constexpr bool dead_key(int tj, int r) { return 16 * tj + r >= NT; }

// Scores (S^T accumulators s[tj][t][r], t = query blocks TB .. TB + NTI - 1) -> probabilities, in place.  bias4(tj, ti) supplies the 4
// consecutive keys' bias (+ -inf on padded keys).  Written in STAGES over all NTI blocks -- scale + bias; the region mask behind ONE wave-uniform
// branch; max; exp + sum; normalise -- so that the NTI dependency chains (MFMA result -> max -> lane exchange -> exp -> sum -> lane exchange ->
// scale) sit in the same basic blocks and the scheduler overlaps them: with the branch inside a per-block loop every query block was its own
// scheduling region and, at one wave per SIMD, paid each of those latencies in full.
template <int TB, int NTI, typename BiasF>
__device__ __forceinline__ void softmax_part(f32x4_t (&s)[4][NTI], float scale, BiasF bias4, const LaneGeom& G, const MaskBits& mb) {
    const bool masked = G.edge_r || G.edge_c;
#pragma unroll
    for (int t = 0; t < NTI; ++t)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const f32x4_t b = bias4(tj, TB + t);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!dead_key(tj, r)) s[tj][t][r] = fmaf(s[tj][t][r], scale, b[r]);
        }
    if (masked) {   // wave-uniform: only windows on the last window row / column of a shifted block hold several regions
        const uint32_t er = G.edge_r ? 0xffffu : 0u, ec = G.edge_c ? 0xffffu : 0u;
#pragma unroll
        for (int t = 0; t < NTI; ++t) {
            const int ti = TB + t;
            // bit (4 tj + r) set <=> key j lies in another region than query i
            const uint32_t bits = ((((mb.ai >> ti) & 1u) ? ~mb.aj : mb.aj) & er) | ((((mb.bi >> ti) & 1u) ? ~mb.bj : mb.bj) & ec);
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (!dead_key(tj, r)) s[tj][t][r] = fmaf((float)((bits >> (4 * tj + r)) & 1u), -100.0f, s[tj][t][r]);
        }
    }
    float m[NTI], sum[NTI];
#pragma unroll
    for (int t = 0; t < NTI; ++t) {
        m[t] = -INFINITY;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!dead_key(tj, r)) m[t] = fmaxf(m[t], s[tj][t][r]);
    }
#pragma unroll
    for (int t = 0; t < NTI; ++t) m[t] = rows4_max(m[t]);
#pragma unroll
    for (int t = 0; t < NTI; ++t) {
        sum[t] = 0.f;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (dead_key(tj, r)) continue;
                const float e = __expf(s[tj][t][r] - m[t]);
                s[tj][t][r] = e;
                sum[t] += e;
            }
    }
#pragma unroll
    for (int t = 0; t < NTI; ++t) sum[t] = rows4_sum(sum[t]);
#pragma unroll
    for (int t = 0; t < NTI; ++t) {
        // padded query rows (token >= NT) get probability 0 everywhere: their q / dO fragments are unguarded re-reads of token
        // 0's rows (see the load note in the kernels), and a zero P row keeps them out of dS, dK, dV and dbias
        const float inv = (16 * (TB + t) + (int)(threadIdx.x & 15) < NT) ? 1.0f / sum[t] : 0.f;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[tj][t][r] = dead_key(tj, r) ? 0.f : s[tj][t][r] * inv;
    }
}

// B-operand fragment (k-step sk over keys) of row block ti from the register tile
template <typename T16>
__device__ __forceinline__ uint4 pack_frag(const f32x4_t (&p)[4][4], int ti, int sk) {
    uint4 f;
    f.x = h16<T16>::pack2(p[2 * sk][ti][0], p[2 * sk][ti][1]);
    f.y = h16<T16>::pack2(p[2 * sk][ti][2], p[2 * sk][ti][3]);
    f.z = h16<T16>::pack2(p[2 * sk + 1][ti][0], p[2 * sk + 1][ti][1]);
    f.w = h16<T16>::pack2(p[2 * sk + 1][ti][2], p[2 * sk + 1][ti][3]);
    return f;
}

template <typename T16>
__device__ __forceinline__ void store8_bf16(bf16* p, const f32x4_t& lo, const f32x4_t& hi) {
    uint4 o;
    o.x = h16<T16>::pack2(lo[0], lo[1]);
    o.y = h16<T16>::pack2(lo[2], lo[3]);
    o.z = h16<T16>::pack2(hi[0], hi[1]);
    o.w = h16<T16>::pack2(hi[2], hi[3]);
    *reinterpret_cast<uint4*>(p) = o;
    store_b128_guard();
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wavefront execute in order; this only stops the compiler from moving LDS accesses across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T16>
__global__ __launch_bounds__(256) void swin_attn_fwd_mfma_kernel(SwinMArgs a) {
    __shared__ __attribute__((aligned(16))) char sVall[4 * TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g4 = lane >> 4;
    const WgMap wm = wg_map(a);
    if (wm.bx < 0) return;
    const int head = wm.head, C = a.heads * DH, pitch = 3 * C;
    char* sV = sVall + wave * TILE;
    // bias registers in the accumulator layout; padded keys carry -inf (their probabilities become exactly 0)
    float bias[4][4][4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * tj + 4 * g4 + r, i = 16 * ti + c;
                bias[tj][ti][r] = (j >= NT) ? -INFINITY : (i < NT ? a.bias_t[((size_t)head * NT + j) * NT + i] : 0.f);
            }
    const MaskBits mb = make_mask_bits(a.shift);
    const int w0 = (wm.bx * 4 + wave) * a.wpw, w1 = min(a.n_win_total, w0 + a.wpw);
    for (int g = w0; g < w1; ++g) {
        const LaneGeom G = window_geom(a, g);
        // No guarded loads: `valid ? *p : 0` compiles to one basic block per load (each ending in a full s_waitcnt) and a load
        // next to an LDS store stays in program order.  Padded tokens re-read token 0's rows (G.row is clamped); a padded key has
        // bias -inf (probability exactly 0), a padded query row is zeroed in softmax_rows and never stored.  All twelve loads of
        // the window are issued before the first LDS write.  (Requesting the NEXT window's rows a window ahead, as the backward
        // does, needs 48 more registers: with the 64 bias registers that is one wave per SIMD, and the forward -- short dependency
        // chains, 3.5 TB/s with two waves -- measured 4-30 % slower that way.)
        uint4 qf[4], kf[4], vf[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16* base = a.qkv + (size_t)G.row[k] * pitch + head * DH + 8 * g4;
            qf[k] = *reinterpret_cast<const uint4*>(base);
            kf[k] = *reinterpret_cast<const uint4*>(base + C);
            vf[k] = *reinterpret_cast<const uint4*>(base + 2 * C);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(sV + tile_wr_off() + k * 1024) = vf[k];
        f32x4_t s[4][4];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) s[tj][ti] = mfma<T16>(kf[tj], qf[ti], f32x4_t{0.f, 0.f, 0.f, 0.f});
        softmax_part<0, 4>(s, a.scale, [&](int tj, int ti) { return f32x4_t{bias[tj][ti][0], bias[tj][ti][1], bias[tj][ti][2], bias[tj][ti][3]}; }, G, mb);
        wave_lds_fence();
        f32x4_t o[2][4];
#pragma unroll
        for (int td = 0; td < 2; ++td)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) o[td][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
            uint4 pf[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) pf[ti] = pack_frag<T16>(s, ti, sk);
#pragma unroll
            for (int td = 0; td < 2; ++td) {
                const uint4 vt = frag_tr_spread(sV, TROW, td, sk);      // o[td][ti][r] = head column 8 g + 4 td + r of token 16 ti + c
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) o[td][ti] = mfma<T16>(vt, pf[ti], o[td][ti]);
            }
        }
        wave_lds_fence();   // the transposed reads are done before the next window overwrites the tile
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
            if (G.valid[ti]) {
                store8_bf16<T16>(a.ctx + (size_t)G.row[ti] * C + head * DH + 8 * g4, o[0][ti], o[1][ti]);
            }
    }
}

// The backward runs the softmax for NTI of the four query blocks at a time: with all four at once and two waves per SIMD the live set (P 64 +
// dP 64 + dbias 64 + operand fragments) spilled 42 registers, so that form takes one block at a time; the one-wave-per-SIMD form (WIDE, 512
// registers) takes two, whose dependency chains overlap.
template <typename T16, int NTI>
__device__ __forceinline__ uint4 pack_part(const f32x4_t (&p)[4][NTI], int t, int sk) {
    uint4 f;
    f.x = h16<T16>::pack2(p[2 * sk][t][0], p[2 * sk][t][1]);
    f.y = h16<T16>::pack2(p[2 * sk][t][2], p[2 * sk][t][3]);
    f.z = h16<T16>::pack2(p[2 * sk + 1][t][0], p[2 * sk + 1][t][1]);
    f.w = h16<T16>::pack2(p[2 * sk + 1][t][2], p[2 * sk + 1][t][3]);
    return f;
}

// The 16 operand rows (q, k, v, dO: 16 bytes each per lane and 16-token block) of one (window, head).  Unguarded loads: padded
// tokens re-read token 0's rows (see the forward kernel).
// sum over the 16 lanes of a DPP row (every lane ends up with the total): quad butterflies, then the two mirrors
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

struct BwdFrags { uint4 q[4], k[4], v[4], o[4]; };
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ uint4 as_uint4(u32x4_t v) { return make_uint4(v[0], v[1], v[2], v[3]); }
// Buffer descriptors + 32-bit byte offsets for every global access of the backward kernel (the launcher checks the tensors are
// below 4 GiB): with flat 64-bit addresses hipcc kept two dozen precomputed row pointers alive across the softmax -- 50 spilled registers.
struct BwdBufs { __amdgpu_buffer_rsrc_t qkv, dctx, dqkv; };
__device__ __forceinline__ void load_bwd_frags(const BwdBufs& B, const LaneGeom& G, int head, int C, int pitch, int g4, BwdFrags& f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t ro = (uint32_t)G.row[k] * (uint32_t)(pitch * 2) + (uint32_t)((head * DH + 8 * g4) * 2);
        f.q[k] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(B.qkv, ro, 0, 0));
        f.k[k] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(B.qkv, ro, C * 2, 0));
        f.v[k] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(B.qkv, ro, C * 4, 0));
        f.o[k] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(B.dctx, (uint32_t)G.row[k] * (uint32_t)(C * 2) + (uint32_t)((head * DH + 8 * g4) * 2), 0, 0));
    }
}
// the lane's 8 consecutive head columns of one token row (blocks td = 0 | 1 of the spread fragments): one 16-byte store at byte offset
// voff (per lane) + soff (wave-uniform) of a [rows][pitch] bf16 matrix, values scaled by mul
template <typename T16>
__device__ __forceinline__ void store8_buf(const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, int soff, const f32x4_t& lo, const f32x4_t& hi, float mul) {
    u32x4_t o;
    o[0] = h16<T16>::pack2(lo[0] * mul, lo[1] * mul);
    o[1] = h16<T16>::pack2(lo[2] * mul, lo[3] * mul);
    o[2] = h16<T16>::pack2(hi[0] * mul, hi[1] * mul);
    o[3] = h16<T16>::pack2(hi[2] * mul, hi[3] * mul);
    __builtin_amdgcn_raw_buffer_store_b128(o, rs, voff, soff, 0);
    store_b128_guard();
}

// dbias accumulates in registers (LDS float atomics cost 2x the rest of the kernel) and is reduced once per workgroup.
// ONE global round trip per window: all sixteen operand rows are requested together -- for the NEXT window, while this window's
// dK product (the last phase, when the softmax registers are dead) runs -- and q stays in registers for the dK product instead
// of being re-read.  (Three dependent round trips per window -- q/k, then v/dO, then q again -- with two wavefronts per SIMD
// to hide them left the kernel at 2 TB/s whatever the stage: profiles/r02_swin_attn.txt.)
template <typename T16, bool WIDE>
__global__ __launch_bounds__(256, WIDE ? 1 : 2) void swin_attn_bwd_mfma_kernel(SwinMArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sBias = reinterpret_cast<float*>(smem);                 // [NT (query i)][BP] shared by the block's 4 wavefronts
    char* wbase = reinterpret_cast<char*>(sBias + NT * BP);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g4 = lane >> 4;
    char* sK = wbase + wave * (2 * TILE + PTILE);                  // [64 x 32] K tile
    char* sX = sK + TILE;                                          // [64 x 32] dO, then Q
    char* sP = sX + TILE;                                          // [64 x 64] P, then dS
    const WgMap wm = wg_map(a);
    if (wm.bx < 0) return;
    const int head = wm.head, C = a.heads * DH, pitch = 3 * C;
    for (int e = threadIdx.x; e < NT * 64; e += 256) {
        const int i = e >> 6, j = e & 63;
        sBias[i * BP + j] = (j < NT) ? a.bias_t[((size_t)head * NT + j) * NT + i] : -INFINITY;
    }
    __syncthreads();
    const MaskBits mb = make_mask_bits(a.shift);
    f32x4_t dbacc[4][4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) dbacc[tj][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int w0 = (wm.bx * 4 + wave) * a.wpw, w1 = min(a.n_win_total, w0 + a.wpw);
    // q|k|v bias gradient: column sums of this wavefront's dq / dk / dv rows ([tensor][td], lane = 4 head columns of 16 token rows),
    // summed BEFORE the bf16 rounding of the stored rows; rows of padded tokens are exactly zero (P = dS = 0 there)
    f32x4_t cs[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int td = 0; td < 2; ++td) cs[t][td] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    BwdBufs bufs;
    bufs.qkv = __builtin_amdgcn_make_buffer_rsrc((void*)a.qkv, 0, (int)a.qkv_bytes, 0x00020000);
    bufs.dqkv = __builtin_amdgcn_make_buffer_rsrc((void*)a.dqkv, 0, (int)a.qkv_bytes, 0x00020000);
    bufs.dctx = __builtin_amdgcn_make_buffer_rsrc((void*)a.dctx, 0, (int)(a.qkv_bytes / 3), 0x00020000);
    BwdFrags fr;
    if (w0 < w1) {
        const LaneGeom G0 = window_geom(a, w0);
        load_bwd_frags(bufs, G0, head, C, pitch, g4, fr);
    }
    for (int g = w0; g < w1; ++g) {
        const LaneGeom G = window_geom(a, g);
        [[maybe_unused]] BwdFrags fr_next;
        if (WIDE) {
            const LaneGeom Gn = window_geom(a, min(g + 1, w1 - 1));
            load_bwd_frags(bufs, Gn, head, C, pitch, g4, fr_next);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            *reinterpret_cast<uint4*>(sK + tile_wr_off() + k * 1024) = fr.k[k];
            *reinterpret_cast<uint4*>(sX + tile_wr_off() + k * 1024) = fr.o[k];
        }
        uint4 dsf[4][2];
        // one 16-query block at a time (see softmax_part); dO rows of the block come back from the tile just written
        wave_lds_fence();
        auto block = [&](auto TBc, auto NTIc) {
            constexpr int TB = decltype(TBc)::value, NTI = decltype(NTIc)::value;
            f32x4_t s[4][NTI], dp[4][NTI];
            uint4 ob[NTI];
#pragma unroll
            for (int t = 0; t < NTI; ++t) ob[t] = *reinterpret_cast<const uint4*>(sX + tile_wr_off() + (TB + t) * 1024);
#pragma unroll
            for (int t = 0; t < NTI; ++t)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    s[tj][t] = mfma<T16>(fr.k[tj], fr.q[TB + t], f32x4_t{0.f, 0.f, 0.f, 0.f});
                    dp[tj][t] = mfma<T16>(fr.v[tj], ob[t], f32x4_t{0.f, 0.f, 0.f, 0.f});
                }
            // padded query rows (i >= NT) read the last real bias row; softmax_part zeroes their probabilities
            softmax_part<TB, NTI>(s, a.scale, [&](int tj, int ti) {
                return *reinterpret_cast<const f32x4_t*>(sBias + min(16 * ti + c, NT - 1) * BP + 16 * tj + 4 * g4);
            }, G, mb);
            // dS = P o (dP - rowsum(P o dP)); dbias += dS.  dS is exactly 0 on padded keys and padded queries (P = 0 on both).
            float delta[NTI];
#pragma unroll
            for (int t = 0; t < NTI; ++t) {
                delta[t] = 0.f;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (!dead_key(tj, r)) delta[t] = fmaf(s[tj][t][r], dp[tj][t][r], delta[t]);
            }
#pragma unroll
            for (int t = 0; t < NTI; ++t) delta[t] = rows4_sum(delta[t]);
#pragma unroll
            for (int t = 0; t < NTI; ++t)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (dead_key(tj, r)) { dp[tj][t][r] = 0.f; continue; }
                        const float ds = s[tj][t][r] * (dp[tj][t][r] - delta[t]);
                        dp[tj][t][r] = ds;
                        dbacc[tj][TB + t][r] += ds;
                    }
            // P to the row-major LDS tile [i][j] (for dV); dS fragments stay in registers (for dQ) and follow P into the tile (for dK)
#pragma unroll
            for (int t = 0; t < NTI; ++t) {
                char* prow = sP + (16 * (TB + t) + c) * PROW;
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    dsf[TB + t][sk] = pack_part<T16, NTI>(dp, t, sk);
                    const uint4 pfr = pack_part<T16, NTI>(s, t, sk);
                    *reinterpret_cast<uint2*>(prow + (32 * sk + 4 * g4) * 2) = make_uint2(pfr.x, pfr.y);          // keys 16 (2 sk) + 4 g + r
                    *reinterpret_cast<uint2*>(prow + (32 * sk + 16 + 4 * g4) * 2) = make_uint2(pfr.z, pfr.w);     // keys 16 (2 sk + 1) + 4 g + r
                }
            }
            if (!WIDE) __builtin_amdgcn_sched_barrier(0);      // the next block starts when this one's registers are free
        };
        if constexpr (WIDE) {
            block(std::integral_constant<int, 0>{}, std::integral_constant<int, SWIN_BWD_NTI>{});
            if constexpr (SWIN_BWD_NTI < 4) block(std::integral_constant<int, SWIN_BWD_NTI>{}, std::integral_constant<int, SWIN_BWD_NTI>{});
            if constexpr (SWIN_BWD_NTI < 2) {
                block(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
                block(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
            }
        } else {
            block(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            block(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            block(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
            block(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
        }
        wave_lds_fence();
        // dQ^T[d][i] = scale * sum_j K[j][d] dS[i][j]
        {
            f32x4_t acc[2][4];
#pragma unroll
            for (int td = 0; td < 2; ++td)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) acc[td][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    const uint4 kt = frag_tr_spread(sK, TROW, td, sk);
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti) acc[td][ti] = mfma<T16>(kt, dsf[ti][sk], acc[td][ti]);
                }
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
                if (G.valid[ti]) {
                    store8_buf<T16>(bufs.dqkv, (uint32_t)G.row[ti] * (uint32_t)(pitch * 2) + (uint32_t)((head * DH + 8 * g4) * 2), 0, acc[0][ti], acc[1][ti], a.scale);
                }
            if (a.csum) {
#pragma unroll
                for (int td = 0; td < 2; ++td)
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti) cs[0][td] += acc[td][ti] * a.scale;
            }
        }
        // dV^T[d][j] = sum_i dO[i][d] P[i][j]: both operands are transposed reads (dO tile, P tile)
        {
            f32x4_t acc[2][4];
#pragma unroll
            for (int td = 0; td < 2; ++td)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) acc[td][tj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                uint4 pt[4];
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) pt[tj] = frag_tr(sP, PROW, 16 * tj, sk);
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    const uint4 ot = frag_tr_spread(sX, TROW, td, sk);
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj) acc[td][tj] = mfma<T16>(ot, pt[tj], acc[td][tj]);
                }
            }
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                if (G.valid[tj]) {
                    store8_buf<T16>(bufs.dqkv, (uint32_t)G.row[tj] * (uint32_t)(pitch * 2) + (uint32_t)((head * DH + 8 * g4) * 2), C * 4, acc[0][tj], acc[1][tj], 1.0f);
                }
            if (a.csum) {
#pragma unroll
                for (int td = 0; td < 2; ++td)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj) cs[2][td] += acc[td][tj];
            }
        }
        wave_lds_fence();
        // dK^T[d][j] = scale * sum_i Q[i][d] dS[i][j]: Q (still in registers) and dS replace dO and P in their tiles
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(sX + tile_wr_off() + k * 1024) = fr.q[k];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            char* prow = sP + (16 * ti + c) * PROW;
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                *reinterpret_cast<uint2*>(prow + (32 * sk + 4 * g4) * 2) = make_uint2(dsf[ti][sk].x, dsf[ti][sk].y);
                *reinterpret_cast<uint2*>(prow + (32 * sk + 16 + 4 * g4) * 2) = make_uint2(dsf[ti][sk].z, dsf[ti][sk].w);
            }
        }
        wave_lds_fence();
        if (!WIDE) {   // the next window's operand rows (the last window re-reads itself: no guarded loads), in flight during the dK product
            const LaneGeom Gn = window_geom(a, min(g + 1, w1 - 1));
            load_bwd_frags(bufs, Gn, head, C, pitch, g4, fr);
        } else {
            fr = fr_next;     // requested at the top of this window (a second register set: one wave per SIMD has 512 registers)
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            f32x4_t acc[2][4];
#pragma unroll
            for (int td = 0; td < 2; ++td)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) acc[td][tj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                uint4 st[4];
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) st[tj] = frag_tr(sP, PROW, 16 * tj, sk);
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    const uint4 qt = frag_tr_spread(sX, TROW, td, sk);
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj) acc[td][tj] = mfma<T16>(qt, st[tj], acc[td][tj]);
                }
            }
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                if (G.valid[tj]) {
                    store8_buf<T16>(bufs.dqkv, (uint32_t)G.row[tj] * (uint32_t)(pitch * 2) + (uint32_t)((head * DH + 8 * g4) * 2), C * 2, acc[0][tj], acc[1][tj], a.scale);
                }
            if (a.csum) {
#pragma unroll
                for (int td = 0; td < 2; ++td)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj) cs[1][td] += acc[td][tj] * a.scale;
            }
        }
        wave_lds_fence();
    }
    if (a.csum) {   // one row per wavefront slot: [slot][tensor * C + head * 32 + 8 g + 4 td + e] (the spread column order); empty slots write zeros
        float* wrow = a.csum + (size_t)(wm.bx * 4 + wave) * (3 * C) + head * DH + 8 * g4;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int td = 0; td < 2; ++td) {
                float4 v;
                v.x = row16_sum(cs[t][td][0]); v.y = row16_sum(cs[t][td][1]); v.z = row16_sum(cs[t][td][2]); v.w = row16_sum(cs[t][td][3]);
                if (c == 0) *reinterpret_cast<float4*>(wrow + t * C + 4 * td) = v;
            }
    }
    if (a.dbias_t) {
        // reduce the 4 wavefronts' register accumulators through LDS (the per-wave tiles are free now), then one global
        // atomic per (i, j) per workgroup
        float* red = reinterpret_cast<float*>(wbase);       // [64 (key j)][BP]
        for (int w = 0; w < 4; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* p = red + (16 * tj + 4 * g4 + r) * BP + 16 * ti + c;
                            *p = (w == 0) ? dbacc[tj][ti][r] : *p + dbacc[tj][ti][r];
                        }
            }
        }
        __syncthreads();
        float* part = a.dbias_part ? a.dbias_part + ((size_t)head * a.gx + wm.bx) * (NT * NT) : nullptr;
        for (int e = threadIdx.x; e < NT * NT; e += 256) {
            const int j = e / NT, i = e - j * NT;
            if (part) part[e] = red[j * BP + i];
            else atomicAdd(a.dbias_t + (size_t)head * NT * NT + e, red[j * BP + i]);
        }
    }
}
}  // namespace

// bf16 / f16, window 7, head width 32 only; anything else returns MOREC_E_UNSUPPORTED and the caller falls back to swin.hip's kernels
// csum / csum_rows (backward, optional): scratch for the per-wavefront column sums of dqkv and the number of rows it holds;
// *csum_rows_needed reports how many the launch geometry needs (the caller folds that many rows)
int morec_swin_attn_mfma_launch(const morec_swin_attn_desc* d, const void* qkv, const float* bias_t, void* ctx, const void* dctx,
                                void* dqkv, float* dbias_t, bool backward, hipStream_t s, float* csum, long csum_rows, int* csum_rows_needed) {
    if (!is_h16(d->dtype) || d->window != WS || d->dh != DH) return MOREC_E_UNSUPPORTED;
    if ((d->heads * DH) % 8) return MOREC_E_UNSUPPORTED;
    SwinMArgs a{};
    a.qkv = reinterpret_cast<const bf16*>(qkv); a.bias_t = bias_t; a.ctx = reinterpret_cast<bf16*>(ctx);
    a.dctx = reinterpret_cast<const bf16*>(dctx); a.dqkv = reinterpret_cast<bf16*>(dqkv); a.dbias_t = dbias_t;
    a.n_img = d->n_img; a.H = d->H; a.W = d->W; a.shift = d->shift; a.heads = d->heads; a.scale = d->scale;
    a.n_win_total = d->n_img * (d->H / WS) * (d->W / WS);
    const long tiles = (long)a.n_win_total * d->heads;
    // Windows per wavefront.  Every BACKWARD workgroup ends with the reduction of its four wavefronts' dbias registers through LDS + 2 401 atomics: at the
    // 1 ... 4 windows per wavefront the deep stages get from tiles / 8192 that was a third of its time -- at least 8 there (Swin-T stage 3, 704 images:
    // 182 -> 148 us; Swin-B's stage 2, 352 images: 183 -> 145 us; profiles/r05_swin_attn_pmc.txt).  The forward has no such tail and loses 8-17 % to the
    // smaller grid, so it keeps the plain rule.  MOREC_SWIN_WPW_MIN overrides the backward's floor.
    static const int wpw_min_bwd = [] { const char* e = getenv("MOREC_SWIN_WPW_MIN"); return e ? std::max(1, atoi(e)) : 8; }();
    a.wpw = (int)std::max<long>(backward ? wpw_min_bwd : 1, std::min<long>(32, tiles / 8192));
    a.gx = (a.n_win_total + 4 * a.wpw - 1) / (4 * a.wpw);
    if (csum_rows_needed) *csum_rows_needed = a.gx * 4;
    a.csum = (backward && csum && csum_rows >= (long)a.gx * 4) ? csum : nullptr;
    if (backward && csum && !a.csum) return MOREC_E_UNSUPPORTED;
    dim3 grid(((a.gx + 7) / 8) * 8 * d->heads), block(256);
    const unsigned long long qkv_bytes = (unsigned long long)d->n_img * d->H * d->W * 3 * d->heads * DH * 2;
    if (backward && qkv_bytes >= 0xffffffffull) return MOREC_E_UNSUPPORTED;     // 32-bit buffer offsets in the backward kernel
    a.qkv_bytes = (unsigned)qkv_bytes;
    if (backward && dbias_t && morec_deterministic()) {
        a.dbias_part = morec_det_scratch(s, (size_t)d->heads * a.gx * (NT * NT));
        if (!a.dbias_part) return (int)hipErrorOutOfMemory;
    }
    const size_t lds = (size_t)NT * BP * sizeof(float) + 4 * (2 * TILE + PTILE);
    static const int wide = [] { const char* e = getenv("MOREC_SWIN_BWD_WIDE"); return e ? atoi(e) : 1; }();
    by_h16(d->dtype, [&](auto* t) {
        using T = MOREC_TAG_T(t);
        if (!backward) {
            hipLaunchKernelGGL(swin_attn_fwd_mfma_kernel<T>, grid, block, 0, s, a);
            return;
        }
        static const bool attr_set = [&] {      // thread-safe one-time set-up per storage type (the LDS size is a compile-time function of the tile constants)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&swin_attn_bwd_mfma_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&swin_attn_bwd_mfma_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            return true;
        }();
        (void)attr_set;
        if (wide) hipLaunchKernelGGL((swin_attn_bwd_mfma_kernel<T, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((swin_attn_bwd_mfma_kernel<T, false>), grid, block, lds, s, a);
    });
    MOREC_CHECK_LAUNCH();
    if (a.dbias_part)      // deterministic mode: the workgroups' tiles in window-group order, one fold per head
        for (int h = 0; h < d->heads; ++h) {
            const int rc = morec_det_fold_add(a.dbias_part + (size_t)h * a.gx * (NT * NT), dbias_t + (size_t)h * (NT * NT), a.gx, (size_t)(NT * NT), (size_t)(NT * NT), s);
            if (rc) return rc;
        }
    return MOREC_OK;
}
