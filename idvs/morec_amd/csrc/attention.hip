// attention.hip -- small-tile multi-head attention, forward and backward, one wavefront per
// (sequence, head).  The sequences on this path are tiny and fixed (user history S = 20 / 10, titles
// T = 30; abstracts / bodies 50 at most): the problem is "many independent 32 x 32 (64 x 64) score tiles", not long context, so the design is
//   * Q / K / V (and dO) d-chunks staged in LDS as fp32 (chunking makes any head width work:
//     BERT dh = 64, SASRec d_k = 256 ... 2048),
//   * each lane owns a 4 x 4 block of the 32 x 32 score tile in registers (8 x 8 lanes); 8 x 8 blocks of a 64 x 64 tile for 32 < T <= 64,
//   * softmax row statistics by wave64 shuffles across the 8 lanes that share a row,
//   * P (and dS) parked in LDS for the second product.
// HBM traffic is exactly one read of qkv (+ dctx) and one write of ctx (dqkv): the kernel is
// HBM/latency-bound; FLOPs are < 1 % of the encoder's.
// Reference arithmetic: SASRec T/model/encoders.py:24-27 + T/model/modules.py:27-31 (mask built from
// log_mask inside the kernel: key kept iff log_mask[b, j] != 0 and j <= i, additive -1e9);
// BERT: HF BertSelfAttention eager path (additive finfo.min on padded keys).
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"

namespace {
// Tile edge TP = 8 TB: the 8 x 8 lanes of the wavefront own TB x TB blocks of the TP x TP score tile.  TB = 4 (T <= 32: titles of 30 tokens,
// histories of 20 / 10) is the shape everything on the benchmarked path has; TB = 8 (T <= 64) serves the reference's longer inputs --
// abstracts / bodies of 50 tokens (T/parameters.py:43-44), longer behaviour sequences -- on the same code.
template <int TB>
struct Tile {
    static constexpr int TP = 8 * TB;      // padded tile edge
    static constexpr int PP = TP + 4;      // pitch (floats) of the TP x TP probability tiles
};

struct AttnArgs {
    const void* qkv;
    const float* key_keep;
    void* ctx;         // fwd: output; bwd: dctx input
    void* dqkv;        // bwd only
    int n_seq, T, n_heads, dh;
    int causal;
    float scale, mask_value;
    DropRng drop;      // attention-probability dropout (modules.py:30; HF attention_probs_dropout_prob)
    const int32_t* cu; // packed-row offsets (unpadded layout) or nullptr
    int total_rows;    // rows of the packed buffers (>= cu[n_seq]); the spare ones are zero-filled by the blocks behind the grid
};

// stage a [T x DC] chunk (columns col0 + d0 .. of the packed row) into LDS as fp32, zero-padded to TP rows
template <typename T, int DC, int TB>
__device__ __forceinline__ void stage_chunk(const T* __restrict__ src, size_t row0, int pitch, int col0, int d0, int dh,
                                            int Tlen, float* __restrict__ dst) {
    constexpr int P = DC + 4;
    constexpr int VEC = Tile<TB>::TP * DC / 4;  // float4 slots
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < VEC / 64; ++i) {
        const int v = lane + i * 64;
        const int r = v / (DC / 4), c = (v % (DC / 4)) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < Tlen && d0 + c < dh) io<T>::load4(src + (row0 + r) * (size_t)pitch + col0 + d0 + c, o);
        *reinterpret_cast<float4*>(dst + r * P + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// s[r][c] += sum_d X[i0 + r][d] * Y[j0 + c][d] over one staged chunk
template <int DC, int TB>
__device__ __forceinline__ void block_dot(const float* __restrict__ X, const float* __restrict__ Y, int i0, int j0,
                                          float (&s)[TB][TB]) {
    constexpr int P = DC + 4;
#pragma unroll 4
    for (int d = 0; d < DC; d += 4) {
        float4 x[TB], y[TB];
#pragma unroll
        for (int r = 0; r < TB; ++r) x[r] = *reinterpret_cast<const float4*>(X + (i0 + r) * P + d);
#pragma unroll
        for (int c = 0; c < TB; ++c) y[c] = *reinterpret_cast<const float4*>(Y + (j0 + c) * P + d);
#pragma unroll
        for (int r = 0; r < TB; ++r)
#pragma unroll
            for (int c = 0; c < TB; ++c)
                s[r][c] += x[r].x * y[c].x + x[r].y * y[c].y + x[r].z * y[c].z + x[r].w * y[c].w;
    }
}

// masked, scaled softmax of the lane's TB x TB block; rows are shared by the 8 lanes with equal (lane >> 3)
template <int TB>
__device__ __forceinline__ void block_softmax(float (&s)[TB][TB], int i0, int j0, int Tlen, int causal, float scale,
                                              float mask_value, const float* __restrict__ keep_row) {
    float keep[TB];
#pragma unroll
    for (int c = 0; c < TB; ++c) keep[c] = (j0 + c < Tlen) ? keep_row[j0 + c] : 0.f;
#pragma unroll
    for (int r = 0; r < TB; ++r) {
        const int i = i0 + r;
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < TB; ++c) {
            const int j = j0 + c;
            const bool kept = (keep[c] != 0.f) && (!causal || j <= i);
            // reference arithmetic: score * scale + additive mask (the large-magnitude mask absorbs the score)
            const float v = s[r][c] * scale + (kept ? 0.f : mask_value);
            s[r][c] = (j < Tlen) ? v : -INFINITY;
            m = fmaxf(m, s[r][c]);
        }
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        m = fmaxf(m, __shfl_xor(m, 4, 64));
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < TB; ++c) {
            const float e = (j0 + c < Tlen) ? expf(s[r][c] - m) : 0.f;
            s[r][c] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64);
        const float inv = (i < Tlen) ? 1.0f / sum : 0.f;   // padded query rows contribute nothing downstream
#pragma unroll
        for (int c = 0; c < TB; ++c) s[r][c] *= inv;
    }
}

// dropout keep-mask of the lane's TB x TB block: element index ((tile * TP + i) * TP + j)  (TP = 32 for T <= 32: the same stream as
// the MFMA kernels of attention_mfma.hip draw)
template <int TB>
__device__ __forceinline__ void block_drop_mask(const DropRng& d, uint64_t tile, int i0, int j0, float (&m)[TB][TB]) {
    constexpr uint64_t TP = Tile<TB>::TP;
#pragma unroll
    for (int r = 0; r < TB; ++r)
#pragma unroll
        for (int c4 = 0; c4 < TB; c4 += 4) {
            bool kp[4];
            drop_keep_vec<4>(d, (tile * TP + (uint64_t)(i0 + r)) * TP + (uint64_t)(j0 + c4), kp);   // j0 + c4 is a multiple of 4: even start
#pragma unroll
            for (int c = 0; c < 4; ++c) m[r][c4 + c] = kp[c] ? d.inv_keep : 0.f;
        }
}

// o[r][0..CW-1] = sum_k W[k][w0 + r] * V[k][c0 .. c0+CW-1]   (W stored [k][TP + pad]: "weights by row k");
// CW = DC / 8 columns per lane so that the 8 x 8 lane grid covers a [TP x DC] output chunk exactly.
template <int DC, int TB>
__device__ __forceinline__ void block_pv(const float* __restrict__ W, const float* __restrict__ V, int w0, int c0,
                                         int klen, float (&o)[TB][DC / 8]) {
    constexpr int P = DC + 4, CW = DC / 8, PP = Tile<TB>::PP;
#pragma unroll
    for (int r = 0; r < TB; ++r)
#pragma unroll
        for (int c = 0; c < CW; ++c) o[r][c] = 0.f;
    for (int k = 0; k < klen; ++k) {
        float wr[TB];
#pragma unroll
        for (int r4 = 0; r4 < TB; r4 += 4) {
            const float4 w = *reinterpret_cast<const float4*>(W + k * PP + w0 + r4);
            wr[r4] = w.x; wr[r4 + 1] = w.y; wr[r4 + 2] = w.z; wr[r4 + 3] = w.w;
        }
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(V + k * P + c0 + 4 * q);
#pragma unroll
            for (int r = 0; r < TB; ++r) {
                o[r][4 * q + 0] += wr[r] * v.x; o[r][4 * q + 1] += wr[r] * v.y;
                o[r][4 * q + 2] += wr[r] * v.z; o[r][4 * q + 3] += wr[r] * v.w;
            }
        }
    }
}

template <typename T, int CW, int TB>
__device__ __forceinline__ void store_rows(T* __restrict__ dst, size_t row0, int pitch, int col, int r0, int Tlen,
                                           int dcol, int dh, const float (&o)[TB][CW]) {
#pragma unroll
    for (int r = 0; r < TB; ++r) {
        if (r0 + r < Tlen) {
            T* p = dst + (row0 + r0 + r) * (size_t)pitch + col;
#pragma unroll
            for (int q = 0; q < CW / 4; ++q) {
                const float v[4] = {o[r][4 * q], o[r][4 * q + 1], o[r][4 * q + 2], o[r][4 * q + 3]};
                if (dcol + 4 * q < dh) io<T>::store4(p + 4 * q, v);
            }
        }
    }
}

// row r of a TB-wide register block -> TB consecutive floats of an LDS row
template <int TB>
__device__ __forceinline__ void put_row(float* __restrict__ dst, const float (&v)[TB]) {
#pragma unroll
    for (int c = 0; c < TB; c += 4) *reinterpret_cast<float4*>(dst + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
}

template <int DC, int TB>
constexpr size_t attn_fwd_lds() { return (size_t)(2 * Tile<TB>::TP * (DC + 4) + Tile<TB>::TP * Tile<TB>::PP) * sizeof(float); }
template <int DC, int TB>
constexpr size_t attn_bwd_lds() { return (size_t)(4 * Tile<TB>::TP * (DC + 4) + 3 * Tile<TB>::TP * Tile<TB>::PP) * sizeof(float); }

template <typename T, int DC, int TB>
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnArgs a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {      // spare rows of a bucket-padded packed layout: ctx = 0 there
        zero_dead_rows(a.ctx, a.cu, a.n_seq, a.total_rows, (size_t)a.n_heads * a.dh * sizeof(T), (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    constexpr int P = DC + 4, TP = Tile<TB>::TP, PP = Tile<TB>::PP;
    const int seq_ = blockIdx.x / a.n_heads;
    const size_t row0 = a.cu ? (size_t)a.cu[seq_] : (size_t)seq_ * a.T;
    if (a.cu) a.T = a.cu[seq_ + 1] - a.cu[seq_];
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* sA = smem_f;
    float* sB = sA + TP * P;
    float* sPt = sB + TP * P;          // P transposed: [key j][query i]
    const int lane = threadIdx.x;
    const int head = blockIdx.x % a.n_heads;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const int i0 = (lane >> 3) * TB, j0 = (lane & 7) * TB;

    float s[TB][TB];
#pragma unroll
    for (int r = 0; r < TB; ++r)
#pragma unroll
        for (int c = 0; c < TB; ++c) s[r][c] = 0.f;
    for (int d0 = 0; d0 < a.dh; d0 += DC) {
        stage_chunk<T, DC, TB>(qkv, row0, pitch, head * a.dh, d0, a.dh, a.T, sA);
        stage_chunk<T, DC, TB>(qkv, row0, pitch, H + head * a.dh, d0, a.dh, a.T, sB);
        __syncthreads();
        block_dot<DC, TB>(sA, sB, i0, j0, s);
        __syncthreads();
    }
    block_softmax<TB>(s, i0, j0, a.T, a.causal, a.scale, a.mask_value, a.key_keep + row0);
    if (a.drop.thresh) {
        float m[TB][TB];
        block_drop_mask<TB>(a.drop, blockIdx.x, i0, j0, m);
#pragma unroll
        for (int r = 0; r < TB; ++r)
#pragma unroll
            for (int c = 0; c < TB; ++c) s[r][c] *= m[r][c];
    }
#pragma unroll
    for (int c = 0; c < TB; ++c) {
        float col[TB];
#pragma unroll
        for (int r = 0; r < TB; ++r) col[r] = s[r][c];
        put_row<TB>(sPt + (j0 + c) * PP + i0, col);
    }
    __syncthreads();
    T* ctx = reinterpret_cast<T*>(a.ctx);
    constexpr int CW = DC / 8;
    const int c0 = (lane & 7) * CW;
    for (int d0 = 0; d0 < a.dh; d0 += DC) {
        stage_chunk<T, DC, TB>(qkv, row0, pitch, 2 * H + head * a.dh, d0, a.dh, a.T, sA);
        __syncthreads();
        float o[TB][CW];
        block_pv<DC, TB>(sPt, sA, i0, c0, a.T, o);
        store_rows<T, CW, TB>(ctx, row0, H, head * a.dh + d0 + c0, i0, a.T, d0 + c0, a.dh, o);
        __syncthreads();
    }
}

// Backward: recomputes P from Q, K (no forward state kept), then
//   dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(P o dP)) * scale, dQ = dS K, dK = dS^T Q.
template <typename T, int DC, int TB>
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnArgs a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {      // spare rows: dqkv = 0 there
        zero_dead_rows(a.dqkv, a.cu, a.n_seq, a.total_rows, (size_t)3 * a.n_heads * a.dh * sizeof(T), (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    constexpr int P = DC + 4, TP = Tile<TB>::TP, PP = Tile<TB>::PP;
    const int seq_ = blockIdx.x / a.n_heads;
    const size_t row0 = a.cu ? (size_t)a.cu[seq_] : (size_t)seq_ * a.T;
    if (a.cu) a.T = a.cu[seq_ + 1] - a.cu[seq_];
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* sQ = smem_f;
    float* sK = sQ + TP * P;
    float* sV = sK + TP * P;
    float* sO = sV + TP * P;
    float* sP = sO + TP * P;           // P   [query i][key j]
    float* sS = sP + TP * PP;          // dS  [query i][key j]
    float* sSt = sS + TP * PP;         // dS^T [key j][query i]
    const int lane = threadIdx.x;
    const int head = blockIdx.x % a.n_heads;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* dctx = reinterpret_cast<const T*>(a.ctx);
    T* dqkv = reinterpret_cast<T*>(a.dqkv);
    const int i0 = (lane >> 3) * TB, j0 = (lane & 7) * TB;

    float s[TB][TB], dp[TB][TB];
#pragma unroll
    for (int r = 0; r < TB; ++r)
#pragma unroll
        for (int c = 0; c < TB; ++c) { s[r][c] = 0.f; dp[r][c] = 0.f; }
    for (int d0 = 0; d0 < a.dh; d0 += DC) {
        stage_chunk<T, DC, TB>(qkv, row0, pitch, head * a.dh, d0, a.dh, a.T, sQ);
        stage_chunk<T, DC, TB>(qkv, row0, pitch, H + head * a.dh, d0, a.dh, a.T, sK);
        stage_chunk<T, DC, TB>(qkv, row0, pitch, 2 * H + head * a.dh, d0, a.dh, a.T, sV);
        stage_chunk<T, DC, TB>(dctx, row0, H, head * a.dh, d0, a.dh, a.T, sO);
        __syncthreads();
        block_dot<DC, TB>(sQ, sK, i0, j0, s);
        block_dot<DC, TB>(sO, sV, i0, j0, dp);
        __syncthreads();
    }
    block_softmax<TB>(s, i0, j0, a.T, a.causal, a.scale, a.mask_value, a.key_keep + row0);
    float msk[TB][TB];
    if (a.drop.thresh) {
        block_drop_mask<TB>(a.drop, blockIdx.x, i0, j0, msk);
#pragma unroll
        for (int r = 0; r < TB; ++r)
#pragma unroll
            for (int c = 0; c < TB; ++c) dp[r][c] *= msk[r][c];   // dP = dP_dropped o mask / (1 - p)
    }
#pragma unroll
    for (int r = 0; r < TB; ++r) {
        float delta = 0.f;
#pragma unroll
        for (int c = 0; c < TB; ++c) delta += s[r][c] * dp[r][c];
        delta += __shfl_xor(delta, 1, 64);
        delta += __shfl_xor(delta, 2, 64);
        delta += __shfl_xor(delta, 4, 64);
#pragma unroll
        for (int c = 0; c < TB; ++c) dp[r][c] = s[r][c] * (dp[r][c] - delta) * a.scale;   // dp now holds dS
        if (a.drop.thresh) {   // dV uses the DROPPED probabilities
#pragma unroll
            for (int c = 0; c < TB; ++c) s[r][c] *= msk[r][c];
        }
        put_row<TB>(sP + (i0 + r) * PP + j0, s[r]);
        put_row<TB>(sS + (i0 + r) * PP + j0, dp[r]);
    }
#pragma unroll
    for (int c = 0; c < TB; ++c) {
        float col[TB];
#pragma unroll
        for (int r = 0; r < TB; ++r) col[r] = dp[r][c];
        put_row<TB>(sSt + (j0 + c) * PP + i0, col);
    }
    __syncthreads();
    constexpr int CW = DC / 8;
    const int c0 = (lane & 7) * CW;
    for (int d0 = 0; d0 < a.dh; d0 += DC) {
        stage_chunk<T, DC, TB>(qkv, row0, pitch, head * a.dh, d0, a.dh, a.T, sQ);
        stage_chunk<T, DC, TB>(qkv, row0, pitch, H + head * a.dh, d0, a.dh, a.T, sK);
        stage_chunk<T, DC, TB>(dctx, row0, H, head * a.dh, d0, a.dh, a.T, sO);
        __syncthreads();
        float o[TB][CW];
        // dQ[i0..][cols] = sum_j dS^T[j][i0..] * K[j][cols]
        block_pv<DC, TB>(sSt, sK, i0, c0, a.T, o);
        store_rows<T, CW, TB>(dqkv, row0, pitch, head * a.dh + d0 + c0, i0, a.T, d0 + c0, a.dh, o);
        // dK[j..][cols] = sum_i dS[i][j..] * Q[i][cols]   (the lane's row block i0 doubles as its key block)
        block_pv<DC, TB>(sS, sQ, i0, c0, a.T, o);
        store_rows<T, CW, TB>(dqkv, row0, pitch, H + head * a.dh + d0 + c0, i0, a.T, d0 + c0, a.dh, o);
        // dV[j..][cols] = sum_i P[i][j..] * dO[i][cols]
        block_pv<DC, TB>(sP, sO, i0, c0, a.T, o);
        store_rows<T, CW, TB>(dqkv, row0, pitch, 2 * H + head * a.dh + d0 + c0, i0, a.T, d0 + c0, a.dh, o);
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// 64 < T <= 256: longer texts / behaviour sequences than any launcher of the reference sets (T/parameters.py:42-44: 30 / 50 / 50 tokens,
// max_seq_len 20) but that its command line accepts.  Correctness-first form of the same arithmetic on the 32 x 32 register-block helpers
// above: one wavefront per (sequence, head) walks the T / 32 query tiles; a whole row STRIP of scores ([32 queries] x [T keys], fp32) sits in
// LDS, so the softmax is the plain two-pass one over complete rows (the reference's absorb arithmetic, fully masked rows included) and
// nothing is rescaled on line.  Backward: (A) row statistics m, 1 / sum, delta = rowsum(P o dP) of every query into LDS, (B) per query tile
// the dS strip -> dQ, (C) per key tile the dS / P strips over all queries -> dK, dV; every score tile is recomputed in each phase (three
// times the minimum: this path is a fallback, 5 - 20 x slower per token than the matrix-core kernels).
// Dropout stream: element index ((tile * TPD + i) * TPD + j) with TPD = 32 * ceil(T / 32).
constexpr int T_LONG_MAX = 256;
constexpr int LDC = 64;                                   // head-width chunk of the long kernels
constexpr int LP = LDC + 4, LPP = Tile<4>::PP, LCW = LDC / 8;

// s[r][c] = sum_d X[qt*32 + i0 + r][d] * Y[kt*32 + j0 + c][d] over the whole head width (chunks restaged per call)
template <typename T>
__device__ __forceinline__ void long_dot(const T* __restrict__ xsrc, int xpitch, int xcol, const T* __restrict__ ysrc, int ypitch, int ycol,
                                         size_t row0, int dh, int Tlen, int qt, int kt, float* sX, float* sY, float (&s)[4][4]) {
    const int lane = threadIdx.x, i0 = (lane >> 3) * 4, j0 = (lane & 7) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) s[r][c] = 0.f;
    for (int d0 = 0; d0 < dh; d0 += LDC) {
        stage_chunk<T, LDC, 4>(xsrc, row0 + (size_t)qt * 32, xpitch, xcol, d0, dh, min(32, Tlen - qt * 32), sX);
        stage_chunk<T, LDC, 4>(ysrc, row0 + (size_t)kt * 32, ypitch, ycol, d0, dh, min(32, Tlen - kt * 32), sY);
        __syncthreads();
        block_dot<LDC, 4>(sX, sY, i0, j0, s);
        __syncthreads();
    }
}

// reference arithmetic of one score: scaled + additive mask; keys >= T never enter (-inf)
__device__ __forceinline__ float long_masked(float s, int i, int j, int Tlen, int causal, float scale, float mask_value, const float* __restrict__ keep_row) {
    if (j >= Tlen) return -INFINITY;
    const bool kept = (keep_row[j] != 0.f) && (!causal || j <= i);
    return s * scale + (kept ? 0.f : mask_value);
}

// o[r][c] += sum_{k < klen} W[k][w0 + r] * V[k][c0 + c]
__device__ __forceinline__ void long_pv_acc(const float* __restrict__ W, const float* __restrict__ V, int w0, int c0, int klen, float (&o)[4][LCW]) {
    for (int k = 0; k < klen; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(W + k * LPP + w0);
        const float wr[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < LCW / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(V + k * LP + c0 + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r][4 * q + 0] += wr[r] * v.x; o[r][4 * q + 1] += wr[r] * v.y;
                o[r][4 * q + 2] += wr[r] * v.z; o[r][4 * q + 3] += wr[r] * v.w;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_fwd_long_kernel(AttnArgs a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {
        zero_dead_rows(a.ctx, a.cu, a.n_seq, a.total_rows, (size_t)a.n_heads * a.dh * sizeof(T), (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    const int seq_ = blockIdx.x / a.n_heads, head = blockIdx.x % a.n_heads;
    const size_t row0 = a.cu ? (size_t)a.cu[seq_] : (size_t)seq_ * a.T;
    const int TPD = ((a.T + 31) / 32) * 32;              // mask-stream pitch: from the descriptor's T, not the sequence's own length
    if (a.cu) a.T = a.cu[seq_ + 1] - a.cu[seq_];
    if (a.T <= 0) return;
    const int NT = (a.T + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* sA = smem_f;
    float* sB = sA + 32 * LP;
    float* strip = sB + 32 * LP;                         // [key][query of this tile], pitch LPP
    const int lane = threadIdx.x, i0 = (lane >> 3) * 4, j0 = (lane & 7) * 4;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    T* ctx = reinterpret_cast<T*>(a.ctx);
    const float* keep_row = a.key_keep + row0;
    for (int qt = 0; qt < NT; ++qt) {
        for (int kt = 0; kt < NT; ++kt) {
            float s[4][4];
            long_dot<T>(qkv, pitch, head * a.dh, qkv, pitch, H + head * a.dh, row0, a.dh, a.T, qt, kt, sA, sB, s);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float col[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) col[r] = long_masked(s[r][c], qt * 32 + i0 + r, kt * 32 + j0 + c, a.T, a.causal, a.scale, a.mask_value, keep_row);
                put_row<4>(strip + (kt * 32 + j0 + c) * LPP + i0, col);
            }
        }
        __syncthreads();
        {   // softmax of row q = lane & 31 over the keys j = h, h + 2, ... (h = lane >> 5), the two halves combined by one exchange
            const int q = lane & 31, h = lane >> 5, i = qt * 32 + q;
            float m = -INFINITY;
            for (int j = h; j < a.T; j += 2) m = fmaxf(m, strip[j * LPP + q]);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
            for (int j = h; j < a.T; j += 2) sum += expf(strip[j * LPP + q] - m);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = (i < a.T) ? 1.0f / sum : 0.f;
            for (int j = h; j < NT * 32; j += 2) {
                float p = (j < a.T) ? expf(strip[j * LPP + q] - m) * inv : 0.f;
                if (a.drop.thresh) p = drop_keep(a.drop, ((uint64_t)blockIdx.x * TPD + (uint64_t)i) * TPD + (uint64_t)j) ? p * a.drop.inv_keep : 0.f;
                strip[j * LPP + q] = p;
            }
        }
        __syncthreads();
        const int c0 = (lane & 7) * LCW;
        for (int d0 = 0; d0 < a.dh; d0 += LDC) {
            float o[4][LCW];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < LCW; ++c) o[r][c] = 0.f;
            for (int kt = 0; kt < NT; ++kt) {
                stage_chunk<T, LDC, 4>(qkv, row0 + (size_t)kt * 32, pitch, 2 * H + head * a.dh, d0, a.dh, min(32, a.T - kt * 32), sA);
                __syncthreads();
                long_pv_acc(strip + kt * 32 * LPP, sA, i0, c0, min(32, a.T - kt * 32), o);
                __syncthreads();
            }
            store_rows<T, LCW, 4>(ctx, row0 + (size_t)qt * 32, H, head * a.dh + d0 + c0, i0, min(32, a.T - qt * 32), d0 + c0, a.dh, o);
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_long_kernel(AttnArgs a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {
        zero_dead_rows(a.dqkv, a.cu, a.n_seq, a.total_rows, (size_t)3 * a.n_heads * a.dh * sizeof(T), (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    const int seq_ = blockIdx.x / a.n_heads, head = blockIdx.x % a.n_heads;
    const size_t row0 = a.cu ? (size_t)a.cu[seq_] : (size_t)seq_ * a.T;
    const int TPD = ((a.T + 31) / 32) * 32;
    if (a.cu) a.T = a.cu[seq_ + 1] - a.cu[seq_];
    if (a.T <= 0) return;
    const int NT = (a.T + 31) / 32, TPAD = NT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* sA = smem_f;                                  // Q / dO side tile
    float* sB = sA + 32 * LP;                            // K / V side tile
    float* U1 = sB + 32 * LP;                            // strip 1: [TPAD][LPP]
    float* U2 = U1 + T_LONG_MAX * LPP;                   // strip 2
    float* st_m = U2 + T_LONG_MAX * LPP;                 // row statistics of every query: max, 1 / sum, delta
    float* st_i = st_m + T_LONG_MAX;
    float* st_d = st_i + T_LONG_MAX;
    const int lane = threadIdx.x, i0 = (lane >> 3) * 4, j0 = (lane & 7) * 4;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* dctx = reinterpret_cast<const T*>(a.ctx);
    T* dqkv = reinterpret_cast<T*>(a.dqkv);
    const float* keep_row = a.key_keep + row0;
    const int colQ = head * a.dh, colK = H + head * a.dh, colV = 2 * H + head * a.dh;
    auto keep_scale = [&](int i, int j) -> float {       // dropout factor of element (i, j): 0 or 1 / (1 - p)
        if (!a.drop.thresh) return 1.f;
        return drop_keep(a.drop, ((uint64_t)blockIdx.x * TPD + (uint64_t)i) * TPD + (uint64_t)j) ? a.drop.inv_keep : 0.f;
    };
    // ---- phase A: statistics.  U1 = masked scaled scores [key][q], U2 = dP o mask [key][q] of one query tile at a time
    for (int qt = 0; qt < NT; ++qt) {
        for (int kt = 0; kt < NT; ++kt) {
            float s[4][4], dp[4][4];
            long_dot<T>(qkv, pitch, colQ, qkv, pitch, colK, row0, a.dh, a.T, qt, kt, sA, sB, s);
            long_dot<T>(dctx, H, colQ, qkv, pitch, colV, row0, a.dh, a.T, qt, kt, sA, sB, dp);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float col[4], dcol[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = qt * 32 + i0 + r, j = kt * 32 + j0 + c;
                    col[r] = long_masked(s[r][c], i, j, a.T, a.causal, a.scale, a.mask_value, keep_row);
                    dcol[r] = dp[r][c] * keep_scale(i, j);
                }
                put_row<4>(U1 + (kt * 32 + j0 + c) * LPP + i0, col);
                put_row<4>(U2 + (kt * 32 + j0 + c) * LPP + i0, dcol);
            }
        }
        __syncthreads();
        {
            const int q = lane & 31, h = lane >> 5, i = qt * 32 + q;
            float m = -INFINITY;
            for (int j = h; j < a.T; j += 2) m = fmaxf(m, U1[j * LPP + q]);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
            for (int j = h; j < a.T; j += 2) sum += expf(U1[j * LPP + q] - m);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = (i < a.T) ? 1.0f / sum : 0.f;
            float delta = 0.f;
            for (int j = h; j < a.T; j += 2) delta += expf(U1[j * LPP + q] - m) * inv * U2[j * LPP + q];
            delta += __shfl_xor(delta, 32, 64);
            if (h == 0) { st_m[i] = m; st_i[i] = inv; st_d[i] = delta; }
        }
        __syncthreads();
    }
    const int c0 = (lane & 7) * LCW;
    // P and dS of the lane's 4 x 4 block of tile (qt, kt) from the stored statistics
    auto tile_p_ds = [&](int qt, int kt, float (&p)[4][4], float (&ds)[4][4]) {
        float s[4][4], dp[4][4];
        long_dot<T>(qkv, pitch, colQ, qkv, pitch, colK, row0, a.dh, a.T, qt, kt, sA, sB, s);
        long_dot<T>(dctx, H, colQ, qkv, pitch, colV, row0, a.dh, a.T, qt, kt, sA, sB, dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = qt * 32 + i0 + r;
            const float m = st_m[i], inv = st_i[i], delta = st_d[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = kt * 32 + j0 + c;
                const float v = long_masked(s[r][c], i, j, a.T, a.causal, a.scale, a.mask_value, keep_row);
                const float pr = (j < a.T) ? expf(v - m) * inv : 0.f;
                const float ks = keep_scale(i, j);
                ds[r][c] = pr * (dp[r][c] * ks - delta) * a.scale;
                p[r][c] = pr * ks;                       // dV uses the DROPPED probabilities
            }
        }
    };
    // ---- phase B: dQ.  U1 = dS^T [key][q] of the query tile
    for (int qt = 0; qt < NT; ++qt) {
        for (int kt = 0; kt < NT; ++kt) {
            float p[4][4], ds[4][4];
            tile_p_ds(qt, kt, p, ds);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float col[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) col[r] = ds[r][c];
                put_row<4>(U1 + (kt * 32 + j0 + c) * LPP + i0, col);
            }
        }
        __syncthreads();
        for (int d0 = 0; d0 < a.dh; d0 += LDC) {
            float o[4][LCW];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < LCW; ++c) o[r][c] = 0.f;
            for (int kt = 0; kt < NT; ++kt) {
                stage_chunk<T, LDC, 4>(qkv, row0 + (size_t)kt * 32, pitch, colK, d0, a.dh, min(32, a.T - kt * 32), sB);
                __syncthreads();
                long_pv_acc(U1 + kt * 32 * LPP, sB, i0, c0, min(32, a.T - kt * 32), o);
                __syncthreads();
            }
            store_rows<T, LCW, 4>(dqkv, row0 + (size_t)qt * 32, pitch, colQ + d0 + c0, i0, min(32, a.T - qt * 32), d0 + c0, a.dh, o);
        }
        __syncthreads();
    }
    // ---- phase C: dK, dV.  U1 = dS [q][key of this tile], U2 = P (dropped) [q][key of this tile], all queries
    for (int kt = 0; kt < NT; ++kt) {
        for (int qt = 0; qt < NT; ++qt) {
            float p[4][4], ds[4][4];
            tile_p_ds(qt, kt, p, ds);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                put_row<4>(U1 + (qt * 32 + i0 + r) * LPP + j0, ds[r]);
                put_row<4>(U2 + (qt * 32 + i0 + r) * LPP + j0, p[r]);
            }
        }
        __syncthreads();
        for (int d0 = 0; d0 < a.dh; d0 += LDC) {
            float ok[4][LCW], ov[4][LCW];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < LCW; ++c) { ok[r][c] = 0.f; ov[r][c] = 0.f; }
            for (int qt = 0; qt < NT; ++qt) {
                const int qlen = min(32, a.T - qt * 32);
                stage_chunk<T, LDC, 4>(qkv, row0 + (size_t)qt * 32, pitch, colQ, d0, a.dh, qlen, sA);
                stage_chunk<T, LDC, 4>(dctx, row0 + (size_t)qt * 32, H, colQ, d0, a.dh, qlen, sB);
                __syncthreads();
                long_pv_acc(U1 + qt * 32 * LPP, sA, i0, c0, qlen, ok);      // dK[key i0 ..][cols] += sum_q dS[q][key] Q[q][cols]
                long_pv_acc(U2 + qt * 32 * LPP, sB, i0, c0, qlen, ov);      // dV[key i0 ..][cols] += sum_q P[q][key] dO[q][cols]
                __syncthreads();
            }
            store_rows<T, LCW, 4>(dqkv, row0 + (size_t)kt * 32, pitch, colK + d0 + c0, i0, min(32, a.T - kt * 32), d0 + c0, a.dh, ok);
            store_rows<T, LCW, 4>(dqkv, row0 + (size_t)kt * 32, pitch, colV + d0 + c0, i0, min(32, a.T - kt * 32), d0 + c0, a.dh, ov);
        }
        __syncthreads();
    }
    (void)TPAD;
}

constexpr size_t attn_long_fwd_lds() { return (size_t)(2 * 32 * LP + T_LONG_MAX * LPP) * sizeof(float); }
constexpr size_t attn_long_bwd_lds() { return (size_t)(2 * 32 * LP + 2 * T_LONG_MAX * LPP + 3 * T_LONG_MAX) * sizeof(float); }

template <bool BWD>
int launch_long(const morec_attn_desc* d, const AttnArgs& a, dim3 grid, hipStream_t s) {
    if (!by_dtype(d->dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            if constexpr (!BWD) {
                hipLaunchKernelGGL((attn_fwd_long_kernel<T>), grid, dim3(64), attn_long_fwd_lds(), s, a);
            } else {
                static const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_long_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_long_bwd_lds());
                (void)rc;
                hipLaunchKernelGGL((attn_bwd_long_kernel<T>), grid, dim3(64), attn_long_bwd_lds(), s, a);
            }
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

constexpr int T_MAX = T_LONG_MAX;     // 256: the long kernels above; 64 for the tile kernels (launch_valu)

// one launch of the VALU kernels (forward: d-chunks of 64; backward: of 32), tile edge by sequence length
template <bool BWD>
int launch_valu(const morec_attn_desc* d, const AttnArgs& a, dim3 grid, hipStream_t s) {
    const bool big = d->T > Tile<4>::TP;
    if (!by_dtype(d->dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            if constexpr (!BWD) {
                if (big) {
                    static const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<T, 64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_fwd_lds<64, 8>());
                    (void)rc;
                    hipLaunchKernelGGL((attn_fwd_kernel<T, 64, 8>), grid, dim3(64), (attn_fwd_lds<64, 8>()), s, a);
                } else {
                    hipLaunchKernelGGL((attn_fwd_kernel<T, 64, 4>), grid, dim3(64), (attn_fwd_lds<64, 4>()), s, a);
                }
            } else {
                if (big) {
                    static const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<T, 32, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_bwd_lds<32, 8>());
                    (void)rc;
                    hipLaunchKernelGGL((attn_bwd_kernel<T, 32, 8>), grid, dim3(64), (attn_bwd_lds<32, 8>()), s, a);
                } else {
                    hipLaunchKernelGGL((attn_bwd_kernel<T, 32, 4>), grid, dim3(64), (attn_bwd_lds<32, 4>()), s, a);
                }
            }
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

int check_desc(const morec_attn_desc* d) {
    if (!d) return MOREC_E_ARG;
    if (d->n_seq <= 0 || d->T <= 0 || d->n_heads <= 0 || d->dh <= 0) return MOREC_E_ARG;
    if (d->T > T_MAX) return MOREC_E_UNSUPPORTED;      // 256: the strip kernels (64 < T <= 256); beyond, a row strip no longer fits LDS
    if (d->dh % 8) return MOREC_E_ALIGN;
    if (d->p_drop < 0.f || d->p_drop >= 1.f) return MOREC_E_ARG;
    if (d->total_rows < 0 || d->spare_rows_max < 0 || (d->spare_rows_max > 0 && (!d->cu_seqlens || d->total_rows <= 0))) return MOREC_E_ARG;
    return MOREC_OK;
}
}  // namespace
// blocks appended behind the (sequence, head) grid: 16 spare rows each (morec_attn_desc.spare_rows_max)
int attn_spare_blocks(const morec_attn_desc* d) { return (d->cu_seqlens && d->total_rows > 0) ? (d->spare_rows_max + 15) / 16 : 0; }

// bf16 fast path on the matrix cores (attention_mfma.hip); MOREC_E_UNSUPPORTED = shape outside it
int morec_attn_mfma_launch(const morec_attn_desc* d, const void* qkv, const float* key_keep, void* ctx_or_dctx,
                           void* dqkv, bool backward, hipStream_t s, float* csum = nullptr);
int colsum_f32_launch(const float* in, float* out, int rows, int N, hipStream_t s);
// fp32 path on the matrix cores (attention_f32mfma.hip: T <= 32, head width a multiple of 16); MOREC_E_UNSUPPORTED = shape outside it
int morec_attn_f32mfma_launch(const morec_attn_desc* d, const void* qkv, const float* key_keep, void* ctx_or_dctx, void* dqkv, bool backward,
                              hipStream_t s);

// A 16-bit problem outside the matrix-core path's shape rules (T <= 32, head width a multiple of 32) runs on the exact-fp32 VALU
// kernels -- correct, several times slower.  Said ONCE per process on stderr (MOREC_QUIET=1 silences it) so that e.g. a run with
// num_words_title > 32 does not lose the MFMA attention without a trace.
static void warn_valu_fallback(const morec_attn_desc* d) {
    if (!is_h16(d->dtype)) return;
    static const bool once = [](const morec_attn_desc* q) {
        const char* e = getenv("MOREC_QUIET");
        if (!(e && e[0] == '1'))
            fprintf(stderr, "libmorec_hip: attention with T = %d, head width %d is outside the MFMA path (T <= 64, head width %% 32 == 0): "
                            "running the VALU kernels for this and every later such call\n", q->T, q->dh);
        return true;
    }(d);
    (void)once;
}

extern "C" int morec_attn_fwd(const morec_attn_desc* d, const void* qkv, const float* key_keep, void* ctx,
                              void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!qkv || !key_keep || !ctx) return MOREC_E_ARG;
    AttnArgs a{qkv, key_keep, ctx, nullptr, d->n_seq, d->T, d->n_heads, d->dh, d->causal, d->scale, d->mask_value,
               make_drop(d->p_drop, d->seed), d->cu_seqlens, d->total_rows};
    dim3 grid(d->n_seq * d->n_heads + attn_spare_blocks(d)), block(64);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    rc = morec_attn_mfma_launch(d, qkv, key_keep, ctx, nullptr, false, s);
    if (rc != MOREC_E_UNSUPPORTED) return rc;
    rc = morec_attn_f32mfma_launch(d, qkv, key_keep, ctx, nullptr, false, s);
    if (rc != MOREC_E_UNSUPPORTED) return rc;
    warn_valu_fallback(d);
    (void)block;
    if (d->T > Tile<8>::TP) return launch_long<false>(d, a, grid, s);
    return launch_valu<false>(d, a, grid, s);
}

extern "C" int morec_attn_bwd(const morec_attn_desc* d, const void* qkv, const float* key_keep, const void* dctx,
                              void* dqkv, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!qkv || !key_keep || !dctx || !dqkv) return MOREC_E_ARG;
    AttnArgs a{qkv, key_keep, const_cast<void*>(dctx), dqkv, d->n_seq, d->T, d->n_heads, d->dh, d->causal, d->scale,
               d->mask_value, make_drop(d->p_drop, d->seed), d->cu_seqlens, d->total_rows};
    dim3 grid(d->n_seq * d->n_heads + attn_spare_blocks(d)), block(64);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    rc = morec_attn_mfma_launch(d, qkv, key_keep, const_cast<void*>(dctx), dqkv, true, s);
    if (rc != MOREC_E_UNSUPPORTED) return rc;
    rc = morec_attn_f32mfma_launch(d, qkv, key_keep, const_cast<void*>(dctx), dqkv, true, s);
    if (rc != MOREC_E_UNSUPPORTED) return rc;
    warn_valu_fallback(d);
    (void)block;
    if (d->T > Tile<8>::TP) return launch_long<true>(d, a, grid, s);
    return launch_valu<true>(d, a, grid, s);
}

// morec_attn_bwd + the bias gradient of the fused q|k|v projection: dbias[3 H] += column sums of the dqkv rows (MFMA path: fp32
// sums of the rows before their bf16 rounding; the fallback sums the stored rows).
// On the MFMA path every (sequence, head) wavefront leaves its own column sums in ws ([n_seq][3 H] fp32) and one small kernel
// folds them -- instead of a second pass over the [rows x 3 H] tensor (54 us of a 115 us attention backward at 51200 rows,
// profiles/r02b_bench_kernel_stats.csv).  Other shapes / dtypes: the plain backward followed by morec_colsum.
extern "C" int morec_attn_bwd_dbias(const morec_attn_desc* d, const void* qkv, const float* key_keep, const void* dctx,
                                    void* dqkv, int rows, float* dbias, float* ws, size_t ws_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!qkv || !key_keep || !dctx || !dqkv || !dbias || rows <= 0) return MOREC_E_ARG;
    const int H3 = 3 * d->n_heads * d->dh;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (ws && ws_bytes >= (size_t)d->n_seq * H3 * sizeof(float)) {
        rc = morec_attn_mfma_launch(d, qkv, key_keep, const_cast<void*>(dctx), dqkv, true, s, ws);
        if (rc == MOREC_OK) return colsum_f32_launch(ws, dbias, d->n_seq, H3, s);
        if (rc != MOREC_E_UNSUPPORTED) return rc;
    }
    rc = morec_attn_bwd(d, qkv, key_keep, dctx, dqkv, stream);
    if (rc) return rc;
    return morec_colsum(dqkv, dbias, rows, H3, H3, d->dtype, stream);
}
