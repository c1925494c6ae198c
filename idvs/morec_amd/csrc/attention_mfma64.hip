// attention_mfma64.hip -- the matrix-core attention of attention_mfma.hip for 32 < T <= 64: the reference's abstracts / bodies of 50 tokens
// (T/parameters.py:43-44: num_words_abstract = num_words_body = 50) and behaviour sequences of up to 64 items.  Same contract, same lane
// layouts, one wavefront per (sequence, head); the score tile is 64 x 64 = 4 x 4 MFMA 16x16x32 blocks, a softmax row lives in 4 lanes x 16
// registers, and every product that contracts over keys or queries takes TWO k-steps (slots 0..31 | 32..63 of the row-major LDS tiles, read
// with ds_read_b64_tr_b16 exactly as in the 32-row kernels).  Until round 5 these shapes ran the exact-fp32 VALU kernels even in the 16-bit
// modes (attention.hip); they remain the fallback for head widths that are not a multiple of 32 and the path of the fp32 modes.
// Reference arithmetic: T/model/modules.py:24-49 (SASRec), HF modeling_bert.py BertSelfAttention (eager).
#include "common.hpp"

namespace {
constexpr int TT = 64;                        // rows / columns of the score tile
constexpr int NB = TT / 16;                   // 16-row MFMA blocks per side
constexpr int DCH = 64;                       // head-width chunk staged per pass (elements)
constexpr int PITCH = DCH * 2 + 32;           // bytes per LDS tile row (pitch / 32 odd: b128 and transpose reads conflict-free)
constexpr int TILE = TT * PITCH;              // one [64 x DCH] 16-bit tile
constexpr int PP = TT * 2 + 32;               // bytes per row of the [64 x 64] probability / dS tiles
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

struct AttnMArgs {      // (same fields as attention_mfma.hip's)
    const bf16* qkv;
    const float* key_keep;
    bf16* ctx;          // fwd: output; bwd: dctx input
    bf16* dqkv;
    float* csum;        // bwd, optional: [n_seq][3 H] fp32 column sums of this sequence's dqkv rows
    int n_seq, T, n_heads, dh, causal;
    float scale, mask_value;
    DropRng drop;
    const int32_t* cu;
    int total_rows;
};

__device__ __forceinline__ uint4 load16(const bf16* p) { return *reinterpret_cast<const uint4*>(p); }

// rows [0, T) x columns [col0, col0 + ncols) -> registers (rows >= T repeat row T - 1; see attention_mfma.hip on unguarded loads)
__device__ __forceinline__ void load_tile_regs(const bf16* __restrict__ src, size_t row0, int pitch, int col0, int ncols, int Tlen, uint4 (&v)[8]) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 8 + (lane >> 3), s = lane & 7;
        v[i] = load16(src + (row0 + min(r, Tlen - 1)) * (size_t)pitch + col0 + (s * 8 < ncols ? s * 8 : 0));
    }
}
__device__ __forceinline__ void write_tile_lds(char* __restrict__ tile, const uint4 (&v)[8]) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(tile + (i * 8 + (lane >> 3)) * PITCH + (lane & 7) * 16) = v[i];
}
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ src, size_t row0, int pitch, int col0, int ncols, int Tlen, char* __restrict__ tile) {
    uint4 v[8];
    load_tile_regs(src, row0, pitch, col0, ncols, Tlen, v);
    write_tile_lds(tile, v);
}

// NT fragment: 8 consecutive d of row (blk*16 + lane&15), d offset ks*32 + (lane>>4)*8
__device__ __forceinline__ bf16x8_t frag_nt(const char* tile, int blk, int ks) {
    const int lane = threadIdx.x;
    const uint4 v = *reinterpret_cast<const uint4*>(tile + (blk * 16 + (lane & 15)) * PITCH + (ks * 32 + (lane >> 4) * 8) * 2);
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t frag_nt_global(const bf16* __restrict__ src, size_t row0, int pitch, int col, int blk, int Tlen) {
    const int lane = threadIdx.x;
    const int r = blk * 16 + (lane & 15);
    const uint4 v = load16(src + (row0 + min(r, Tlen - 1)) * (size_t)pitch + col + (lane >> 4) * 8);
    return __builtin_bit_cast(bf16x8_t, v);
}
// transposed fragment over the 32 rows that start at `tile`: lane (c = lane&15, g = lane>>4) receives, for column col0 + c, rows
// 4g .. 4g+3 (elements 0..3) and 16+4g .. 16+4g+3 (elements 4..7)
__device__ __forceinline__ bf16x8_t frag_tr(const char* tile, int pitch_bytes, int col0) {
    const int lane = threadIdx.x;
    const int c = lane & 15, g = lane >> 4;
    const char* p0 = tile + (4 * g + (c >> 2)) * pitch_bytes + (col0 + 4 * (c & 3)) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 16 * pitch_bytes));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
// the same with the block's 16 columns spread as 4 groups of 4 (d-contiguous outputs per lane: attention_mfma.hip)
__device__ __forceinline__ bf16x8_t frag_tr_spread(const char* tile, int pitch_bytes, int col0, int qstride) {
    const int lane = threadIdx.x;
    const int c = lane & 15, g = lane >> 4;
    const char* p0 = tile + (4 * g + (c >> 2)) * pitch_bytes + (col0 + qstride * (c & 3)) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 16 * pitch_bytes));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

template <typename T16, int ND>
__device__ __forceinline__ void store_row_d(bf16* dst, size_t row, int pitch, int col, const f32x4_t (&o)[ND]) {
    static_assert(ND % 2 == 0, "pairs of blocks");
    bf16* p = dst + row * (size_t)pitch + col;
#pragma unroll
    for (int h = 0; h < ND / 2; ++h) {
        uint4 v;
        v.x = h16<T16>::pack2(o[2 * h][0], o[2 * h][1]);
        v.y = h16<T16>::pack2(o[2 * h][2], o[2 * h][3]);
        v.z = h16<T16>::pack2(o[2 * h + 1][0], o[2 * h + 1][1]);
        v.w = h16<T16>::pack2(o[2 * h + 1][2], o[2 * h + 1][3]);
        *reinterpret_cast<uint4*>(p + 8 * h) = v;
        store_b128_guard();
    }
}

// register fragment of a [query][key] quantity over one 32-key k-step: blocks x0 (keys 0..15 of the step), x1 (16..31)
template <typename T16>
__device__ __forceinline__ bf16x8_t frag_regs(const f32x4_t& x0, const f32x4_t& x1) {
    const uint4 v = make_uint4(h16<T16>::pack2(x0[0], x0[1]), h16<T16>::pack2(x0[2], x0[3]), h16<T16>::pack2(x1[0], x1[1]), h16<T16>::pack2(x1[2], x1[3]));
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ void load_keep(const AttnMArgs& a, const float* keep_row, float (&keep)[NB][4]) {
    const int g = threadIdx.x >> 4;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = kb * 16 + 4 * g + r;
            const float v = keep_row[min(j, a.T - 1)];
            keep[kb][r] = (j < a.T) ? v : 0.f;
        }
}

__device__ __forceinline__ void softmax_regs(f32x4_t (&s)[NB][NB], const AttnMArgs& a, const float (&keep)[NB][4]) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
        const int i = qb * 16 + c;
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kb * 16 + 4 * g + r;
                const bool kept = (keep[kb][r] != 0.f) && (!a.causal || j <= i);
                const float v = s[qb][kb][r] * a.scale + (kept ? 0.f : a.mask_value);
                s[qb][kb][r] = (j < a.T) ? v : -INFINITY;
                m = fmaxf(m, s[qb][kb][r]);
            }
        m = rows4_max(m);
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (kb * 16 + 4 * g + r < a.T) ? expf(s[qb][kb][r] - m) : 0.f;
                s[qb][kb][r] = e;
                sum += e;
            }
        sum = rows4_sum(sum);
        const float inv = (i < a.T) ? 1.0f / sum : 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qb][kb][r] *= inv;
    }
}

// keep-mask scale of the lane's cells: element index ((tile * 64 + query) * 64 + key) -- the stream of the 64-wide VALU kernels (attention.hip)
__device__ __forceinline__ void drop_mask_regs(const DropRng& d, uint64_t tile, float (&m)[NB][NB][4]) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            bool kp[4];
            drop_keep_vec<4>(d, (tile * TT + (uint64_t)(qb * 16 + c)) * TT + (uint64_t)(kb * 16 + 4 * g), kp);   // even start
#pragma unroll
            for (int r = 0; r < 4; ++r) m[qb][kb][r] = kp[r] ? d.inv_keep : 0.f;
        }
}

// ctx[query][d0 .. d0 + 16 ND) = P_d V for the four query blocks: two k-steps over the 64 keys
template <typename T16, int ND>
__device__ __forceinline__ void fwd_pv(const AttnMArgs& a, const char* sV, const bf16x8_t (&pf)[NB][2], size_t row0, int H, int col0) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    bf16x8_t vf[2][ND];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int db = 0; db < ND; ++db) vf[st][db] = frag_tr_spread(sV + st * 32 * PITCH, PITCH, 4 * db, 4 * ND);
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
        f32x4_t o[ND];
#pragma unroll
        for (int db = 0; db < ND; ++db) {
            o[db] = h16<T16>::mma16(vf[0][db], pf[qb][0], zero);
            o[db] = h16<T16>::mma16(vf[1][db], pf[qb][1], o[db]);
        }
        const int q = qb * 16 + c;
        if (q < a.T) store_row_d<T16, ND>(a.ctx, row0 + q, H, col0 + 4 * ND * g, o);
    }
}

__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

// dQ = dS K, dK = dS^T Q, dV = P_d^T dO for one head-width chunk of 16 ND columns
template <typename T16, int ND>
__device__ __forceinline__ void bwd_products(const AttnMArgs& a, const char* sQ, const char* sK, const char* sO, const bf16x8_t (&dsf)[NB][2],
                                             const bf16x8_t (&dsT)[NB][2], const bf16x8_t (&pT)[NB][2], size_t row0, int pitch, int H, int col0, int seq) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t cq[ND], ck[ND], cv[ND];
#pragma unroll
    for (int db = 0; db < ND; ++db) { cq[db] = zero; ck[db] = zero; cv[db] = zero; }
    bf16x8_t kt[2][ND], qt[2][ND], ot[2][ND];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int db = 0; db < ND; ++db) {
            kt[st][db] = frag_tr_spread(sK + st * 32 * PITCH, PITCH, 4 * db, 4 * ND);   // K [key slots][d]
            qt[st][db] = frag_tr_spread(sQ + st * 32 * PITCH, PITCH, 4 * db, 4 * ND);   // Q [query slots][d]
            ot[st][db] = frag_tr_spread(sO + st * 32 * PITCH, PITCH, 4 * db, 4 * ND);   // dO [query slots][d]
        }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int r = b * 16 + c;
        f32x4_t dq[ND], dk[ND], dv[ND];
#pragma unroll
        for (int db = 0; db < ND; ++db) {
            dq[db] = h16<T16>::mma16(kt[0][db], dsf[b][0], zero);      // dQ[query r][d] = sum_key dS[r][key] K[key][d]
            dq[db] = h16<T16>::mma16(kt[1][db], dsf[b][1], dq[db]);
            dk[db] = h16<T16>::mma16(qt[0][db], dsT[b][0], zero);      // dK[key r][d] = sum_query dS[query][r] Q[query][d]
            dk[db] = h16<T16>::mma16(qt[1][db], dsT[b][1], dk[db]);
            dv[db] = h16<T16>::mma16(ot[0][db], pT[b][0], zero);       // dV[key r][d] = sum_query P_d[query][r] dO[query][d]
            dv[db] = h16<T16>::mma16(ot[1][db], pT[b][1], dv[db]);
        }
        if (r < a.T) {
            const int dcol = col0 + 4 * ND * g;
            store_row_d<T16, ND>(a.dqkv, row0 + r, pitch, dcol, dq);
            store_row_d<T16, ND>(a.dqkv, row0 + r, pitch, H + dcol, dk);
            store_row_d<T16, ND>(a.dqkv, row0 + r, pitch, 2 * H + dcol, dv);
            if (a.csum) {
#pragma unroll
                for (int db = 0; db < ND; ++db)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cq[db][e] += dq[db][e];
                        ck[db][e] += dk[db][e];
                        cv[db][e] += dv[db][e];
                    }
            }
        }
    }
    if (a.csum) {
#pragma unroll
        for (int db = 0; db < ND; ++db)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cq[db][e] = row16_sum(cq[db][e]);
                ck[db][e] = row16_sum(ck[db][e]);
                cv[db][e] = row16_sum(cv[db][e]);
            }
        if (c == 0) {
            float* w = a.csum + (size_t)seq * 3 * H + col0 + 4 * ND * g;
#pragma unroll
            for (int db = 0; db < ND; ++db) {
                *reinterpret_cast<float4*>(w + 4 * db) = make_float4(cq[db][0], cq[db][1], cq[db][2], cq[db][3]);
                *reinterpret_cast<float4*>(w + H + 4 * db) = make_float4(ck[db][0], ck[db][1], ck[db][2], ck[db][3]);
                *reinterpret_cast<float4*>(w + 2 * H + 4 * db) = make_float4(cv[db][0], cv[db][1], cv[db][2], cv[db][3]);
            }
        }
    }
}

template <typename T16>
__global__ __launch_bounds__(64) void attn_fwd_mfma64_kernel(AttnMArgs a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {      // spare rows of a bucket-padded packed layout: ctx = 0 there
        zero_dead_rows(a.ctx, a.cu, a.n_seq, a.total_rows, (size_t)a.n_heads * a.dh * 2, (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    __shared__ __attribute__((aligned(16))) char sV[TILE];
    const int seq = blockIdx.x / a.n_heads, head = blockIdx.x % a.n_heads;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const size_t row0 = a.cu ? (size_t)a.cu[seq] : (size_t)seq * a.T;
    if (a.cu) a.T = a.cu[seq + 1] - a.cu[seq];
    if (a.T <= 0) return;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t s[NB][NB];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) s[qb][kb] = zero;
    const int nch = (a.dh + DCH - 1) / DCH;
    float keep[NB][4];
    load_keep(a, a.key_keep + row0, keep);
    uint4 pv[8];
    if (nch == 1) load_tile_regs(a.qkv, row0, pitch, 2 * H + head * a.dh, a.dh, a.T, pv);      // the V tile rides with the first fragments
    for (int d = 0; d < a.dh; d += 32) {     // one MFMA k-step per 32 head columns, operands straight from global memory
        bf16x8_t qf[NB], kf[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            qf[b] = frag_nt_global(a.qkv, row0, pitch, head * a.dh + d, b, a.T);
            kf[b] = frag_nt_global(a.qkv, row0, pitch, H + head * a.dh + d, b, a.T);
        }
#pragma unroll
        for (int qb = 0; qb < NB; ++qb)
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) s[qb][kb] = h16<T16>::mma16(kf[kb], qf[qb], s[qb][kb]);
    }
    if (nch == 1) write_tile_lds(sV, pv);
    __syncthreads();
    softmax_regs(s, a, keep);
    if (a.drop.thresh) {
        float m[NB][NB][4];
        drop_mask_regs(a.drop, blockIdx.x, m);
#pragma unroll
        for (int qb = 0; qb < NB; ++qb)
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[qb][kb][r] *= m[qb][kb][r];
    }
    bf16x8_t pf[NB][2];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
        pf[qb][0] = frag_regs<T16>(s[qb][0], s[qb][1]);
        pf[qb][1] = frag_regs<T16>(s[qb][2], s[qb][3]);
    }
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * DCH, nc = min(DCH, a.dh - d0);
        if (nch > 1) {
            stage_tile(a.qkv, row0, pitch, 2 * H + head * a.dh + d0, nc, a.T, sV);
            __syncthreads();
        }
        if (nc == 64) fwd_pv<T16, 4>(a, sV, pf, row0, H, head * a.dh + d0);
        else fwd_pv<T16, 2>(a, sV, pf, row0, H, head * a.dh + d0);
        if (nch > 1) __syncthreads();
    }
}

template <typename T16>
__global__ __launch_bounds__(64) void attn_bwd_mfma64_kernel(AttnMArgs a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {      // spare rows: dqkv = 0 there
        zero_dead_rows(a.dqkv, a.cu, a.n_seq, a.total_rows, (size_t)3 * a.n_heads * a.dh * 2, (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    __shared__ __attribute__((aligned(16))) char sQ[TILE];
    __shared__ __attribute__((aligned(16))) char sK[TILE];
    __shared__ __attribute__((aligned(16))) char sO[TILE];
    __shared__ __attribute__((aligned(16))) char sP[TT * PP];    // dropped probabilities  [query][key]
    __shared__ __attribute__((aligned(16))) char sS[TT * PP];    // dS                     [query][key]
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int seq = blockIdx.x / a.n_heads, head = blockIdx.x % a.n_heads;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const size_t row0 = a.cu ? (size_t)a.cu[seq] : (size_t)seq * a.T;
    if (a.cu) a.T = a.cu[seq + 1] - a.cu[seq];
    if (a.T <= 0) {
        if (a.csum)
            for (int j = lane; j < 3 * a.dh; j += 64) a.csum[(size_t)seq * 3 * H + (j / a.dh) * H + head * a.dh + j % a.dh] = 0.f;
        return;
    }
    const bf16* dctx = a.ctx;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t s[NB][NB], dp[NB][NB];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) { s[qb][kb] = zero; dp[qb][kb] = zero; }
    const int nch = (a.dh + DCH - 1) / DCH;
    float keep[NB][4];
    load_keep(a, a.key_keep + row0, keep);
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * DCH, nc = min(DCH, a.dh - d0);
        if (ch > 0) __syncthreads();
        {   // all global loads of the chunk before the first LDS write
            uint4 pq[8], pk[8], po[8];
            load_tile_regs(a.qkv, row0, pitch, head * a.dh + d0, nc, a.T, pq);
            load_tile_regs(a.qkv, row0, pitch, H + head * a.dh + d0, nc, a.T, pk);
            load_tile_regs(dctx, row0, H, head * a.dh + d0, nc, a.T, po);
            __builtin_amdgcn_sched_barrier(0);
            write_tile_lds(sQ, pq);
            write_tile_lds(sK, pk);
            write_tile_lds(sO, po);
        }
        __syncthreads();
        for (int ks = 0; ks < nc / 32; ++ks) {
            const int vcol = 2 * H + head * a.dh + d0 + ks * 32;     // V is only consumed d-contiguous: no LDS tile
            bf16x8_t qf[NB], kf[NB], of[NB], vf[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                vf[b] = frag_nt_global(a.qkv, row0, pitch, vcol, b, a.T);
                qf[b] = frag_nt(sQ, b, ks);
                kf[b] = frag_nt(sK, b, ks);
                of[b] = frag_nt(sO, b, ks);
            }
#pragma unroll
            for (int qb = 0; qb < NB; ++qb)
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) {
                    s[qb][kb] = h16<T16>::mma16(kf[kb], qf[qb], s[qb][kb]);
                    dp[qb][kb] = h16<T16>::mma16(vf[kb], of[qb], dp[qb][kb]);
                }
        }
    }
    softmax_regs(s, a, keep);
    float msk[NB][NB][4];
    if (a.drop.thresh) drop_mask_regs(a.drop, blockIdx.x, msk);
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (a.drop.thresh) dp[qb][kb][r] *= msk[qb][kb][r];     // dP = dP_dropped o mask / (1 - p)
                delta += s[qb][kb][r] * dp[qb][kb][r];
            }
        delta = rows4_sum(delta);
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dp[qb][kb][r] = s[qb][kb][r] * (dp[qb][kb][r] - delta) * a.scale;   // dS
                if (a.drop.thresh) s[qb][kb][r] *= msk[qb][kb][r];                  // dV takes the dropped probabilities
            }
            const int q = qb * 16 + c, k0 = kb * 16 + 4 * g;
            *reinterpret_cast<uint2*>(sP + q * PP + k0 * 2) = make_uint2(h16<T16>::pack2(s[qb][kb][0], s[qb][kb][1]), h16<T16>::pack2(s[qb][kb][2], s[qb][kb][3]));
            *reinterpret_cast<uint2*>(sS + q * PP + k0 * 2) = make_uint2(h16<T16>::pack2(dp[qb][kb][0], dp[qb][kb][1]), h16<T16>::pack2(dp[qb][kb][2], dp[qb][kb][3]));
        }
    }
    bf16x8_t dsf[NB][2];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
        dsf[qb][0] = frag_regs<T16>(dp[qb][0], dp[qb][1]);
        dsf[qb][1] = frag_regs<T16>(dp[qb][2], dp[qb][3]);
    }
    __syncthreads();
    // transposed [key block][query slots] fragments of dS and P: two k-steps over the 64 queries
    bf16x8_t dsT[NB][2], pT[NB][2];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            dsT[kb][st] = frag_tr(sS + st * 32 * PP, PP, kb * 16);
            pT[kb][st] = frag_tr(sP + st * 32 * PP, PP, kb * 16);
        }
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * DCH, nc = min(DCH, a.dh - d0);
        if (nch > 1) {      // (one chunk: the tiles of the score phase are still in place)
            __syncthreads();
            stage_tile(a.qkv, row0, pitch, head * a.dh + d0, nc, a.T, sQ);
            stage_tile(a.qkv, row0, pitch, H + head * a.dh + d0, nc, a.T, sK);
            stage_tile(dctx, row0, H, head * a.dh + d0, nc, a.T, sO);
            __syncthreads();
        }
        if (nc == 64) bwd_products<T16, 4>(a, sQ, sK, sO, dsf, dsT, pT, row0, pitch, H, head * a.dh + d0, seq);
        else bwd_products<T16, 2>(a, sQ, sK, sO, dsf, dsT, pT, row0, pitch, H, head * a.dh + d0, seq);
    }
}
}  // namespace

int attn_spare_blocks(const morec_attn_desc* d);      // attention.hip
// 32 < T <= 64, head width a multiple of 32, 16-bit: MOREC_OK or an error; MOREC_E_UNSUPPORTED = outside this path
int morec_attn_mfma64_launch(const morec_attn_desc* d, const void* qkv, const float* key_keep, void* ctx_or_dctx, void* dqkv, bool backward,
                             hipStream_t s, float* csum) {
    if (!is_h16(d->dtype) || d->dh % 32 != 0 || d->T > TT) return MOREC_E_UNSUPPORTED;
    AttnMArgs a{reinterpret_cast<const bf16*>(qkv), key_keep, reinterpret_cast<bf16*>(ctx_or_dctx), reinterpret_cast<bf16*>(dqkv), csum, d->n_seq, d->T,
                d->n_heads, d->dh, d->causal, d->scale, d->mask_value, make_drop(d->p_drop, d->seed), d->cu_seqlens, d->total_rows};
    dim3 grid(d->n_seq * d->n_heads + attn_spare_blocks(d)), block(64);
    by_h16(d->dtype, [&](auto* t) {
        using T = MOREC_TAG_T(t);
        if (backward)
            hipLaunchKernelGGL(attn_bwd_mfma64_kernel<T>, grid, block, 0, s, a);
        else
            hipLaunchKernelGGL(attn_fwd_mfma64_kernel<T>, grid, block, 0, s, a);
    });
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
