// attention_mfma.hip -- bf16 small-tile attention on the matrix cores (T <= 32, head width a multiple of 32).
//
// Same contract as attention.hip (which stays the exact-fp32 path): one wavefront per (sequence, head), no
// forward state kept for the backward, counter-based dropout on the probabilities.  What changes is where
// the arithmetic runs:
//   * S = Q K^T and dP = dO V^T: v_mfma_f32_16x16x32_bf16 with both operands read d-contiguous (NT form)
//     from LDS tiles, issued "swapped" so a lane ends up with S[query = qb*16 + (lane&15)]
//     [key = kb*16 + 4*(lane>>4) + r] -- a whole softmax row lives in 4 lanes x 8 registers, the row
//     max / sum are two shuffles (xor 16, 32);
//   * the second products (P V, dS K, dS^T Q, P^T dO) contract over keys or queries, i.e. over the ROW index
//     of the row-major LDS tiles: those fragments come from ds_read_b64_tr_b16 (hardware transpose read,
//     semantics measured in profiles/r01_probe_mfma_trb16.txt: lane c of a 16-lane group receives
//     element (c & 3) of the 8-byte words addressed by lanes 4j + (c >> 2), j = 0..3).  The k-slot <-> key
//     permutation this induces is the same one the packed P / dS register fragments use, so nothing is shuffled.
// LDS rows carry a 32-byte pad (pitch / 32 odd) which makes both the b128 and the transpose reads conflict-free.
#include <stdlib.h>
#include "common.hpp"

namespace {
constexpr int DCH = 64;                       // head-width chunk staged per pass (elements)
constexpr int PITCH = DCH * 2 + 32;           // bytes per LDS tile row
constexpr int TILE = 32 * PITCH;              // one [32 x DCH] bf16 tile
constexpr int PP = 32 * 2 + 32;               // bytes per row of the [32 x 32] bf16 probability tiles
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

struct AttnMArgs {
    const bf16* qkv;
    const float* key_keep;
    bf16* ctx;          // fwd: output; bwd: dctx input
    bf16* dqkv;
    float* csum;        // bwd, optional: [n_seq][3 H] fp32 column sums of this sequence's dqkv rows (before their bf16 rounding), folded by the caller
    int n_seq, T, n_heads, dh, causal;
    float scale, mask_value;
    DropRng drop;
    const int32_t* cu;  // packed-row offsets (unpadded layout) or nullptr
    int total_rows;     // rows of the packed buffers; the spare ones (>= cu[n_seq]) are zero-filled by the blocks behind the grid
};


// No guarded loads in these kernels.  `if (row < T) v = *p;` -- and equally `ok ? *p : 0`, which the compiler turns back into a
// branch -- makes every load its own basic block that ends in s_waitcnt vmcnt(0): the 12 tile loads, 4 fragment loads and 8
// padding-flag loads of a wavefront then run as two dozen SERIAL memory round trips.  Instead every address is clamped to
// the last valid row of the sequence (T >= 1 is checked at kernel entry) and the duplicate rows are left in place: a row
// >= T is only ever multiplied by a probability / dS entry that is exactly 0 (masked key, or query row with inv = 0), and
// rows >= T of the outputs are never stored.  Columns past the head width (dh = 32 in a 64-wide tile) are never read.
__device__ __forceinline__ uint4 load16(const bf16* p) { return *reinterpret_cast<const uint4*>(p); }

// Rows [0, T) x columns [col0, col0 + ncols) of a row-major global matrix -> registers -> LDS tile (rows >= T repeat row T - 1), in two
// halves: ALL global loads of a kernel phase are issued before the first LDS write.  (Interleaved load / ds_write pairs are
// kept in program order -- the generic LDS pointer may alias the source as far as the compiler knows -- and every pair then
// costs a full memory round trip.)
__device__ __forceinline__ void load_tile_regs(const bf16* __restrict__ src, size_t row0, int pitch, int col0, int ncols,
                                               int Tlen, uint4 (&v)[4]) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 8 + (lane >> 3), s = lane & 7;
        v[i] = load16(src + (row0 + min(r, Tlen - 1)) * (size_t)pitch + col0 + (s * 8 < ncols ? s * 8 : 0));
    }
}
__device__ __forceinline__ void write_tile_lds(char* __restrict__ tile, const uint4 (&v)[4]) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(tile + (i * 8 + (lane >> 3)) * PITCH + (lane & 7) * 16) = v[i];
}

__device__ __forceinline__ void stage_tile(const bf16* __restrict__ src, size_t row0, int pitch, int col0, int ncols,
                                           int Tlen, char* __restrict__ tile) {
    uint4 v[4];
    load_tile_regs(src, row0, pitch, col0, ncols, Tlen, v);
    write_tile_lds(tile, v);
}

// NT fragment: 8 consecutive d of row (blk*16 + lane&15), d offset ks*32 + (lane>>4)*8
__device__ __forceinline__ bf16x8_t frag_nt(const char* tile, int blk, int ks) {
    const int lane = threadIdx.x;
    const uint4 v = *reinterpret_cast<const uint4*>(tile + (blk * 16 + (lane & 15)) * PITCH + (ks * 32 + (lane >> 4) * 8) * 2);
    return __builtin_bit_cast(bf16x8_t, v);
}

// NT fragment straight from global memory (no LDS tile): 8 consecutive d of row (blk*16 + lane&15), rows >= Tlen repeat row Tlen - 1.
// Used for the operands that are only ever consumed d-contiguous (Q / K in the forward scores, V in the backward dP):
// every LDS tile dropped raises the number of resident wavefronts of this latency-bound kernel.
__device__ __forceinline__ bf16x8_t frag_nt_global(const bf16* __restrict__ src, size_t row0, int pitch, int col, int blk, int Tlen) {
    const int lane = threadIdx.x;
    const int r = blk * 16 + (lane & 15);
    const uint4 v = load16(src + (row0 + min(r, Tlen - 1)) * (size_t)pitch + col + (lane >> 4) * 8);
    return __builtin_bit_cast(bf16x8_t, v);
}

// transposed fragment of a row-major tile: lane (c = lane&15, g = lane>>4) receives, for column col0 + c, the rows
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

// Same read with the 16 columns of the block spread as 4 groups of 4: lane c receives column col0 + qstride * (c >> 2) + (c & 3).
// With col0 = 4 * db and qstride = 4 * NB the MFMA outputs of NB consecutive blocks land d-CONTIGUOUS in a lane
// (d = 4 NB g + 4 db + r): a row is then written as 4 lanes x 8 NB bytes (one full 128-byte line at NB = 4) instead of
// NB separate 32-byte pieces -- the partial-line stores were what bound these kernels (writes at ~1 TB/s).
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

// store the 4 * NB consecutive d values of one row held by the lane (NB accumulator blocks, 16-byte pieces)
template <typename T16, int NB>
__device__ __forceinline__ void store_row_d(bf16* dst, size_t row, int pitch, int col, const f32x4_t (&o)[NB]) {
    static_assert(NB % 2 == 0, "pairs of blocks");
    bf16* p = dst + row * (size_t)pitch + col;
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
        uint4 v;
        v.x = h16<T16>::pack2(o[2 * h][0], o[2 * h][1]);
        v.y = h16<T16>::pack2(o[2 * h][2], o[2 * h][3]);
        v.z = h16<T16>::pack2(o[2 * h + 1][0], o[2 * h + 1][1]);
        v.w = h16<T16>::pack2(o[2 * h + 1][2], o[2 * h + 1][3]);
        *reinterpret_cast<uint4*>(p + 8 * h) = v;
        store_b128_guard();      // (MFMAs of the next block follow: see common.hpp)
    }
}

// register fragment of a [query][key] quantity held as x[qb][kb][r]: k-slots (g, e) = keys 4g+e | 16+4g+(e-4)
template <typename T16>
__device__ __forceinline__ bf16x8_t frag_regs(const f32x4_t (&x)[2]) {
    const uint4 v = make_uint4(h16<T16>::pack2(x[0][0], x[0][1]), h16<T16>::pack2(x[0][2], x[0][3]), h16<T16>::pack2(x[1][0], x[1][1]), h16<T16>::pack2(x[1][2], x[1][3]));
    return __builtin_bit_cast(bf16x8_t, v);
}

// the lane's 8 key-padding flags (keys kb*16 + 4g + r), zero past T
__device__ __forceinline__ void load_keep(const AttnMArgs& a, const float* keep_row, float (&keep)[2][4]) {
    const int g = threadIdx.x >> 4;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // branch-free (clamped index, then select): a guarded load becomes its own basic block with an immediate
            // s_waitcnt vmcnt(0) -- eight serialised memory round trips per wavefront
            const int j = kb * 16 + 4 * g + r;
            const float v = keep_row[min(j, a.T - 1)];
            keep[kb][r] = (j < a.T) ? v : 0.f;
        }
}

// masked, scaled softmax (+ optional dropout mask out) of the lane's scores s[qb][kb][r]
__device__ __forceinline__ void softmax_regs(f32x4_t (&s)[2][2], const AttnMArgs& a, const float (&keep)[2][4]) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int i = qb * 16 + c;
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
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
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (kb * 16 + 4 * g + r < a.T) ? expf(s[qb][kb][r] - m) : 0.f;
                s[qb][kb][r] = e;
                sum += e;
            }
        sum = rows4_sum(sum);
        const float inv = (i < a.T) ? 1.0f / sum : 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qb][kb][r] *= inv;
    }
}

__device__ __forceinline__ void drop_mask_regs(const DropRng& d, uint64_t tile, float (&m)[2][2][4]) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
        {
            bool kp[4];
            drop_keep_vec<4>(d, (tile * 32 + (uint64_t)(qb * 16 + c)) * 32 + (uint64_t)(kb * 16 + 4 * g), kp);   // even start
#pragma unroll
            for (int r = 0; r < 4; ++r) m[qb][kb][r] = kp[r] ? d.inv_keep : 0.f;
        }
}

// ctx[query][d0 .. d0 + 16 NB) = P_d V for both query blocks; every lane writes 8 NB contiguous bytes of its row
template <typename T16, int NB>
__device__ __forceinline__ void fwd_pv(const AttnMArgs& a, const char* sV, const bf16x8_t (&pf)[2], size_t row0, int H, int col0) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    bf16x8_t vf[NB];
#pragma unroll
    for (int db = 0; db < NB; ++db) vf[db] = frag_tr_spread(sV, PITCH, 4 * db, 4 * NB);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        f32x4_t o[NB];
#pragma unroll
        for (int db = 0; db < NB; ++db) o[db] = h16<T16>::mma16(vf[db], pf[qb], zero);
        const int q = qb * 16 + c;
        if (q < a.T) store_row_d<T16, NB>(a.ctx, row0 + q, H, col0 + 4 * NB * g, o);
    }
}

// sum over the 16 lanes of a DPP row (every lane ends up with the total): quad butterflies, then the two mirrors
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

// dQ = dS K, dK = dS^T Q, dV = P_d^T dO for one head-width chunk of 16 NB columns (same d-contiguous output layout)
template <typename T16, int NB>
__device__ __forceinline__ void bwd_products(const AttnMArgs& a, const char* sQ, const char* sK, const char* sO,
                                             const bf16x8_t (&dsf)[2], const bf16x8_t (&dsT)[2], const bf16x8_t (&pT)[2],
                                             size_t row0, int pitch, int H, int col0, int seq) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t cq[NB], ck[NB], cv[NB];
#pragma unroll
    for (int db = 0; db < NB; ++db) { cq[db] = zero; ck[db] = zero; cv[db] = zero; }
    bf16x8_t kt[NB], qt[NB], ot[NB];
#pragma unroll
    for (int db = 0; db < NB; ++db) {
        kt[db] = frag_tr_spread(sK, PITCH, 4 * db, 4 * NB);   // K [key slots][d]
        qt[db] = frag_tr_spread(sQ, PITCH, 4 * db, 4 * NB);   // Q [query slots][d]
        ot[db] = frag_tr_spread(sO, PITCH, 4 * db, 4 * NB);   // dO [query slots][d]
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int r = b * 16 + c;
        f32x4_t dq[NB], dk[NB], dv[NB];
#pragma unroll
        for (int db = 0; db < NB; ++db) {
            // dQ[query r][d] = sum_key dS[r][key] K[key][d]
            dq[db] = h16<T16>::mma16(kt[db], dsf[b], zero);
            // dK[key r][d] = sum_query dS[query][r] Q[query][d];  dV[key r][d] = sum_query P_d[query][r] dO[query][d]
            dk[db] = h16<T16>::mma16(qt[db], dsT[b], zero);
            dv[db] = h16<T16>::mma16(ot[db], pT[b], zero);
        }
        if (r < a.T) {
            const int dcol = col0 + 4 * NB * g;
            store_row_d<T16, NB>(a.dqkv, row0 + r, pitch, dcol, dq);
            store_row_d<T16, NB>(a.dqkv, row0 + r, pitch, H + dcol, dk);
            store_row_d<T16, NB>(a.dqkv, row0 + r, pitch, 2 * H + dcol, dv);
            if (a.csum) {
#pragma unroll
                for (int db = 0; db < NB; ++db)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {      // fp32 sums of the rows BEFORE their bf16 rounding (three VALU ops per value cheaper)
                        cq[db][e] += dq[db][e];
                        ck[db][e] += dk[db][e];
                        cv[db][e] += dv[db][e];
                    }
            }
        }
    }
    if (a.csum) {   // bias gradient of the fused projection: this (sequence, head)'s column sums, rows = the 16 lanes of a DPP row
#pragma unroll
        for (int db = 0; db < NB; ++db)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cq[db][e] = row16_sum(cq[db][e]);
                ck[db][e] = row16_sum(ck[db][e]);
                cv[db][e] = row16_sum(cv[db][e]);
            }
        if (c == 0) {
            float* w = a.csum + (size_t)seq * 3 * H + col0 + 4 * NB * g;
#pragma unroll
            for (int db = 0; db < NB; ++db) {
                *reinterpret_cast<float4*>(w + 4 * db) = make_float4(cq[db][0], cq[db][1], cq[db][2], cq[db][3]);
                *reinterpret_cast<float4*>(w + H + 4 * db) = make_float4(ck[db][0], ck[db][1], ck[db][2], ck[db][3]);
                *reinterpret_cast<float4*>(w + 2 * H + 4 * db) = make_float4(cv[db][0], cv[db][1], cv[db][2], cv[db][3]);
            }
        }
    }
}

template <typename T16>
__global__ __launch_bounds__(64) void attn_fwd_mfma_kernel(AttnMArgs a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {      // spare rows of a bucket-padded packed layout: ctx = 0 there
        zero_dead_rows(a.ctx, a.cu, a.n_seq, a.total_rows, (size_t)a.n_heads * a.dh * 2, (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    __shared__ __attribute__((aligned(16))) char sV[TILE];
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int seq = blockIdx.x / a.n_heads, head = blockIdx.x % a.n_heads;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const size_t row0 = a.cu ? (size_t)a.cu[seq] : (size_t)seq * a.T;
    if (a.cu) a.T = a.cu[seq + 1] - a.cu[seq];   // unpadded layout: this sequence's own length
    if (a.T <= 0) return;                         // nothing to read or write (and the clamped loads need a valid last row)
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t s[2][2] = {{zero, zero}, {zero, zero}};
    const int nch = (a.dh + DCH - 1) / DCH;
    float keep[2][4];
    load_keep(a, a.key_keep + row0, keep);
    if (nch == 1) {
        // ONE memory round trip per wavefront: the V tile, the Q / K fragments of both k-steps and the padding flags are all
        // requested before the first of them is used (a missing second k-step, dh = 32, re-reads the first and is not multiplied)
        uint4 pv[4];
        load_tile_regs(a.qkv, row0, pitch, 2 * H + head * a.dh, a.dh, a.T, pv);
        bf16x8_t qf[2][2], kf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bool ok = ks * 32 < a.dh;
            const int dcol = head * a.dh + (ok ? ks * 32 : 0);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                qf[ks][b] = frag_nt_global(a.qkv, row0, pitch, dcol, b, a.T);
                kf[ks][b] = frag_nt_global(a.qkv, row0, pitch, H + dcol, b, a.T);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the whole load group ahead of the first use (the scheduler otherwise sinks the V loads)
        write_tile_lds(sV, pv);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks * 32 < a.dh) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) s[qb][kb] = h16<T16>::mma16(kf[ks][kb], qf[ks][qb], s[qb][kb]);
            }
        }
    } else {
        for (int d = 0; d < a.dh; d += 32) {     // one MFMA k-step per 32 head columns, operands straight from global memory
            const bf16x8_t qf[2] = {frag_nt_global(a.qkv, row0, pitch, head * a.dh + d, 0, a.T), frag_nt_global(a.qkv, row0, pitch, head * a.dh + d, 1, a.T)};
            const bf16x8_t kf[2] = {frag_nt_global(a.qkv, row0, pitch, H + head * a.dh + d, 0, a.T), frag_nt_global(a.qkv, row0, pitch, H + head * a.dh + d, 1, a.T)};
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) s[qb][kb] = h16<T16>::mma16(kf[kb], qf[qb], s[qb][kb]);
        }
    }
    __syncthreads();
    softmax_regs(s, a, keep);
    if (a.drop.thresh) {
        float m[2][2][4];
        drop_mask_regs(a.drop, blockIdx.x, m);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[qb][kb][r] *= m[qb][kb][r];
    }
    const bf16x8_t pf[2] = {frag_regs<T16>(s[0]), frag_regs<T16>(s[1])};
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
__global__ __launch_bounds__(64) void attn_bwd_mfma_kernel(AttnMArgs a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {      // spare rows: dqkv = 0 there
        zero_dead_rows(a.dqkv, a.cu, a.n_seq, a.total_rows, (size_t)3 * a.n_heads * a.dh * 2, (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    __shared__ __attribute__((aligned(16))) char sQ[TILE];
    __shared__ __attribute__((aligned(16))) char sK[TILE];
    __shared__ __attribute__((aligned(16))) char sO[TILE];
    __shared__ __attribute__((aligned(16))) char sP[32 * PP];    // dropped probabilities  [query][key] bf16
    __shared__ __attribute__((aligned(16))) char sS[32 * PP];    // dS                     [query][key] bf16
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int seq = blockIdx.x / a.n_heads, head = blockIdx.x % a.n_heads;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const size_t row0 = a.cu ? (size_t)a.cu[seq] : (size_t)seq * a.T;
    if (a.cu) a.T = a.cu[seq + 1] - a.cu[seq];   // unpadded layout: this sequence's own length
    if (a.T <= 0) {                               // nothing to read or write (and the clamped loads need a valid last row)
        if (a.csum)
            for (int j = lane; j < 3 * a.dh; j += 64) a.csum[(size_t)seq * 3 * H + (j / a.dh) * H + head * a.dh + j % a.dh] = 0.f;
        return;
    }
    const bf16* dctx = a.ctx;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t s[2][2] = {{zero, zero}, {zero, zero}}, dp[2][2] = {{zero, zero}, {zero, zero}};
    const int nch = (a.dh + DCH - 1) / DCH;
    float keep[2][4];
    load_keep(a, a.key_keep + row0, keep);
    if (nch == 1) {
        // ONE memory round trip: the Q / K / dO tiles, the V fragments of both k-steps and the padding flags are requested together
        uint4 pq[4], pk[4], po[4];
        load_tile_regs(a.qkv, row0, pitch, head * a.dh, a.dh, a.T, pq);
        load_tile_regs(a.qkv, row0, pitch, H + head * a.dh, a.dh, a.T, pk);
        load_tile_regs(dctx, row0, H, head * a.dh, a.dh, a.T, po);
        bf16x8_t vf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bool ok = ks * 32 < a.dh;
            const int vcol = 2 * H + head * a.dh + (ok ? ks * 32 : 0);
#pragma unroll
            for (int b = 0; b < 2; ++b) vf[ks][b] = frag_nt_global(a.qkv, row0, pitch, vcol, b, a.T);
        }
        __builtin_amdgcn_sched_barrier(0);
        write_tile_lds(sQ, pq);
        write_tile_lds(sK, pk);
        write_tile_lds(sO, po);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks * 32 < a.dh) {
                const bf16x8_t qf[2] = {frag_nt(sQ, 0, ks), frag_nt(sQ, 1, ks)};
                const bf16x8_t kf[2] = {frag_nt(sK, 0, ks), frag_nt(sK, 1, ks)};
                const bf16x8_t of[2] = {frag_nt(sO, 0, ks), frag_nt(sO, 1, ks)};
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        s[qb][kb] = h16<T16>::mma16(kf[kb], qf[qb], s[qb][kb]);
                        dp[qb][kb] = h16<T16>::mma16(vf[ks][kb], of[qb], dp[qb][kb]);
                    }
            }
        }
    }
    for (int ch = 0; nch > 1 && ch < nch; ++ch) {
        const int d0 = ch * DCH, nc = min(DCH, a.dh - d0);
        stage_tile(a.qkv, row0, pitch, head * a.dh + d0, nc, a.T, sQ);
        stage_tile(a.qkv, row0, pitch, H + head * a.dh + d0, nc, a.T, sK);
        stage_tile(dctx, row0, H, head * a.dh + d0, nc, a.T, sO);
        __syncthreads();
        for (int ks = 0; ks < nc / 32; ++ks) {
            bf16x8_t qf[2] = {frag_nt(sQ, 0, ks), frag_nt(sQ, 1, ks)};
            bf16x8_t kf[2] = {frag_nt(sK, 0, ks), frag_nt(sK, 1, ks)};
            const int vcol = 2 * H + head * a.dh + d0 + ks * 32;     // V is only consumed d-contiguous: no LDS tile
            bf16x8_t vf[2] = {frag_nt_global(a.qkv, row0, pitch, vcol, 0, a.T), frag_nt_global(a.qkv, row0, pitch, vcol, 1, a.T)};
            bf16x8_t of[2] = {frag_nt(sO, 0, ks), frag_nt(sO, 1, ks)};
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    s[qb][kb] = h16<T16>::mma16(kf[kb], qf[qb], s[qb][kb]);
                    dp[qb][kb] = h16<T16>::mma16(vf[kb], of[qb], dp[qb][kb]);
                }
        }
        if (nch > 1) __syncthreads();
    }
    softmax_regs(s, a, keep);
    float msk[2][2][4];
    if (a.drop.thresh) drop_mask_regs(a.drop, blockIdx.x, msk);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (a.drop.thresh) dp[qb][kb][r] *= msk[qb][kb][r];     // dP = dP_dropped o mask / (1 - p)
                delta += s[qb][kb][r] * dp[qb][kb][r];
            }
        delta = rows4_sum(delta);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
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
    const bf16x8_t dsf[2] = {frag_regs<T16>(dp[0]), frag_regs<T16>(dp[1])};
    __syncthreads();
    // transposed [key][query-slot] fragments of dS and P for the two key blocks
    const bf16x8_t dsT[2] = {frag_tr(sS, PP, 0), frag_tr(sS, PP, 16)};
    const bf16x8_t pT[2] = {frag_tr(sP, PP, 0), frag_tr(sP, PP, 16)};
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * DCH, nc = min(DCH, a.dh - d0);
        if (nch > 1) {
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
int morec_attn_mfma64_launch(const morec_attn_desc* d, const void* qkv, const float* key_keep, void* ctx_or_dctx, void* dqkv, bool backward,
                             hipStream_t s, float* csum);      // attention_mfma64.hip: 32 < T <= 64
// returns MOREC_E_UNSUPPORTED when the shape is outside this fast path (caller falls back to attention.hip)
int morec_attn_mfma_launch(const morec_attn_desc* d, const void* qkv, const float* key_keep, void* ctx_or_dctx,
                           void* dqkv, bool backward, hipStream_t s, float* csum) {
    if (is_h16(d->dtype) && d->dh % 32 == 0 && d->T > 32 && d->T <= 64) {
        static const bool off = [] { const char* e = getenv("MOREC_ATTN_MFMA64"); return e && e[0] == '0'; }();      // MOREC_ATTN_MFMA64=0: the VALU kernels (A/B)
        if (!off) return morec_attn_mfma64_launch(d, qkv, key_keep, ctx_or_dctx, dqkv, backward, s, csum);
    }
    if (!is_h16(d->dtype) || d->dh % 32 != 0 || d->T > 32) return MOREC_E_UNSUPPORTED;
    AttnMArgs a{reinterpret_cast<const bf16*>(qkv), key_keep, reinterpret_cast<bf16*>(ctx_or_dctx),
                reinterpret_cast<bf16*>(dqkv), csum, d->n_seq, d->T, d->n_heads, d->dh, d->causal, d->scale, d->mask_value,
                make_drop(d->p_drop, d->seed), d->cu_seqlens, d->total_rows};
    dim3 grid(d->n_seq * d->n_heads + attn_spare_blocks(d)), block(64);
    by_h16(d->dtype, [&](auto* t) {
        using T = MOREC_TAG_T(t);
        if (backward)
            hipLaunchKernelGGL(attn_bwd_mfma_kernel<T>, grid, block, 0, s, a);
        else
            hipLaunchKernelGGL(attn_fwd_mfma_kernel<T>, grid, block, 0, s, a);
    });
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
