// attention_f32mfma.hip -- fp32 small-tile attention (T <= 32, head width a multiple of 16) on the matrix cores: the attention of the
// exact-fp32 parity mode and of the fp32x3 mode (fp32 tensors; compute_dtype "fp32" / "fp32x3").  Same contract as attention.hip
// (one wavefront per (sequence, head), nothing kept from the forward for the backward, counter-based dropout on the probabilities,
// the masks of T/model/encoders.py:24-27 / HF eager attention as additive values) -- what changes is where the arithmetic runs: the
// VALU kernels of attention.hip spend 0.6 / 1.15 ms per BERT-base layer (55 k tokens) on 4 x 4 register blocks, 21 ms of the
// fp32x3 step; here every product is v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation).
//
// Register layout ("A"): a [query][key] quantity X lives as  x[kb][qb][r] = X[q = qb*16 + c][key = kb*16 + 4g + r],  c = lane & 15,
// g = lane >> 4 -- the MFMA's output layout with the KEY side as its M operand.  A softmax row is then 4 lanes (g) x 8 registers:
// two xor-shuffles per reduction.  The layout is also exactly what the second products want as their B operand
// (B[k = key][n = query], lane (n = c, k = g)), so P V and dS K take P / dS straight from registers; dS^T Q and P^T dO contract
// over the QUERY index and read dS / P back from an LDS tile ([q][key], pitch 36 floats: conflict-free b32 reads).
// A 16-byte fragment read serves four MFMAs: the e-th takes element e of every lane, i.e. d = 4g + e -- a permutation of the
// contraction index applied to both operands alike.
// Output blocks come out as  o[qb][db][r] = O[row = qb*16 + c][d = db*16 + 4g + r]: 16-byte stores along a row.
#include <stdlib.h>
#include "common.hpp"

namespace {
constexpr int DC = 64;                 // head-width chunk staged per pass (floats)
constexpr int P = DC + 4;              // pitch of the [32 x DC] operand tiles: rows 4 banks apart, b128 and b32 reads conflict-free
constexpr int PP = 36;                 // pitch of the [32 x 32] dS / P tiles
constexpr int TILE = 32 * P;           // floats
constexpr int PTILE = 32 * PP;

struct A32Args {
    const float* qkv;
    const float* key_keep;
    float* ctx;          // fwd: output; bwd: dctx (read)
    float* dqkv;
    int n_seq, T, n_heads, dh, causal;
    float scale, mask_value;
    DropRng drop;
    const int32_t* cu;
    int total_rows;
};

// rows [0, T) x columns [col, col + ncols) of a row-major matrix -> registers -> [32][P] LDS tile, zero beyond.  Two halves: a kernel phase
// issues the loads of ALL its tiles (from clamped addresses: no guarded load, no basic block per load) before the first LDS write -- interleaved
// load / ds_write pairs stay in program order (the LDS pointer may alias the source as far as the compiler knows) and every tile would cost
// its own memory round trip.  The zeroing is a select on the way into LDS.
__device__ __forceinline__ void load32(const float* __restrict__ src, size_t row0, int pitch, int col, int ncols, int Tlen, float4 (&v)[8]) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = lane + 64 * i, r = s >> 4, c4 = (s & 15) * 4;
        v[i] = *reinterpret_cast<const float4*>(src + (row0 + min(r, Tlen - 1)) * (size_t)pitch + col + (c4 < ncols ? c4 : 0));
    }
}
__device__ __forceinline__ void write32(float* __restrict__ dst, int ncols, int Tlen, const float4 (&v)[8]) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = lane + 64 * i, r = s >> 4, c4 = (s & 15) * 4;
        const bool ok = r < Tlen && c4 < ncols;
        *reinterpret_cast<float4*>(dst + r * P + c4) = ok ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ float4 frag(const float* tile, int blk, int d0) {
    const int lane = threadIdx.x;
    return *reinterpret_cast<const float4*>(tile + (blk * 16 + (lane & 15)) * P + d0 + 4 * (lane >> 4));
}

__device__ __forceinline__ f32x4_t mma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// x[kb][qb] += sum_d M[kb*16 + .][d] * N[qb*16 + .][d] over `nd` columns of the staged tiles (layout A: M = the key-side operand)
__device__ __forceinline__ void dot_nt(const float* tm, const float* tn, int nd, f32x4_t (&x)[2][2]) {
    for (int d0 = 0; d0 < nd; d0 += 16) {
        const float4 fm[2] = {frag(tm, 0, d0), frag(tm, 1, d0)};
        const float4 fn[2] = {frag(tn, 0, d0), frag(tn, 1, d0)};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                x[kb][qb] = mma4(fm[kb].x, fn[qb].x, x[kb][qb]);
                x[kb][qb] = mma4(fm[kb].y, fn[qb].y, x[kb][qb]);
                x[kb][qb] = mma4(fm[kb].z, fn[qb].z, x[kb][qb]);
                x[kb][qb] = mma4(fm[kb].w, fn[qb].w, x[kb][qb]);
            }
    }
}

// o[qb][db] = sum_key W[q][key] * V[key][db*16 ..]: W in layout-A registers, V rows from the staged tile (ndb 16-column blocks)
__device__ __forceinline__ void pv_regs(const f32x4_t (&w)[2][2], const float* tv, int ndb, f32x4_t (&o)[2][4]) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        if (db < ndb) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = tv[(kb * 16 + 4 * g + e) * P + db * 16 + c];
                    o[0][db] = mma4(a, w[kb][0][e], o[0][db]);
                    o[1][db] = mma4(a, w[kb][1][e], o[1][db]);
                }
        }
    }
}

// o[kb][db] = sum_q W[q][kb*16 ..] * X[q][db*16 ..]: W from its [q][key] LDS tile, X rows from the staged tile
__device__ __forceinline__ void pv_lds(const float* tw, const float* tx, int ndb, f32x4_t (&o)[2][4]) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    float b[2][2][4];      // [qb][kb][e] = W[qb*16 + 4g + e][kb*16 + c]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[qb][kb][e] = tw[(qb * 16 + 4 * g + e) * PP + kb * 16 + c];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        if (db < ndb) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = tx[(qb * 16 + 4 * g + e) * P + db * 16 + c];
                    o[0][db] = mma4(a, b[qb][0][e], o[0][db]);
                    o[1][db] = mma4(a, b[qb][1][e], o[1][db]);
                }
        }
    }
}

__device__ __forceinline__ void zero_o(f32x4_t (&o)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// rows rb*16 + c of the output, columns col + db*16 + 4g .. +3
__device__ __forceinline__ void store_o(float* __restrict__ dst, size_t row0, int pitch, int col, int ndb, int Tlen, const f32x4_t (&o)[2][4]) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int r = rb * 16 + c;
        if (r < Tlen) {
            float* p = dst + (row0 + r) * (size_t)pitch + col + 4 * g;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                if (db < ndb) {
                    *reinterpret_cast<float4*>(p + db * 16) = make_float4(o[rb][db][0], o[rb][db][1], o[rb][db][2], o[rb][db][3]);
                    store_b128_guard();      // (the accumulators are rewritten by the next product's MFMAs: see common.hpp)
                }
        }
    }
}

// masked, scaled softmax in layout A (reference arithmetic: score * scale + additive mask; keys >= T never enter); returns with s = P
__device__ __forceinline__ void softmax_a(f32x4_t (&s)[2][2], int Tlen, int causal, float scale, float mask_value, const float* __restrict__ keep_row) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    float keep[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = kb * 16 + 4 * g + r;
            const float k = keep_row[min(j, Tlen - 1)];
            keep[kb][r] = j < Tlen ? k : 0.f;
        }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int i = qb * 16 + c;
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kb * 16 + 4 * g + r;
                const bool kept = (keep[kb][r] != 0.f) && (!causal || j <= i);
                const float v = s[kb][qb][r] * scale + (kept ? 0.f : mask_value);
                s[kb][qb][r] = (j < Tlen) ? v : -INFINITY;
                m = fmaxf(m, s[kb][qb][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kb * 16 + 4 * g + r;
                const float e = (j < Tlen) ? expf(s[kb][qb][r] - m) : 0.f;
                s[kb][qb][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = (i < Tlen) ? 1.0f / sum : 0.f;      // padded query rows contribute nothing downstream
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][qb][r] *= inv;
    }
}

// dropout keep-mask (x 1 / (1 - p)) in layout A: element index ((tile * 32 + i) * 32 + j), the stream of attention.hip / attention_mfma.hip
__device__ __forceinline__ void drop_mask_a(const DropRng& d, uint64_t tile, float (&m)[2][2][4]) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            bool kp[4];
            drop_keep_vec<4>(d, (tile * 32 + (uint64_t)(qb * 16 + c)) * 32 + (uint64_t)(kb * 16 + 4 * g), kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) m[kb][qb][r] = kp[r] ? d.inv_keep : 0.f;
        }
}

__global__ __launch_bounds__(64) void attn_fwd_f32mfma_kernel(A32Args a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {      // spare rows of a bucket-padded packed layout: ctx = 0 there
        zero_dead_rows(a.ctx, a.cu, a.n_seq, a.total_rows, (size_t)a.n_heads * a.dh * sizeof(float), (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    const int seq_ = blockIdx.x / a.n_heads, head = blockIdx.x % a.n_heads;
    const size_t row0 = a.cu ? (size_t)a.cu[seq_] : (size_t)seq_ * a.T;
    if (a.cu) a.T = a.cu[seq_ + 1] - a.cu[seq_];
    if (a.T <= 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* sA = smem_f;
    float* sB = sA + TILE;
    const int H = a.n_heads * a.dh, pitch = 3 * H;

    f32x4_t s[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float4 vv[8];      // the first V chunk: requested as soon as the last Q / K chunk is in LDS, consumed behind the softmax
    for (int d0 = 0; d0 < a.dh; d0 += DC) {
        const int nd = min(DC, a.dh - d0);
        float4 vq[8], vk[8];
        load32(a.qkv, row0, pitch, head * a.dh + d0, nd, a.T, vq);            // Q
        load32(a.qkv, row0, pitch, H + head * a.dh + d0, nd, a.T, vk);        // K
        write32(sA, nd, a.T, vq);
        write32(sB, nd, a.T, vk);
        __syncthreads();
        if (d0 + DC >= a.dh) load32(a.qkv, row0, pitch, 2 * H + head * a.dh, min(DC, a.dh), a.T, vv);
        dot_nt(sB, sA, nd, s);
        __syncthreads();
    }
    softmax_a(s, a.T, a.causal, a.scale, a.mask_value, a.key_keep + row0);
    if (a.drop.thresh) {
        float m[2][2][4];
        drop_mask_a(a.drop, blockIdx.x, m);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kb][qb][r] *= m[kb][qb][r];
    }
    for (int d0 = 0; d0 < a.dh; d0 += DC) {
        const int nd = min(DC, a.dh - d0);
        if (d0 > 0) load32(a.qkv, row0, pitch, 2 * H + head * a.dh + d0, nd, a.T, vv);
        write32(sA, nd, a.T, vv);                                             // V
        __syncthreads();
        f32x4_t o[2][4];
        zero_o(o);
        pv_regs(s, sA, nd / 16, o);
        store_o(a.ctx, row0, H, head * a.dh + d0, nd / 16, a.T, o);
        __syncthreads();
    }
}

// Backward: recomputes P from Q, K, then  dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(P o dP)) * scale, dQ = dS K, dK = dS^T Q.
__global__ __launch_bounds__(64) void attn_bwd_f32mfma_kernel(A32Args a) {
    if ((int)blockIdx.x >= a.n_seq * a.n_heads) {      // spare rows: dqkv = 0 there
        zero_dead_rows(a.dqkv, a.cu, a.n_seq, a.total_rows, (size_t)3 * a.n_heads * a.dh * sizeof(float), (int)blockIdx.x - a.n_seq * a.n_heads);
        return;
    }
    a.drop = drop_resolve(a.drop);
    const int seq_ = blockIdx.x / a.n_heads, head = blockIdx.x % a.n_heads;
    const size_t row0 = a.cu ? (size_t)a.cu[seq_] : (size_t)seq_ * a.T;
    if (a.cu) a.T = a.cu[seq_ + 1] - a.cu[seq_];
    if (a.T <= 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* sQ = smem_f;
    float* sK = sQ + TILE;
    float* sO = sK + TILE;
    float* sV = sO + TILE;           // pass 1 only; the dS / P tiles of pass 2 take its place
    float* sS = sV;                  // dS [q][key]
    float* sP = sS + PTILE;          // P (dropped) [q][key]
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int H = a.n_heads * a.dh, pitch = 3 * H;
    const float* dctx = a.ctx;
    const bool one_chunk = a.dh <= DC;

    f32x4_t s[2][2], dp[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { s[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int d0 = 0; d0 < a.dh; d0 += DC) {
        const int nd = min(DC, a.dh - d0);
        {   // (LDS caps this kernel at one wavefront per SIMD: the 128 staging registers are free)
            float4 vq[8], vk[8], vv[8], vo[8];
            load32(a.qkv, row0, pitch, head * a.dh + d0, nd, a.T, vq);
            load32(a.qkv, row0, pitch, H + head * a.dh + d0, nd, a.T, vk);
            load32(a.qkv, row0, pitch, 2 * H + head * a.dh + d0, nd, a.T, vv);
            load32(dctx, row0, H, head * a.dh + d0, nd, a.T, vo);
            write32(sQ, nd, a.T, vq);
            write32(sK, nd, a.T, vk);
            write32(sV, nd, a.T, vv);
            write32(sO, nd, a.T, vo);
        }
        __syncthreads();
        dot_nt(sK, sQ, nd, s);
        dot_nt(sV, sO, nd, dp);
        __syncthreads();
    }
    softmax_a(s, a.T, a.causal, a.scale, a.mask_value, a.key_keep + row0);
    float msk[2][2][4];
    if (a.drop.thresh) {
        drop_mask_a(a.drop, blockIdx.x, msk);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dp[kb][qb][r] *= msk[kb][qb][r];      // dP = dP_dropped o mask / (1 - p)
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) delta += s[kb][qb][r] * dp[kb][qb][r];
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[kb][qb][r] = s[kb][qb][r] * (dp[kb][qb][r] - delta) * a.scale;      // dp now holds dS
            if (a.drop.thresh) {      // dV uses the DROPPED probabilities
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kb][qb][r] *= msk[kb][qb][r];
            }
            const int q = qb * 16 + c, j = kb * 16 + 4 * g;
            *reinterpret_cast<float4*>(sS + q * PP + j) = make_float4(dp[kb][qb][0], dp[kb][qb][1], dp[kb][qb][2], dp[kb][qb][3]);
            *reinterpret_cast<float4*>(sP + q * PP + j) = make_float4(s[kb][qb][0], s[kb][qb][1], s[kb][qb][2], s[kb][qb][3]);
        }
    }
    __syncthreads();
    for (int d0 = 0; d0 < a.dh; d0 += DC) {
        const int nd = min(DC, a.dh - d0), ndb = nd / 16;
        if (!one_chunk) {      // (a single chunk is still staged from pass 1)
            float4 vq[8], vk[8], vo[8];
            load32(a.qkv, row0, pitch, head * a.dh + d0, nd, a.T, vq);
            load32(a.qkv, row0, pitch, H + head * a.dh + d0, nd, a.T, vk);
            load32(dctx, row0, H, head * a.dh + d0, nd, a.T, vo);
            write32(sQ, nd, a.T, vq);
            write32(sK, nd, a.T, vk);
            write32(sO, nd, a.T, vo);
            __syncthreads();
        }
        f32x4_t o[2][4];
        zero_o(o);
        pv_regs(dp, sK, ndb, o);                 // dQ = dS K
        store_o(a.dqkv, row0, pitch, head * a.dh + d0, ndb, a.T, o);
        zero_o(o);
        pv_lds(sS, sQ, ndb, o);                  // dK = dS^T Q
        store_o(a.dqkv, row0, pitch, H + head * a.dh + d0, ndb, a.T, o);
        zero_o(o);
        pv_lds(sP, sO, ndb, o);                  // dV = P^T dO
        store_o(a.dqkv, row0, pitch, 2 * H + head * a.dh + d0, ndb, a.T, o);
        if (!one_chunk) __syncthreads();
    }
}

constexpr int LDS_FWD = 2 * TILE * (int)sizeof(float);
constexpr int LDS_BWD = (3 * TILE + (2 * PTILE > TILE ? 2 * PTILE : TILE)) * (int)sizeof(float);
}  // namespace

int attn_spare_blocks(const morec_attn_desc* d);      // attention.hip

// fp32 tensors, T <= 32, head width a multiple of 16: MOREC_OK / an error, or MOREC_E_UNSUPPORTED (shape outside this path: the VALU
// kernels of attention.hip take it).  MOREC_ATTN_F32_MFMA=0 switches the path off (A/B).
int morec_attn_f32mfma_launch(const morec_attn_desc* d, const void* qkv, const float* key_keep, void* ctx_or_dctx, void* dqkv, bool backward,
                              hipStream_t s) {
    static const bool off = [] { const char* e = getenv("MOREC_ATTN_F32_MFMA"); return e && e[0] == '0'; }();
    if (off || d->dtype != MOREC_F32 || d->T > 32 || d->dh % 16 != 0) return MOREC_E_UNSUPPORTED;
    const int H = d->n_heads * d->dh;
    if (!aligned16(qkv) || !aligned16(ctx_or_dctx) || (dqkv && !aligned16(dqkv)) || H % 4) return MOREC_E_UNSUPPORTED;
    A32Args a{reinterpret_cast<const float*>(qkv), key_keep, reinterpret_cast<float*>(ctx_or_dctx), reinterpret_cast<float*>(dqkv),
              d->n_seq, d->T, d->n_heads, d->dh, d->causal, d->scale, d->mask_value, make_drop(d->p_drop, d->seed), d->cu_seqlens, d->total_rows};
    dim3 grid(d->n_seq * d->n_heads + attn_spare_blocks(d)), block(64);
    if (backward)
        hipLaunchKernelGGL(attn_bwd_f32mfma_kernel, grid, block, LDS_BWD, s, a);
    else
        hipLaunchKernelGGL(attn_fwd_f32mfma_kernel, grid, block, LDS_FWD, s, a);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
