// swin.hip -- the Swin-specific kernels of the vision item tower (V/model/encoders.py:24-31 over HF
// SwinForImageClassification, transformers/models/swin/modeling_swin.py):
//   * patchify         4x4/4 patch-embedding conv as im2col rows for the MFMA GEMM          (:247-286)
//   * window attention 7x7 windows; window partition, cyclic shift and window reverse are
//                      folded into the row addressing (no permuted copies), relative-position
//                      bias + -100 shift-region mask, fp32 softmax, PV                       (:401-468, :486-505, :584-607)
//   * patch merging    2x2 strided gather-concat (and its inverse for the backward pass)     (:309-326)
//   * mean pool over the last stage's tokens (AdaptiveAvgPool1d(1), :876-879) and DropPath scales (:42-60)
// The token-wise work (LayerNorms, q/k/v/o, MLP) runs on the shared LayerNorm / GEMM kernels.
#include <algorithm>
#include "common.hpp"

namespace {
constexpr int DH = 32;   // head width of every Swin variant (C / heads = 96/3 = 128/4 = 192/6 = 32)

struct SwinAttnArgs {
    const void* qkv;      // [rows, 3C]  q | k | v, natural token order (n, y, x)
    const float* bias_t;  // [heads][j][i] relative-position bias, transposed so that lanes (= query i) read coalesced
    void* ctx;            // [rows, C]   forward: output; backward: the saved forward output
    const void* dctx;     // backward: gradient of ctx
    void* dqkv;           // backward: [rows, 3C]
    float* dbias_t;       // backward: [heads][j][i] accumulators (atomicAdd)
    float* dbias_part;    // deterministic mode: [heads][gridDim.x][NT * NT] per-block partials instead (folded in block order by the launcher)
    int n_img, H, W, shift, heads;
    float scale;
    int n_win_total, wpb; // windows in the launch, windows per block
};

// natural row index and cyclic-shift region of token t of window (wr, wc) of image img
template <int WS>
__device__ __forceinline__ void token_geom(const SwinAttnArgs& a, int img, int wr, int wc, int t, int& row, int& reg) {
    const int wy = t / WS, wx = t - wy * WS;
    const int ys = wr * WS + wy, xs = wc * WS + wx;          // coordinates in the shifted image
    int y = ys + a.shift, x = xs + a.shift;                  // roll(-shift): shifted[ys] = original[(ys + shift) % H]
    if (y >= a.H) y -= a.H;
    if (x >= a.W) x -= a.W;
    row = (img * a.H + y) * a.W + x;
    reg = 0;
    if (a.shift > 0) {
        const int hr = (ys >= a.H - WS) + (ys >= a.H - a.shift);
        const int wq = (xs >= a.W - WS) + (xs >= a.W - a.shift);
        reg = hr * 3 + wq;
    }
}

// stage the [NT x 32] K and V tiles (fp32) of one (window, head) through the row table
template <typename T, int NT>
__device__ __forceinline__ void stage_kv(const T* __restrict__ qkv, const int* __restrict__ sRow, int pitch, int C, int head,
                                         float* __restrict__ sK, float* __restrict__ sV) {
    constexpr int EV = vio<T>::EV, VPR = DH / EV;
    for (int v = threadIdx.x; v < NT * 2 * VPR; v += 64) {
        const int r = v / (2 * VPR), rem = v - r * 2 * VPR;
        const int which = rem / VPR, part = rem - which * VPR;
        float o[EV];
        vio<T>::load(qkv + (size_t)sRow[r] * pitch + (1 + which) * C + head * DH + part * EV, o);
        store_f32v<EV>((which ? sV : sK) + r * DH + part * EV, o);
    }
}

template <typename T>
__device__ __forceinline__ void load_row32(const T* __restrict__ p, float (&o)[DH]) {
    constexpr int EV = vio<T>::EV;
#pragma unroll
    for (int q = 0; q < DH / EV; ++q) {
        float t[EV];
        vio<T>::load(p + q * EV, t);
#pragma unroll
        for (int k = 0; k < EV; ++k) o[q * EV + k] = t[k];
    }
}
template <typename T>
__device__ __forceinline__ void store_row32(T* __restrict__ p, const float (&o)[DH]) {
    constexpr int EV = vio<T>::EV;
#pragma unroll
    for (int q = 0; q < DH / EV; ++q) {
        float t[EV];
#pragma unroll
        for (int k = 0; k < EV; ++k) t[k] = o[q * EV + k];
        vio<T>::store(p + q * EV, t);
    }
}

// scores of query row `lane` against all NT keys, softmaxed in place (every lane owns a full row: no cross-lane traffic)
template <int NT>
__device__ __forceinline__ void row_softmax(const float (&q)[DH], const float* __restrict__ sK, const float* __restrict__ sB,
                                            const int* __restrict__ sReg, int lane, int myreg, float scale, float (&s)[NT]) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4 k = *reinterpret_cast<const float4*>(sK + j * DH + d);   // same address in every lane: broadcast
            acc = fmaf(q[d], k.x, acc); acc = fmaf(q[d + 1], k.y, acc);
            acc = fmaf(q[d + 2], k.z, acc); acc = fmaf(q[d + 3], k.w, acc);
        }
        // reference arithmetic: q.k * scaling + (bias + shift mask)
        s[j] = acc * scale + (sB[j * NT + lane] + (sReg[j] != myreg ? -100.0f : 0.0f));
        m = fmaxf(m, s[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        s[j] = __expf(s[j] - m);
        sum += s[j];
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < NT; ++j) s[j] *= inv;
}

template <typename T, int WS>
__global__ __launch_bounds__(64) void swin_attn_fwd_kernel(SwinAttnArgs a) {
    constexpr int NT = WS * WS;
    __shared__ __attribute__((aligned(16))) float sK[NT * DH];
    __shared__ __attribute__((aligned(16))) float sV[NT * DH];
    __shared__ float sB[NT * NT];
    __shared__ int sRow[NT], sReg[NT];
    const int lane = threadIdx.x, head = blockIdx.y;
    const int C = a.heads * DH, pitch = 3 * C;
    const int nWx = a.W / WS, nW = (a.H / WS) * nWx;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    T* ctx = reinterpret_cast<T*>(a.ctx);
    for (int e = lane; e < NT * NT; e += 64) sB[e] = a.bias_t[(size_t)head * NT * NT + e];
    const int g0 = blockIdx.x * a.wpb, g1 = min(a.n_win_total, g0 + a.wpb);
    for (int g = g0; g < g1; ++g) {
        const int img = g / nW, wi = g - img * nW;
        const int wr = wi / nWx, wc = wi - wr * nWx;
        __syncthreads();   // the previous window's readers are done with sK / sV / sRow / sReg
        int myrow = 0, myreg = 0;
        if (lane < NT) {
            token_geom<WS>(a, img, wr, wc, lane, myrow, myreg);
            sRow[lane] = myrow;
            sReg[lane] = myreg;
        }
        __syncthreads();
        stage_kv<T, NT>(qkv, sRow, pitch, C, head, sK, sV);
        float q[DH];
        if (lane < NT) load_row32<T>(qkv + (size_t)myrow * pitch + head * DH, q);
        __syncthreads();
        if (lane < NT) {
            float s[NT];
            row_softmax<NT>(q, sK, sB, sReg, lane, myreg, a.scale, s);
            float o[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(sV + j * DH + d);
                    o[d] = fmaf(s[j], v.x, o[d]); o[d + 1] = fmaf(s[j], v.y, o[d + 1]);
                    o[d + 2] = fmaf(s[j], v.z, o[d + 2]); o[d + 3] = fmaf(s[j], v.w, o[d + 3]);
                }
            }
            store_row32<T>(ctx + (size_t)myrow * C + head * DH, o);   // window reverse + un-shift = same row the query came from
        }
    }
}

// Backward.  P is recomputed from Q, K; delta_i = sum_j P_ij dP_ij = dO_i . O_i with O the saved forward output, so dS_ij is
// formed as soon as dP_ij is and only ONE NT-long register array (P) is live.  Phase A (lane = query i): P, dS, dQ, dbias;
// P and dS go to LDS (pitch NT = 49, odd: conflict-free both row-wise and column-wise).  Phase B (lane = key j): the lane's
// own q_i / dO_i rows (already in registers) replace K / V in LDS, then dV_j = sum_i P_ij dO_i, dK_j = scale sum_i dS_ij Q_i.
template <typename T, int WS>
__global__ __launch_bounds__(64) void swin_attn_bwd_kernel(SwinAttnArgs a) {
    constexpr int NT = WS * WS;
    __shared__ __attribute__((aligned(16))) float sA[NT * DH];   // K, then Q
    __shared__ __attribute__((aligned(16))) float sBv[NT * DH];  // V, then dO
    __shared__ float sBias[NT * NT];
    __shared__ float sP[NT * NT];
    __shared__ float sS[NT * NT];
    __shared__ int sRow[NT], sReg[NT];
    const int lane = threadIdx.x, head = blockIdx.y;
    const int C = a.heads * DH, pitch = 3 * C;
    const int nWx = a.W / WS, nW = (a.H / WS) * nWx;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* ctx = reinterpret_cast<const T*>(a.ctx);
    const T* dctx = reinterpret_cast<const T*>(a.dctx);
    T* dqkv = reinterpret_cast<T*>(a.dqkv);
    for (int e = lane; e < NT * NT; e += 64) sBias[e] = a.bias_t[(size_t)head * NT * NT + e];
    float dbacc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) dbacc[j] = 0.f;
    const int g0 = blockIdx.x * a.wpb, g1 = min(a.n_win_total, g0 + a.wpb);
    for (int g = g0; g < g1; ++g) {
        const int img = g / nW, wi = g - img * nW;
        const int wr = wi / nWx, wc = wi - wr * nWx;
        __syncthreads();
        int myrow = 0, myreg = 0;
        if (lane < NT) {
            token_geom<WS>(a, img, wr, wc, lane, myrow, myreg);
            sRow[lane] = myrow;
            sReg[lane] = myreg;
        }
        __syncthreads();
        stage_kv<T, NT>(qkv, sRow, pitch, C, head, sA, sBv);
        float q[DH], dO[DH];
        float delta = 0.f;
        if (lane < NT) {
            load_row32<T>(qkv + (size_t)myrow * pitch + head * DH, q);
            load_row32<T>(dctx + (size_t)myrow * C + head * DH, dO);
            float o[DH];
            load_row32<T>(ctx + (size_t)myrow * C + head * DH, o);
#pragma unroll
            for (int d = 0; d < DH; ++d) delta = fmaf(dO[d], o[d], delta);
        }
        __syncthreads();
        if (lane < NT) {
            float s[NT];
            row_softmax<NT>(q, sA, sBias, sReg, lane, myreg, a.scale, s);
            float dq[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) dq[d] = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(sBv + j * DH + d);
                    dp = fmaf(dO[d], v.x, dp); dp = fmaf(dO[d + 1], v.y, dp);
                    dp = fmaf(dO[d + 2], v.z, dp); dp = fmaf(dO[d + 3], v.w, dp);
                }
                const float ds = s[j] * (dp - delta);     // gradient at the pre-softmax score (= gradient of the bias entry)
                dbacc[j] += ds;
                sP[lane * NT + j] = s[j];
                sS[lane * NT + j] = ds;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 k = *reinterpret_cast<const float4*>(sA + j * DH + d);
                    dq[d] = fmaf(ds, k.x, dq[d]); dq[d + 1] = fmaf(ds, k.y, dq[d + 1]);
                    dq[d + 2] = fmaf(ds, k.z, dq[d + 2]); dq[d + 3] = fmaf(ds, k.w, dq[d + 3]);
                }
            }
#pragma unroll
            for (int d = 0; d < DH; ++d) dq[d] *= a.scale;
            store_row32<T>(dqkv + (size_t)myrow * pitch + head * DH, dq);
        }
        __syncthreads();   // every lane is done with K / V
        if (lane < NT) {
            store_f32v<DH>(sA + lane * DH, q);
            store_f32v<DH>(sBv + lane * DH, dO);
        }
        __syncthreads();
        if (lane < NT) {   // lane = key j
            float dk[DH], dv[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
#pragma unroll 7
            for (int i = 0; i < NT; ++i) {
                const float pij = sP[i * NT + lane], dsij = sS[i * NT + lane];
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 o4 = *reinterpret_cast<const float4*>(sBv + i * DH + d);
                    const float4 q4 = *reinterpret_cast<const float4*>(sA + i * DH + d);
                    dv[d] = fmaf(pij, o4.x, dv[d]); dv[d + 1] = fmaf(pij, o4.y, dv[d + 1]);
                    dv[d + 2] = fmaf(pij, o4.z, dv[d + 2]); dv[d + 3] = fmaf(pij, o4.w, dv[d + 3]);
                    dk[d] = fmaf(dsij, q4.x, dk[d]); dk[d + 1] = fmaf(dsij, q4.y, dk[d + 1]);
                    dk[d + 2] = fmaf(dsij, q4.z, dk[d + 2]); dk[d + 3] = fmaf(dsij, q4.w, dk[d + 3]);
                }
            }
#pragma unroll
            for (int d = 0; d < DH; ++d) dk[d] *= a.scale;
            store_row32<T>(dqkv + (size_t)myrow * pitch + C + head * DH, dk);
            store_row32<T>(dqkv + (size_t)myrow * pitch + 2 * C + head * DH, dv);
        }
    }
    if (lane < NT && a.dbias_part) {
        float* part = a.dbias_part + ((size_t)head * gridDim.x + blockIdx.x) * (NT * NT);
#pragma unroll
        for (int j = 0; j < NT; ++j) part[j * NT + lane] = dbacc[j];
    } else if (lane < NT && a.dbias_t) {
#pragma unroll
        for (int j = 0; j < NT; ++j) atomicAdd(a.dbias_t + (size_t)head * NT * NT + j * NT + lane, dbacc[j]);
    }
}

// relative-position index of (query i, key j) in a WS x WS window (modeling_swin.py:350-365)
__device__ __forceinline__ int rel_index(int i, int j, int ws) {
    const int yi = i / ws, xi = i - yi * ws, yj = j / ws, xj = j - yj * ws;
    return (yi - yj + ws - 1) * (2 * ws - 1) + (xi - xj + ws - 1);
}

// bias_t[h][j][i] = table[rel_index(i, j)][h]
__global__ void swin_bias_expand_kernel(const float* __restrict__ table, float* __restrict__ bias_t, int ws, int heads) {
    const int NT = ws * ws;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= heads * NT * NT) return;
    const int h = e / (NT * NT), r = e - h * NT * NT;
    const int j = r / NT, i = r - j * NT;
    bias_t[e] = table[rel_index(i, j, ws) * heads + h];
}

// dtable[rel_index(i, j)][h] += dbias_t[h][j][i]: ONE thread per table entry gathers its (i, j) pairs in a fixed order (the first version
// scattered the 2 401 cells of a head onto the 169 entries with atomics: up to 49 adds per entry in arrival order)
__global__ void swin_bias_reduce_kernel(const float* __restrict__ dbias_t, float* __restrict__ dtable, int ws, int heads) {
    const int NT = ws * ws, R = 2 * ws - 1;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * R * heads) return;
    const int rel = e / heads, h = e - rel * heads;
    const int dy = rel / R - (ws - 1), dx = rel % R - (ws - 1);      // yi - yj, xi - xj of every pair behind this entry
    float acc = 0.f;
    for (int yi = 0; yi < ws; ++yi) {
        const int yj = yi - dy;
        if (yj < 0 || yj >= ws) continue;
        for (int xi = 0; xi < ws; ++xi) {
            const int xj = xi - dx;
            if (xj < 0 || xj >= ws) continue;
            acc += dbias_t[((size_t)h * NT + (yj * ws + xj)) * NT + (yi * ws + xi)];
        }
    }
    dtable[rel * heads + h] += acc;
}

// ---- patchify: out[(n, py, px), (c, i, j)] = pixels[n, c, py*ps + i, px*ps + j]; thread = one (row, c, i) run of ps pixels
template <typename T>
__global__ __launch_bounds__(256) void swin_patchify_kernel(const float* __restrict__ px, T* __restrict__ out, int n_img,
                                                            int Cin, int R, int ps, int ld_out) {
    const int G = R / ps;
    const int runs = Cin * ps;                       // runs per output row
    const size_t total = (size_t)n_img * G * G * runs;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / runs;
        const int run = (int)(e - row * runs);
        const int c = run / ps, i = run - c * ps;
        const int n = (int)(row / (G * G)), rr = (int)(row - (size_t)n * G * G);
        const int py = rr / G, pxx = rr - py * G;
        const float* src = px + (((size_t)n * Cin + c) * R + (py * ps + i)) * R + pxx * ps;
        T* dst = out + row * ld_out + run * ps;
        if (ps == 4) {
            float v[4];
            io<float>::load4(src, v);
            io<T>::store4(dst, v);
        } else {
            for (int j = 0; j < ps; ++j) io<T>::store1(dst + j, src[j]);
        }
    }
}

// Same rows straight from decoded uint8 HWC images (what PIL / the LMDB records of V/data_utils/dataset.py:61-99 hold after
// Resize): ToTensor (x / 255) and Normalize((x - mean) / std) of `:69-73` are applied in flight, in the reference's fp32
// operation order, so the fp32 rows are bit-identical to patchifying the host-normalised tensor -- at a quarter of the H2D and
// HBM bytes.  thread = one (row, i) run: ps pixels x 3 interleaved channels.
template <typename T>
__global__ __launch_bounds__(256) void swin_patchify_u8_kernel(const uint8_t* __restrict__ px, T* __restrict__ out, int n_img,
                                                               int Cin, int R, int ps, int ld_out, float mean, float std) {
    const int G = R / ps;
    const size_t total = (size_t)n_img * G * G * ps;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / ps;
        const int i = (int)(e - row * ps);
        const int n = (int)(row / (G * G)), rr = (int)(row - (size_t)n * G * G);
        const int py = rr / G, pxx = rr - py * G;
        const uint8_t* src = px + (((size_t)n * R + (py * ps + i)) * R + pxx * ps) * Cin;
        T* dst = out + row * ld_out + i * ps;
        for (int j = 0; j < ps; ++j)
            for (int c = 0; c < Cin; ++c) {
                const float v = ((float)src[j * Cin + c] / 255.0f - mean) / std;
                io<T>::store1(dst + c * ps * ps + j, v);
            }
    }
}

// ---- patch merging gather: merged[(n, y2, x2), q*C + c] = x[(n, 2 y2 + r, 2 x2 + cc), c], q = cc * 2 + r
// (channel blocks ordered (r0,c0), (r1,c0), (r0,c1), (r1,c1): modeling_swin.py:318-320).  reverse: the inverse copy.
template <typename T>
__global__ __launch_bounds__(256) void swin_merge_kernel(const T* __restrict__ in, T* __restrict__ out, int n_img, int H, int W,
                                                         int C, int reverse) {
    constexpr int EV = vio<T>::EV;
    const int H2 = H / 2, W2 = W / 2, vpr = 4 * C / EV;
    const size_t total = (size_t)n_img * H2 * W2 * vpr;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t mrow = e / vpr;
        const int col = (int)(e - mrow * vpr) * EV;
        const int q = col / C, c = col - q * C;
        const int cc = q >> 1, r = q & 1;
        const int n = (int)(mrow / (H2 * W2)), rr = (int)(mrow - (size_t)n * H2 * W2);
        const int y2 = rr / W2, x2 = rr - y2 * W2;
        const size_t xrow = ((size_t)n * H + 2 * y2 + r) * W + 2 * x2 + cc;
        float v[EV];
        if (!reverse) {
            vio<T>::load(in + xrow * C + c, v);
            vio<T>::store(out + mrow * 4 * C + col, v);
        } else {
            vio<T>::load(in + mrow * 4 * C + col, v);
            vio<T>::store(out + xrow * C + c, v);
        }
    }
}

// ---- mean pool over the Tn tokens of each image, and its backward
template <typename T>
__global__ __launch_bounds__(256) void swin_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int n_img, int Tn, int C) {
    constexpr int EV = vio<T>::EV;
    const int vpr = C / EV;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_img * vpr) return;
    const int n = e / vpr, c = (e - n * vpr) * EV;
    float acc[EV];
#pragma unroll
    for (int k = 0; k < EV; ++k) acc[k] = 0.f;
    for (int t = 0; t < Tn; ++t) {
        float v[EV];
        vio<T>::load(x + ((size_t)n * Tn + t) * C + c, v);
#pragma unroll
        for (int k = 0; k < EV; ++k) acc[k] += v[k];
    }
    const float inv = 1.0f / (float)Tn;
#pragma unroll
    for (int k = 0; k < EV; ++k) acc[k] *= inv;
    vio<T>::store(out + (size_t)n * C + c, acc);
}
template <typename T>
__global__ __launch_bounds__(256) void swin_pool_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, int n_img, int Tn, int C) {
    constexpr int EV = vio<T>::EV;
    const int vpr = C / EV;
    const size_t total = (size_t)n_img * Tn * vpr;
    const float inv = 1.0f / (float)Tn;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / vpr;
        const int c = (int)(e - row * vpr) * EV;
        float v[EV];
        vio<T>::load(dout + (row / Tn) * C + c, v);
#pragma unroll
        for (int k = 0; k < EV; ++k) v[k] *= inv;
        vio<T>::store(dx + row * C + c, v);
    }
}

// ---- out = res + rowscale[row / rps] * (a + bias)   (stage-final residual sums that no LayerNorm consumes directly)
template <typename T>
__global__ __launch_bounds__(256) void bias_residual_kernel(const T* __restrict__ a, const float* __restrict__ bias,
                                                            const T* __restrict__ res, const float* __restrict__ rowscale,
                                                            int rps, T* __restrict__ out, size_t M, int N) {
    constexpr int EV = vio<T>::EV;
    const int vpr = N / EV;
    const size_t total = M * vpr;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / vpr;
        const int c = (int)(e - row * vpr) * EV;
        float v[EV], r[EV];
        vio<T>::load(a + row * N + c, v);
        if (bias) {
            float b[EV];
            load_f32v<EV>(bias + c, b);
#pragma unroll
            for (int k = 0; k < EV; ++k) v[k] += b[k];
        }
        if (rowscale) {
            const float sc = rowscale[row / rps];
#pragma unroll
            for (int k = 0; k < EV; ++k) v[k] *= sc;
        }
        vio<T>::load(res + row * N + c, r);
#pragma unroll
        for (int k = 0; k < EV; ++k) v[k] += r[k];
        vio<T>::store(out + row * N + c, v);
    }
}

__global__ void droppath_scale_kernel(float* __restrict__ out, int n, DropRng d) {
    d = drop_resolve(d);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = drop_keep(d, (uint64_t)i) ? d.inv_keep : 0.f;
}

int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 256 * 32); }

int check_attn(const morec_swin_attn_desc* d) {
    if (!d) return MOREC_E_ARG;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->heads <= 0 || d->window <= 0) return MOREC_E_ARG;
    if (d->dh != DH) return MOREC_E_UNSUPPORTED;              // every published Swin has 32-wide heads
    if (d->window != 7) return MOREC_E_UNSUPPORTED;           // 7 x 7 = 49 tokens <= one wavefront of query rows
    if (d->H % d->window || d->W % d->window) return MOREC_E_UNSUPPORTED;   // (HF pads; no hot-path config needs it)
    if (d->shift < 0 || d->shift >= d->window) return MOREC_E_ARG;
    if (d->dtype != MOREC_F32 && !is_h16(d->dtype)) return MOREC_E_DTYPE;
    return MOREC_OK;
}

SwinAttnArgs make_args(const morec_swin_attn_desc* d, int& gx) {
    SwinAttnArgs a{};
    a.n_img = d->n_img; a.H = d->H; a.W = d->W; a.shift = d->shift; a.heads = d->heads; a.scale = d->scale;
    a.n_win_total = d->n_img * (d->H / d->window) * (d->W / d->window);
    // enough blocks to fill 256 CUs several times over, few enough that the per-block bias load / dbias flush amortise
    const long tiles = (long)a.n_win_total * d->heads;
    int wpb = (int)std::max<long>(1, std::min<long>(64, tiles / 4096));
    a.wpb = wpb;
    gx = (a.n_win_total + wpb - 1) / wpb;
    return a;
}
}  // namespace

// 16-bit fast path on the matrix cores (swin_attn_mfma.hip); MOREC_E_UNSUPPORTED = shape / dtype outside it
int morec_swin_attn_mfma_launch(const morec_swin_attn_desc* d, const void* qkv, const float* bias_t, void* ctx, const void* dctx,
                                void* dqkv, float* dbias_t, bool backward, hipStream_t s, float* csum = nullptr, long csum_rows = 0,
                                int* csum_rows_needed = nullptr);
int colsum_f32_launch(const float* in, float* out, int rows, int N, hipStream_t s);

extern "C" int morec_swin_attn_fwd(const morec_swin_attn_desc* d, const void* qkv, const float* bias_t, void* ctx,
                                   void* stream) {
    int rc = check_attn(d);
    if (rc) return rc;
    if (!qkv || !bias_t || !ctx) return MOREC_E_ARG;
    int gx;
    SwinAttnArgs a = make_args(d, gx);
    a.qkv = qkv; a.bias_t = bias_t; a.ctx = ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    rc = morec_swin_attn_mfma_launch(d, qkv, bias_t, ctx, nullptr, nullptr, nullptr, false, s);
    if (rc != MOREC_E_UNSUPPORTED) return rc;
    dim3 grid(gx, d->heads), block(64);
    if (d->dtype == MOREC_F32) hipLaunchKernelGGL((swin_attn_fwd_kernel<float, 7>), grid, block, 0, s, a);
    else if (d->dtype == MOREC_F16) hipLaunchKernelGGL((swin_attn_fwd_kernel<f16, 7>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((swin_attn_fwd_kernel<bf16, 7>), grid, block, 0, s, a);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_swin_attn_bwd(const morec_swin_attn_desc* d, const void* qkv, const float* bias_t, const void* ctx,
                                   const void* dctx, void* dqkv, float* dbias_t, void* stream) {
    int rc = check_attn(d);
    if (rc) return rc;
    if (!qkv || !bias_t || !ctx || !dctx || !dqkv) return MOREC_E_ARG;
    int gx;
    SwinAttnArgs a = make_args(d, gx);
    a.qkv = qkv; a.bias_t = bias_t; a.ctx = const_cast<void*>(ctx); a.dctx = dctx; a.dqkv = dqkv; a.dbias_t = dbias_t;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    rc = morec_swin_attn_mfma_launch(d, qkv, bias_t, const_cast<void*>(ctx), dctx, dqkv, dbias_t, true, s);
    if (rc != MOREC_E_UNSUPPORTED) return rc;
    dim3 grid(gx, d->heads), block(64);
    const int NT2 = d->window * d->window * d->window * d->window;
    if (dbias_t && morec_deterministic()) {      // per-block partial tiles, folded in block order below
        a.dbias_part = morec_det_scratch(s, (size_t)d->heads * gx * NT2);
        if (!a.dbias_part) return (int)hipErrorOutOfMemory;
    }
    if (d->dtype == MOREC_F32) hipLaunchKernelGGL((swin_attn_bwd_kernel<float, 7>), grid, block, 0, s, a);
    else if (d->dtype == MOREC_F16) hipLaunchKernelGGL((swin_attn_bwd_kernel<f16, 7>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((swin_attn_bwd_kernel<bf16, 7>), grid, block, 0, s, a);
    MOREC_CHECK_LAUNCH();
    if (a.dbias_part)
        for (int h = 0; h < d->heads; ++h) {
            rc = morec_det_fold_add(a.dbias_part + (size_t)h * gx * NT2, dbias_t + (size_t)h * NT2, gx, (size_t)NT2, (size_t)NT2, s);
            if (rc) return rc;
        }
    return MOREC_OK;
}

extern "C" int morec_swin_bias_expand(const float* table, float* bias_t, int window, int heads, void* stream) {
    if (!table || !bias_t || window <= 0 || heads <= 0) return MOREC_E_ARG;
    const int n = heads * window * window * window * window;
    hipLaunchKernelGGL(swin_bias_expand_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       table, bias_t, window, heads);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_swin_bias_reduce(const float* dbias_t, float* dtable, int window, int heads, void* stream) {
    if (!dbias_t || !dtable || window <= 0 || heads <= 0) return MOREC_E_ARG;
    const int n = heads * (2 * window - 1) * (2 * window - 1);
    hipLaunchKernelGGL(swin_bias_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       dbias_t, dtable, window, heads);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_swin_patchify(const float* pixels, void* out, int n_img, int channels, int R, int patch, int ld_out,
                                   int dtype, void* stream) {
    if (!pixels || !out || n_img <= 0 || channels <= 0 || R <= 0 || patch <= 0) return MOREC_E_ARG;
    if (R % patch) return MOREC_E_UNSUPPORTED;
    if (ld_out < channels * patch * patch || ld_out % 4) return MOREC_E_ALIGN;
    if (patch == 4 && !aligned16(pixels)) return MOREC_E_ALIGN;
    const size_t total = (size_t)n_img * (R / patch) * (R / patch) * channels * patch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((swin_patchify_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, pixels, (float*)out, n_img, channels, R, patch, ld_out);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((swin_patchify_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, pixels, (bf16*)out, n_img, channels, R, patch, ld_out);
    else if (dtype == MOREC_F16)
        hipLaunchKernelGGL((swin_patchify_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, pixels, (f16*)out, n_img, channels, R, patch, ld_out);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_swin_patchify_u8(const uint8_t* pixels_hwc, void* out, int n_img, int channels, int R, int patch, int ld_out,
                                      float mean, float std, int dtype, void* stream) {
    if (!pixels_hwc || !out || n_img <= 0 || channels <= 0 || R <= 0 || patch <= 0 || std == 0.f) return MOREC_E_ARG;
    if (R % patch) return MOREC_E_UNSUPPORTED;
    if (ld_out < channels * patch * patch) return MOREC_E_ALIGN;
    const size_t total = (size_t)n_img * (R / patch) * (R / patch) * patch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((swin_patchify_u8_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, pixels_hwc, (float*)out, n_img, channels, R, patch, ld_out, mean, std);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((swin_patchify_u8_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, pixels_hwc, (bf16*)out, n_img, channels, R, patch, ld_out, mean, std);
    else if (dtype == MOREC_F16)
        hipLaunchKernelGGL((swin_patchify_u8_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, pixels_hwc, (f16*)out, n_img, channels, R, patch, ld_out, mean, std);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_swin_merge(const void* in, void* out, int n_img, int H, int W, int C, int reverse, int dtype, void* stream) {
    if (!in || !out || n_img <= 0 || H <= 0 || W <= 0 || C <= 0) return MOREC_E_ARG;
    if (H % 2 || W % 2) return MOREC_E_UNSUPPORTED;           // (HF pads odd maps; no hot-path config needs it)
    if (C % 8) return MOREC_E_ALIGN;
    const size_t total = (size_t)n_img * (H / 2) * (W / 2) * (4 * C / (dtype == MOREC_F32 ? 4 : 8));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((swin_merge_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)in, (float*)out, n_img, H, W, C, reverse);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((swin_merge_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)in, (bf16*)out, n_img, H, W, C, reverse);
    else if (dtype == MOREC_F16)
        hipLaunchKernelGGL((swin_merge_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, (const f16*)in, (f16*)out, n_img, H, W, C, reverse);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_swin_pool_fwd(const void* x, void* out, int n_img, int tokens, int C, int dtype, void* stream) {
    if (!x || !out || n_img <= 0 || tokens <= 0 || C <= 0) return MOREC_E_ARG;
    if (C % 8) return MOREC_E_ALIGN;
    const int n = n_img * (C / (dtype == MOREC_F32 ? 4 : 8));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((swin_pool_fwd_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, s, (const float*)x, (float*)out, n_img, tokens, C);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((swin_pool_fwd_kernel<bf16>), dim3((n + 255) / 256), dim3(256), 0, s, (const bf16*)x, (bf16*)out, n_img, tokens, C);
    else if (dtype == MOREC_F16)
        hipLaunchKernelGGL((swin_pool_fwd_kernel<f16>), dim3((n + 255) / 256), dim3(256), 0, s, (const f16*)x, (f16*)out, n_img, tokens, C);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_swin_pool_bwd(const void* dout, void* dx, int n_img, int tokens, int C, int dtype, void* stream) {
    if (!dout || !dx || n_img <= 0 || tokens <= 0 || C <= 0) return MOREC_E_ARG;
    if (C % 8) return MOREC_E_ALIGN;
    const size_t total = (size_t)n_img * tokens * (C / (dtype == MOREC_F32 ? 4 : 8));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((swin_pool_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)dout, (float*)dx, n_img, tokens, C);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((swin_pool_bwd_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)dout, (bf16*)dx, n_img, tokens, C);
    else if (dtype == MOREC_F16)
        hipLaunchKernelGGL((swin_pool_bwd_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, (const f16*)dout, (f16*)dx, n_img, tokens, C);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_bias_residual(const void* a, const float* bias, const void* res, const float* rowscale,
                                   int rows_per_scale, void* out, int M, int N, int dtype, void* stream) {
    if (!a || !res || !out || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (rowscale && rows_per_scale <= 0) return MOREC_E_ARG;
    if (N % 8) return MOREC_E_ALIGN;
    const size_t total = (size_t)M * (N / (dtype == MOREC_F32 ? 4 : 8));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((bias_residual_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)a, bias, (const float*)res, rowscale, rows_per_scale, (float*)out, (size_t)M, N);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((bias_residual_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)a, bias, (const bf16*)res, rowscale, rows_per_scale, (bf16*)out, (size_t)M, N);
    else if (dtype == MOREC_F16)
        hipLaunchKernelGGL((bias_residual_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, (const f16*)a, bias, (const f16*)res, rowscale, rows_per_scale, (f16*)out, (size_t)M, N);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_droppath_scale(float* out, int n, float p, uint64_t seed, void* stream) {
    if (!out || n <= 0 || p < 0.f || p >= 1.f) return MOREC_E_ARG;
    hipLaunchKernelGGL(droppath_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, n,
                       make_drop(p, seed));
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// morec_swin_attn_bwd + the bias gradient of the fused q|k|v projection (HF SwinSelfAttention query / key / value biases,
// modeling_swin.py:407-409): dbqkv[3 C] (fp32) += column sums of dqkv.  With ws (ws_bytes >= morec-chosen rows x 3 C floats; 64 MiB
// always suffices for the shapes of this library) the bf16 MFMA path sums inside the attention kernel -- fp32 sums of the
// UNROUNDED rows -- and one small kernel folds the per-wavefront rows; otherwise: the plain backward followed by morec_colsum.
extern "C" int morec_swin_attn_bwd_dbias(const morec_swin_attn_desc* d, const void* qkv, const float* bias_t, const void* ctx,
                                         const void* dctx, void* dqkv, float* dbias_t, float* dbqkv, float* ws, size_t ws_bytes,
                                         void* stream) {
    int rc = check_attn(d);
    if (rc) return rc;
    if (!qkv || !bias_t || !ctx || !dctx || !dqkv || !dbqkv) return MOREC_E_ARG;
    const int C3 = 3 * d->heads * d->dh;
    const long rows = (long)d->n_img * d->H * d->W;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (ws) {
        int need = 0;
        rc = morec_swin_attn_mfma_launch(d, qkv, bias_t, const_cast<void*>(ctx), dctx, dqkv, dbias_t, true, s, ws,
                                         (long)(ws_bytes / (C3 * sizeof(float))), &need);
        if (rc == MOREC_OK) return colsum_f32_launch(ws, dbqkv, need, C3, s);
        if (rc != MOREC_E_UNSUPPORTED) return rc;
    }
    rc = morec_swin_attn_bwd(d, qkv, bias_t, ctx, dctx, dqkv, dbias_t, stream);
    if (rc) return rc;
    return morec_colsum(dqkv, dbqkv, (int)rows, C3, C3, d->dtype, stream);
}
