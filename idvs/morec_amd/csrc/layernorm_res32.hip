// layernorm_res32.hip -- LayerNorm of the reference's AUTOCAST data flow: under `torch.cuda.amp.autocast()` (T/run.py:242) the Linear layers
// take and return 16-bit tensors, but LayerNorm runs -- and RETURNS -- fp32 (HF modeling_bert.py BertSelfOutput / BertOutput:
// `LayerNorm(dropout(dense(h)) + input_tensor)`; T/model/modules.py:17,63,93), so the RESIDUAL STREAM of BERT and of the SASRec user
// encoder is fp32 and only the GEMM operands are rounded to 16 bits.  These kernels are that LayerNorm for the `*_res32` compute modes:
//
//   forward   z32 = res32 + dropout(x16 + bias) (+ pos)          x16: the 16-bit output of the sub-layer's last GEMM
//             y32 = LN(z32) * gamma + beta (dropout p_out)        the residual stream (next LayerNorm's res32)
//             y16 = round16(y32)                                  the next GEMM's A operand (autocast's cast in front of a Linear)
//   backward  g   = dy16 (through the GEMM) + dy32 (along the residual stream)   -> dz32 (residual branch, fp32)
//             dzd16 = round16(dropout'(dz32))                     what the sub-layer's weight / input gradient GEMMs read
//
// Same lane layout as layernorm.hip's 16-bit kernels: a lane owns vectors of 8 consecutive elements (16 bytes of a 16-bit row, two
// 16-byte accesses of an fp32 row); one row per LPR lanes, the row in registers, shuffles for the two reductions.  HBM-bound:
// forward 2 + 4 bytes in, 4 + 4 + 2 out per element; backward 2 + 4 + 4 in, 4 + 2 out.
#include <algorithm>
#include "common.hpp"

namespace {
constexpr int EV = 8;

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void load_f32x8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void store_f32x8(float* p, const float (&o)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

// The residual stream in PRE-LayerNorm form: `res` holds the previous LayerNorm's input z, and the stream value is LN(z) = (z - mean[row]) *
// rstd[row] * gamma + beta, recomputed here (a few FMAs per element under a memory-bound kernel) instead of having been written by that
// LayerNorm and read back: 4 of the 16 bytes per element the forward moves.  mean == nullptr: `res` is the stream itself.
struct ResLN { const float* mean; const float* rstd; const float* gamma; const float* beta; };

template <typename T16, int VPL, int LPR, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_res32_kernel(const T16* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ res,
                                                           const float* __restrict__ pos, int pos_period, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float* __restrict__ z_out,
                                                           float* __restrict__ y32, T16* __restrict__ y16, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, int M, int N, DropRng din, DropRng dout, const ResLN rl) {
    din = drop_resolve(din);
    dout = drop_resolve(dout);
    constexpr int GRP = 64 / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * GRP + ((threadIdx.x & 63) / LPR);
    if (row >= M) return;
    float rl_mu = 0.f, rl_rs = 0.f;
    if (rl.mean) { rl_mu = rl.mean[row]; rl_rs = rl.rstd[row]; }
    const size_t base = (size_t)row * N;
    // every row-sized load first, from clamped (unguarded) addresses: one memory round trip per row (see layernorm.hip)
    uint4 rx[VPL];
    float rr[VPL][EV];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
        rx[i] = vio<T16>::load_raw(x + base + cl);
    }
    if (res) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
            load_f32x8(res + base + cl, rr[i]);
        }
    }
    float v[VPL][EV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * LPR + lane) * EV;
        if (FULL || c < N) {
            vio<T16>::unpack(rx[i], v[i]);
            if (bias) {
                float b[EV];
                load_f32v<EV>(bias + c, b);
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] += b[k];
            }
            if (din.thresh) {   // dropout on the sub-layer output BEFORE the residual add (modules.py:16,62; HF Bert*Output)
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] *= din.inv_keep;
                bool kp[EV];
                drop_keep_vec<EV>(din, base + c, kp);
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] = kp[k] ? v[i][k] : 0.f;
            }
            if (res) {
                if (rl.mean) {      // the residual is handed over as the PREVIOUS LayerNorm's input: its output, recomputed (morec_layernorm_fwd_res32_pre)
                    float pg[EV], pb[EV];
                    load_f32v<EV>(rl.gamma + c, pg);
                    load_f32v<EV>(rl.beta + c, pb);
#pragma unroll
                    for (int k = 0; k < EV; ++k) rr[i][k] = (rr[i][k] - rl_mu) * rl_rs * pg[k] + pb[k];
                }
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] += rr[i][k];
            }
            if (pos) {
                float b[EV];
                load_f32v<EV>(pos + (size_t)(row % pos_period) * N + c, b);
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] += b[k];
            }
            if (z_out) store_f32x8(z_out + base + c, v[i]);
#pragma unroll
            for (int k = 0; k < EV; ++k) sum += v[i][k];
        } else {
#pragma unroll
            for (int k = 0; k < EV; ++k) v[i][k] = 0.f;
        }
    }
    const float mean = group_sum<LPR>(sum) / (float)N;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * LPR + lane) * EV;
        if (FULL || c < N) {
#pragma unroll
            for (int k = 0; k < EV; ++k) {
                const float d = v[i][k] - mean;
                sq += d * d;
            }
        }
    }
    const float var = group_sum<LPR>(sq) / (float)N;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * LPR + lane) * EV;
        if (FULL || c < N) {
            float g[EV], b[EV], o[EV];
            load_f32v<EV>(gamma + c, g);
            load_f32v<EV>(beta + c, b);
#pragma unroll
            for (int k = 0; k < EV; ++k) o[k] = (v[i][k] - mean) * rstd * g[k] + b[k];
            if (dout.thresh) {  // dropout on the LayerNorm output (embedding stages: modules.py:93-94)
#pragma unroll
                for (int k = 0; k < EV; ++k) o[k] *= dout.inv_keep;
                bool kp[EV];
                drop_keep_vec<EV>(dout, base + c, kp);
#pragma unroll
                for (int k = 0; k < EV; ++k) o[k] = kp[k] ? o[k] : 0.f;
            }
            if (y32) store_f32x8(y32 + base + c, o);
            if (y16) vio<T16>::store(y16 + base + c, o);
        }
    }
}

// Backward.  A block owns rows_per_block consecutive rows, one row per LPR lanes per trip; per-column dgamma / dbeta / dbias partials
// of the block's row groups are folded through LDS into one atomic per column per block (`det`: into the block's own partial row, folded
// in block order by the launcher -- deterministic mode).
template <typename T16, int VPL, int LPR, bool FULL>
__global__ __launch_bounds__(256, (VPL <= 3 ? 2 : 1)) void ln_bwd_res32_kernel(const T16* __restrict__ dy16, const float* __restrict__ dy32, const float* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, float* __restrict__ dz32, T16* __restrict__ dzd16,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias, int M, int N,
                                                           int rows_per_block, DropRng din, DropRng dout, float* __restrict__ det) {
    din = drop_resolve(din);
    dout = drop_resolve(dout);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int GRP = 64 / LPR, NP = 4 * GRP;
    float* sgm = reinterpret_cast<float*>(smem_raw);     // [N] gamma
    float* sg = sgm + N;                                 // [NP][N] dgamma partials
    float* sb = sg + NP * (size_t)N;                    // [NP][N] dbeta partials
    float* sd = sb + NP * (size_t)N;                    // [NP][N] dbias partials (only when dbias)
    const int lane = threadIdx.x & (LPR - 1), grp = (threadIdx.x >> 6) * GRP + ((threadIdx.x & 63) / LPR);
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float ag[VPL][EV], ab[VPL][EV], ad[VPL][EV];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
#pragma unroll
        for (int k = 0; k < EV; ++k) { ag[i][k] = 0.f; ab[i][k] = 0.f; ad[i][k] = 0.f; }
    }
    for (int c = threadIdx.x; c < N; c += 256) sgm[c] = gamma[c];
    __syncthreads();
    for (int row = r0 + grp; row < r1; row += NP) {
        const size_t base = (size_t)row * N;
        // all loads of the row up front (clamped addresses, no branches between them)
        uint4 ra[VPL];
        float rb[VPL][EV], rz[VPL][EV];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
            if (dy16) ra[i] = vio<T16>::load_raw(dy16 + base + cl);
            load_f32x8(z + base + cl, rz[i]);
        }
        if (dy32) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
                load_f32x8(dy32 + base + cl, rb[i]);
            }
        }
        const float mu = mean[row], rs = rstd[row];
        float g[VPL][EV], xh[VPL][EV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV;
            if (FULL || c < N) {
                float d[EV];
                if (dy16) {
                    vio<T16>::unpack(ra[i], d);
                } else {
#pragma unroll
                    for (int k = 0; k < EV; ++k) d[k] = 0.f;
                }
                if (dy32) {
#pragma unroll
                    for (int k = 0; k < EV; ++k) d[k] += rb[i][k];
                }
                if (dout.thresh) {
#pragma unroll
                    for (int k = 0; k < EV; ++k) d[k] *= dout.inv_keep;
                    bool kp[EV];
                    drop_keep_vec<EV>(dout, base + c, kp);
#pragma unroll
                    for (int k = 0; k < EV; ++k) d[k] = kp[k] ? d[k] : 0.f;
                }
                float gmv[EV];
                load_f32v<EV>(sgm + c, gmv);
#pragma unroll
                for (int k = 0; k < EV; ++k) {
                    xh[i][k] = (rz[i][k] - mu) * rs;
                    ag[i][k] += d[k] * xh[i][k];
                    ab[i][k] += d[k];
                    g[i][k] = d[k] * gmv[k];
                    s1 += g[i][k];
                    s2 += g[i][k] * xh[i][k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < EV; ++k) { g[i][k] = 0.f; xh[i][k] = 0.f; }
            }
        }
        s1 = group_sum<LPR>(s1) / (float)N;
        s2 = group_sum<LPR>(s2) / (float)N;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV;
            if (FULL || c < N) {
                float o[EV];
#pragma unroll
                for (int k = 0; k < EV; ++k) o[k] = rs * (g[i][k] - s1 - xh[i][k] * s2);
                if (dz32) store_f32x8(dz32 + base + c, o);
                if (dzd16) {   // gradient w.r.t. the (dropped-out) sub-layer output, rounded to the GEMMs' operand type
                    if (din.thresh) {
#pragma unroll
                        for (int k = 0; k < EV; ++k) o[k] *= din.inv_keep;
                        bool kp[EV];
                        drop_keep_vec<EV>(din, base + c, kp);
#pragma unroll
                        for (int k = 0; k < EV; ++k) o[k] = kp[k] ? o[k] : 0.f;
                    }
                    vio<T16>::store(dzd16 + base + c, o);
                    if (dbias) {
#pragma unroll
                        for (int k = 0; k < EV; ++k) ad[i][k] += io<T16>::round(o[k]);   // what a column sum over the stored tensor would see
                    }
                }
            }
        }
    }
    if (dgamma || dbias) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV;
            if (FULL || c < N) {
                store_f32v<EV>(sg + (size_t)grp * N + c, ag[i]);
                store_f32v<EV>(sb + (size_t)grp * N + c, ab[i]);
                if (dbias) store_f32v<EV>(sd + (size_t)grp * N + c, ad[i]);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < N; c += 256) {
            float tg = 0.f, tb = 0.f, td = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                tg += sg[q * N + c];
                tb += sb[q * N + c];
                if (dbias) td += sd[q * N + c];
            }
            if (det) {
                float* o = det + (size_t)blockIdx.x * 3 * N;
                o[c] = tg; o[N + c] = tb; o[2 * N + c] = td;
                continue;
            }
            if (dgamma) {
                atomicAdd(dgamma + c, tg);
                atomicAdd(dbeta + c, tb);
            }
            if (dbias) atomicAdd(dbias + c, td);
        }
    }
}

template <typename T16, int V, int L, bool FULL>
int bwd_launch(const void* dy16, const float* dy32, const float* z, const float* mean, const float* rstd, const float* gamma, float* dz32, void* dzd16,
               float* dgamma, float* dbeta, float* dbias, int M, int N, DropRng din, DropRng dout, hipStream_t s) {
    const size_t lds = ((dgamma || dbias) ? (size_t)12 * (64 / L) * N : 0) * sizeof(float) + N * sizeof(float);
    static size_t lds_seen = ~(size_t)0;
    static int slots = 0;
    if (lds != lds_seen) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_res32_kernel<T16, V, L, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, n_cu = 0, nb = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ln_bwd_res32_kernel<T16, V, L, FULL>, 256, lds) != hipSuccess || nb <= 0) {
            (void)hipGetLastError();
            nb = 2;
        }
        slots = nb * n_cu;
        lds_seen = lds;
    }
    const int rpb = std::max(M <= 8192 ? 16 : 64, (((M + slots - 1) / slots) + 15) & ~15);      // (small M: see layernorm.hip, ln_bwd launch)
    dim3 grid((M + rpb - 1) / rpb), block(256);
    float* det = nullptr;
    if ((dgamma || dbias) && morec_deterministic()) {
        det = morec_det_scratch(s, (size_t)grid.x * 3 * N);
        if (!det) return (int)hipErrorOutOfMemory;
    }
    hipLaunchKernelGGL((ln_bwd_res32_kernel<T16, V, L, FULL>), grid, block, lds, s, (const T16*)dy16, dy32, z, mean, rstd, gamma, dz32, (T16*)dzd16,
                       dgamma, dbeta, dbias, M, N, rpb, din, dout, det);
    MOREC_CHECK_LAUNCH();
    if (det) {
        int rc = MOREC_OK;
        if (dgamma) {
            rc = morec_det_fold_add(det, dgamma, (int)grid.x, (size_t)N, (size_t)3 * N, s);
            if (rc == MOREC_OK) rc = morec_det_fold_add(det + N, dbeta, (int)grid.x, (size_t)N, (size_t)3 * N, s);
        }
        if (rc == MOREC_OK && dbias) rc = morec_det_fold_add(det + 2 * (size_t)N, dbias, (int)grid.x, (size_t)N, (size_t)3 * N, s);
        return rc;
    }
    return MOREC_OK;
}

template <typename T16>
int fwd_dispatch(const void* x, const float* bias, const float* res, const float* pos, int pos_period, const float* gamma, const float* beta, float eps,
                 float* z_out, float* y32, void* y16, float* mean, float* rstd, int M, int N, DropRng din, DropRng dout, hipStream_t s,
                 const ResLN rl = ResLN{nullptr, nullptr, nullptr, nullptr}) {
    const int vpl = (N + 64 * EV - 1) / (64 * EV);
#define LNF(V, L, F)                                                                                                                             \
    hipLaunchKernelGGL((ln_fwd_res32_kernel<T16, V, L, F>), dim3((M + 4 * (64 / L) - 1) / (4 * (64 / L))), dim3(256), 0, s, (const T16*)x, bias, res, pos, \
                       pos_period, gamma, beta, eps, z_out, y32, (T16*)y16, mean, rstd, M, N, din, dout, rl)
    if (N <= 16 * EV) LNF(1, 16, false);
    else if (N <= 32 * EV) LNF(1, 32, false);
    else if (N == 96 * EV) LNF(3, 32, true);      // H = 768: 32 lanes x 3 vectors, two rows per wave, no bounds checks
    else if (vpl <= 1) LNF(1, 64, false);
    else if (vpl <= 2) LNF(2, 64, false);
    else if (vpl <= 3) LNF(3, 64, false);
    else if (vpl <= 4) LNF(4, 64, false);
    else if (vpl <= 8) LNF(8, 64, false);
    else return MOREC_E_UNSUPPORTED;
#undef LNF
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

template <typename T16>
int bwd_dispatch(const void* dy16, const float* dy32, const float* z, const float* mean, const float* rstd, const float* gamma, float* dz32, void* dzd16,
                 float* dgamma, float* dbeta, float* dbias, int M, int N, DropRng din, DropRng dout, hipStream_t s) {
    const int vpl = (N + 64 * EV - 1) / (64 * EV);
#define LNB(V, L, F) return bwd_launch<T16, V, L, F>(dy16, dy32, z, mean, rstd, gamma, dz32, dzd16, dgamma, dbeta, dbias, M, N, din, dout, s)
    if (N <= 16 * EV) LNB(1, 16, false);
    else if (N <= 32 * EV) LNB(1, 32, false);
    else if (N == 96 * EV) LNB(3, 32, true);
    else if (vpl <= 1) LNB(1, 64, false);
    else if (vpl <= 2) LNB(2, 64, false);
    else if (vpl <= 3) LNB(3, 64, false);
    else if (vpl <= 4) LNB(4, 64, false);
    else if (vpl <= 8) LNB(8, 64, false);
#undef LNB
    return MOREC_E_UNSUPPORTED;
}
}  // namespace

extern "C" int morec_layernorm_fwd_res32(const void* x16, const float* bias, const float* res32, const float* pos, int pos_period, const float* gamma,
                                         const float* beta, float eps, float* z32, float* y32, void* y16, float* mean, float* rstd, int M, int N,
                                         int dtype16, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, void* stream) {
    if (!x16 || !gamma || !beta || (!y32 && !y16) || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (p_in < 0.f || p_in >= 1.f || p_out < 0.f || p_out >= 1.f) return MOREC_E_ARG;
    if (pos && pos_period <= 0) return MOREC_E_ARG;
    if (N % EV) return MOREC_E_ALIGN;
    if (!aligned16(x16) || (res32 && !aligned16(res32)) || (z32 && !aligned16(z32)) || (y32 && !aligned16(y32)) || (y16 && !aligned16(y16))) return MOREC_E_ALIGN;
    const DropRng din = make_drop(p_in, seed_in), dout = make_drop(p_out, seed_out);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype16 == MOREC_BF16) return fwd_dispatch<bf16>(x16, bias, res32, pos, pos_period, gamma, beta, eps, z32, y32, y16, mean, rstd, M, N, din, dout, s);
    if (dtype16 == MOREC_F16) return fwd_dispatch<f16>(x16, bias, res32, pos, pos_period, gamma, beta, eps, z32, y32, y16, mean, rstd, M, N, din, dout, s);
    return MOREC_E_DTYPE;
}

extern "C" int morec_layernorm_fwd_res32_pre(const void* x16, const float* bias, const float* res_z32, const float* res_mean, const float* res_rstd,
                                             const float* res_gamma, const float* res_beta, const float* gamma, const float* beta, float eps, float* z32,
                                             void* y16, float* mean, float* rstd, int M, int N, int dtype16, float p_in, uint64_t seed_in, void* stream) {
    if (!x16 || !res_z32 || !res_mean || !res_rstd || !res_gamma || !res_beta || !gamma || !beta || !y16 || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (p_in < 0.f || p_in >= 1.f) return MOREC_E_ARG;
    if (N % EV) return MOREC_E_ALIGN;
    if (!aligned16(x16) || !aligned16(res_z32) || (z32 && !aligned16(z32)) || !aligned16(y16) || !aligned16(res_gamma) || !aligned16(res_beta)) return MOREC_E_ALIGN;
    const DropRng din = make_drop(p_in, seed_in), dout = make_drop(0.f, 0);
    const ResLN rl{res_mean, res_rstd, res_gamma, res_beta};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype16 == MOREC_BF16) return fwd_dispatch<bf16>(x16, bias, res_z32, nullptr, 0, gamma, beta, eps, z32, nullptr, y16, mean, rstd, M, N, din, dout, s, rl);
    if (dtype16 == MOREC_F16) return fwd_dispatch<f16>(x16, bias, res_z32, nullptr, 0, gamma, beta, eps, z32, nullptr, y16, mean, rstd, M, N, din, dout, s, rl);
    return MOREC_E_DTYPE;
}

extern "C" int morec_layernorm_bwd_res32(const void* dy16, const float* dy32, const float* z32, const float* mean, const float* rstd, const float* gamma,
                                         float* dz32, void* dzd16, float* dgamma, float* dbeta, float* dbias, int M, int N, int dtype16, float p_in,
                                         uint64_t seed_in, float p_out, uint64_t seed_out, void* stream) {
    if ((!dy16 && !dy32) || !z32 || !mean || !rstd || !gamma || (!dz32 && !dzd16) || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (p_in < 0.f || p_in >= 1.f || p_out < 0.f || p_out >= 1.f) return MOREC_E_ARG;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return MOREC_E_ARG;
    if (dbias && !dzd16) return MOREC_E_ARG;      // the bias gradient is the column sum of the 16-bit sub-layer gradient
    if (N % EV) return MOREC_E_ALIGN;
    if (N > 4096) return MOREC_E_UNSUPPORTED;
    if ((dy16 && !aligned16(dy16)) || (dy32 && !aligned16(dy32)) || !aligned16(z32) || (dz32 && !aligned16(dz32)) || (dzd16 && !aligned16(dzd16))) return MOREC_E_ALIGN;
    const DropRng din = make_drop(p_in, seed_in), dout = make_drop(p_out, seed_out);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype16 == MOREC_BF16) return bwd_dispatch<bf16>(dy16, dy32, z32, mean, rstd, gamma, dz32, dzd16, dgamma, dbeta, dbias, M, N, din, dout, s);
    if (dtype16 == MOREC_F16) return bwd_dispatch<f16>(dy16, dy32, z32, mean, rstd, gamma, dz32, dzd16, dgamma, dbeta, dbias, M, N, din, dout, s);
    return MOREC_E_DTYPE;
}
