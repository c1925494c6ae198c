// layernorm.hip -- LayerNorm forward/backward with the pre-add fused in (bias + residual + position
// rows), one wavefront per row, row held in registers, wave64 shuffles for the two reductions.
// Reference arithmetic: T/model/modules.py:17,63,93 (eps 1e-6) and HF BertSelfOutput / BertOutput /
// BertEmbeddings LayerNorm (eps 1e-12); HBM-bound (one read of each input, one write of each output).
#include "common.hpp"

// VPL = 4-element vectors per lane; a row of N elements needs ceil(N / 256) of them.
template <typename T, int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                     const T* __restrict__ res, const float* __restrict__ pos,
                                                     int pos_period, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ z_out,
                                                     T* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int M, int N, DropRng din,
                                                     DropRng dout) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t base = (size_t)row * N;
    float v[VPL][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < N) {
            io<T>::load4(x + base + c, v[i]);
            if (bias) {
                const float4 b = *reinterpret_cast<const float4*>(bias + c);
                v[i][0] += b.x; v[i][1] += b.y; v[i][2] += b.z; v[i][3] += b.w;
            }
            if (din.thresh) {   // dropout on the sub-layer output BEFORE the residual add (modules.py:16,62; HF Bert*Output)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] = drop_keep(din, base + c + k) ? v[i][k] * din.inv_keep : 0.f;
            }
            if (res) {
                float r[4];
                io<T>::load4(res + base + c, r);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] += r[k];
            }
            if (pos) {
                const float4 b = *reinterpret_cast<const float4*>(pos + (size_t)(row % pos_period) * N + c);
                v[i][0] += b.x; v[i][1] += b.y; v[i][2] += b.z; v[i][3] += b.w;
            }
            if (z_out) {
                io<T>::store4(z_out + base + c, v[i]);
                // the backward pass re-reads z at storage precision: normalise exactly what was stored
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] = io<T>::round(v[i][k]);
            }
            sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)N;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < N) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[i][k] - mean;
                sq += d * d;
            }
        }
    }
    const float var = wave_sum(sq) / (float)N;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < N) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float4 b = *reinterpret_cast<const float4*>(beta + c);
            float o[4];
            o[0] = (v[i][0] - mean) * rstd * g.x + b.x;
            o[1] = (v[i][1] - mean) * rstd * g.y + b.y;
            o[2] = (v[i][2] - mean) * rstd * g.z + b.z;
            o[3] = (v[i][3] - mean) * rstd * g.w + b.w;
            if (dout.thresh) {  // dropout on the LayerNorm output (embedding stages: modules.py:93-94, HF BertEmbeddings)
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = drop_keep(dout, base + c + k) ? o[k] * dout.inv_keep : 0.f;
            }
            io<T>::store4(y + base + c, o);
        }
    }
}

// Backward.  Each block owns RPB consecutive rows; every wave handles TWO rows per trip (both rows' loads are in
// flight together: the kernel is latency-bound otherwise).  Per-column dgamma / dbeta (and the sub-layer bias
// gradient dbias = column sums of dzd) partials are reduced across the block's waves in LDS and leave as ONE
// atomicAdd per column per block.
template <typename T, int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy_a, const T* __restrict__ dy_b,
                                                     const T* __restrict__ z, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     T* __restrict__ dz, T* __restrict__ dzd, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dbias, int M, int N,
                                                     int rows_per_block, DropRng din, DropRng dout) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sg = reinterpret_cast<float*>(smem_raw);  // [4][N] dgamma partials
    float* sb = sg + 4 * (size_t)N;                 // [4][N] dbeta partials
    float* sd = sb + 4 * (size_t)N;                 // [4][N] dbias partials (only when dbias)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float ag[VPL][4], ab[VPL][4], ad[VPL][4], gm[VPL][4];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { ag[i][k] = 0.f; ab[i][k] = 0.f; ad[i][k] = 0.f; gm[i][k] = 0.f; }
        if (c < N) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            gm[i][0] = g.x; gm[i][1] = g.y; gm[i][2] = g.z; gm[i][3] = g.w;
        }
    }
    for (int rowp = r0 + wave * 2; rowp < r1; rowp += 8) {
        float g[2][VPL][4], xh[2][VPL][4];
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, rs[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = min(rowp + j, r1 - 1);          // an odd tail re-reads the last row and skips its stores
            const size_t base = (size_t)row * N;
            const float mu = mean[row];
            rs[j] = rstd[row];
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < N) {
                    float d[4], zz[4];
                    io<T>::load4(dy_a + base + c, d);
                    if (dy_b) {
                        float e[4];
                        io<T>::load4(dy_b + base + c, e);
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[k] += e[k];
                    }
                    if (dout.thresh) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[k] = drop_keep(dout, base + c + k) ? d[k] * dout.inv_keep : 0.f;
                    }
                    io<T>::load4(z + base + c, zz);
                    const bool live = (rowp + j) < r1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        xh[j][i][k] = (zz[k] - mu) * rs[j];
                        if (live) {
                            ag[i][k] += d[k] * xh[j][i][k];
                            ab[i][k] += d[k];
                        }
                        g[j][i][k] = d[k] * gm[i][k];
                        s1[j] += g[j][i][k];
                        s2[j] += g[j][i][k] * xh[j][i][k];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { g[j][i][k] = 0.f; xh[j][i][k] = 0.f; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s1[j] = wave_sum(s1[j]) / (float)N;
            s2[j] = wave_sum(s2[j]) / (float)N;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (rowp + j >= r1) continue;
            const size_t base = (size_t)(rowp + j) * N;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < N) {
                    float o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = rs[j] * (g[j][i][k] - s1[j] - xh[j][i][k] * s2[j]);
                    io<T>::store4(dz + base + c, o);
                    if (dzd) {   // gradient w.r.t. the dropped-out sub-layer output (the residual branch takes dz itself)
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = drop_keep(din, base + c + k) ? o[k] * din.inv_keep : 0.f;
                        io<T>::store4(dzd + base + c, o);
                    }
                    if (dbias) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) ad[i][k] += io<T>::round(o[k]);   // what colsum over the stored tensor would see
                    }
                }
            }
        }
    }
    if (dgamma || dbias) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < N) {
                *reinterpret_cast<float4*>(sg + (size_t)wave * N + c) = make_float4(ag[i][0], ag[i][1], ag[i][2], ag[i][3]);
                *reinterpret_cast<float4*>(sb + (size_t)wave * N + c) = make_float4(ab[i][0], ab[i][1], ab[i][2], ab[i][3]);
                if (dbias) *reinterpret_cast<float4*>(sd + (size_t)wave * N + c) = make_float4(ad[i][0], ad[i][1], ad[i][2], ad[i][3]);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < N; c += 256) {
            if (dgamma) {
                atomicAdd(dgamma + c, sg[c] + sg[N + c] + sg[2 * N + c] + sg[3 * N + c]);
                atomicAdd(dbeta + c, sb[c] + sb[N + c] + sb[2 * N + c] + sb[3 * N + c]);
            }
            if (dbias) atomicAdd(dbias + c, sd[c] + sd[N + c] + sd[2 * N + c] + sd[3 * N + c]);
        }
    }
}

template <typename T>
static int ln_fwd_dispatch(const void* x, const float* bias, const void* res, const float* pos, int pos_period,
                           const float* gamma, const float* beta, float eps, void* z_out, void* y, float* mean,
                           float* rstd, int M, int N, DropRng din, DropRng dout, hipStream_t s) {
    const int vpl = (N + 255) / 256;
    dim3 grid((M + 3) / 4), block(256);
#define LN_FWD(V)                                                                                                   \
    hipLaunchKernelGGL((ln_fwd_kernel<T, V>), grid, block, 0, s, (const T*)x, bias, (const T*)res, pos, pos_period, \
                       gamma, beta, eps, (T*)z_out, (T*)y, mean, rstd, M, N, din, dout)
    if (vpl <= 1) LN_FWD(1);
    else if (vpl <= 2) LN_FWD(2);
    else if (vpl <= 3) LN_FWD(3);
    else if (vpl <= 4) LN_FWD(4);
    else if (vpl <= 8) LN_FWD(8);
    else if (vpl <= 16) LN_FWD(16);
    else return MOREC_E_UNSUPPORTED;
#undef LN_FWD
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_layernorm_fwd(const void* x, const float* bias, const void* res, const float* pos,
                                   int pos_period, const float* gamma, const float* beta, float eps, void* z_out,
                                   void* y, float* mean, float* rstd, int M, int N, int dtype, float p_in,
                                   uint64_t seed_in, float p_out, uint64_t seed_out, void* stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (p_in < 0.f || p_in >= 1.f || p_out < 0.f || p_out >= 1.f) return MOREC_E_ARG;
    const DropRng din = make_drop(p_in, seed_in), dout = make_drop(p_out, seed_out);
    if (N % 4 || (dtype == MOREC_BF16 && N % 4)) return MOREC_E_ALIGN;
    if (pos && pos_period <= 0) return MOREC_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        return ln_fwd_dispatch<float>(x, bias, res, pos, pos_period, gamma, beta, eps, z_out, y, mean, rstd, M, N, din, dout, s);
    if (dtype == MOREC_BF16)
        return ln_fwd_dispatch<bf16>(x, bias, res, pos, pos_period, gamma, beta, eps, z_out, y, mean, rstd, M, N, din, dout, s);
    return MOREC_E_DTYPE;
}

template <typename T>
static int ln_bwd_dispatch(const void* dy_a, const void* dy_b, const void* z, const float* mean, const float* rstd,
                           const float* gamma, void* dz, void* dzd, float* dgamma, float* dbeta, float* dbias, int M,
                           int N, DropRng din, DropRng dout, hipStream_t s) {
    const int vpl = (N + 255) / 256;
    const int rpb = 64;
    dim3 grid((M + rpb - 1) / rpb), block(256);
    const size_t lds = (dgamma || dbias) ? (size_t)12 * N * sizeof(float) : 0;
#define LN_BWD(V)                                                                                               \
    do {                                                                                                        \
        if (lds > 48 * 1024)                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<T, V>),                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
        hipLaunchKernelGGL((ln_bwd_kernel<T, V>), grid, block, lds, s, (const T*)dy_a, (const T*)dy_b,          \
                           (const T*)z, mean, rstd, gamma, (T*)dz, (T*)dzd, dgamma, dbeta, dbias, M, N, rpb, din, \
                           dout);                                                                               \
    } while (0)
    if (vpl <= 1) LN_BWD(1);
    else if (vpl <= 2) LN_BWD(2);
    else if (vpl <= 3) LN_BWD(3);
    else if (vpl <= 4) LN_BWD(4);
    else if (vpl <= 8) LN_BWD(8);
    else if (vpl <= 16) LN_BWD(16);
    else return MOREC_E_UNSUPPORTED;
#undef LN_BWD
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_layernorm_bwd(const void* dy_a, const void* dy_b, const void* z, const float* mean,
                                   const float* rstd, const float* gamma, void* dz, void* dzd, float* dgamma,
                                   float* dbeta, float* dbias, int M, int N, int dtype, float p_in, uint64_t seed_in,
                                   float p_out, uint64_t seed_out, void* stream) {
    if (!dy_a || !z || !mean || !rstd || !gamma || !dz || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (p_in < 0.f || p_in >= 1.f || p_out < 0.f || p_out >= 1.f) return MOREC_E_ARG;
    if ((p_in > 0.f) != (dzd != nullptr)) return MOREC_E_ARG;
    const DropRng din = make_drop(p_in, seed_in), dout = make_drop(p_out, seed_out);
    if ((dgamma == nullptr) != (dbeta == nullptr)) return MOREC_E_ARG;
    if (N % 4) return MOREC_E_ALIGN;
    if (N > 4096) return MOREC_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        return ln_bwd_dispatch<float>(dy_a, dy_b, z, mean, rstd, gamma, dz, dzd, dgamma, dbeta, dbias, M, N, din, dout, s);
    if (dtype == MOREC_BF16)
        return ln_bwd_dispatch<bf16>(dy_a, dy_b, z, mean, rstd, gamma, dz, dzd, dgamma, dbeta, dbias, M, N, din, dout, s);
    return MOREC_E_DTYPE;
}

// dpos[m % period, :] += dz[m, :]: block (p, chunk) sums rows m = p, p+period, ... of its chunk
template <typename T>
__global__ __launch_bounds__(256) void pos_grad_kernel(const T* __restrict__ dz, float* __restrict__ dpos, int M, int N,
                                                       int period, int seq_per_block) {
    const int p = blockIdx.x;
    const int s0 = blockIdx.y * seq_per_block;
    const int nseq = M / period;
    const int s1 = min(nseq, s0 + seq_per_block);
    for (int c = threadIdx.x; c < N; c += 256) {
        float acc = 0.f;
        for (int sq = s0; sq < s1; ++sq) acc += io<T>::load1(dz + ((size_t)sq * period + p) * N + c);
        atomicAdd(dpos + (size_t)p * N + c, acc);
    }
}

extern "C" int morec_pos_grad(const void* dz, float* dpos, int M, int N, int period, int dtype, void* stream) {
    if (!dz || !dpos || M <= 0 || N <= 0 || period <= 0 || M % period) return MOREC_E_ARG;
    const int spb = 64;
    dim3 grid(period, (M / period + spb - 1) / spb);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((pos_grad_kernel<float>), grid, dim3(256), 0, s, (const float*)dz, dpos, M, N, period, spb);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((pos_grad_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)dz, dpos, M, N, period, spb);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
