// layernorm.hip -- LayerNorm forward/backward with the pre-add fused in (bias + residual + position
// rows), one wavefront per row, row held in registers, wave64 shuffles for the two reductions.
// Reference arithmetic: T/model/modules.py:17,63,93 (eps 1e-6) and HF BertSelfOutput / BertOutput /
// BertEmbeddings LayerNorm (eps 1e-12); HBM-bound (one read of each input, one write of each output).
#include <algorithm>
#include "common.hpp"
#include "pos_grad.hpp"

// Every lane moves 16 bytes per access (EV = 4 fp32 / 8 bf16 elements); VPL = such vectors per lane, a row of N
// elements needs ceil(N / (64 EV)) of them.  (With 8-byte bf16 accesses the kernels were memory-INSTRUCTION bound:
// same launch time for bf16 and fp32 rows.)
// LPR = lanes per row: narrow rows (Swin stages: N = 96 ... 256) put 4 or 2 rows in one wavefront instead of idling 3/4 of it
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// FULL: N == VPL * LPR * EV exactly -- no column bounds checks, so the loads of a row are issued back to back
template <typename T, int VPL, int LPR = 64, bool FULL = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                     const T* __restrict__ res, const float* __restrict__ pos,
                                                     int pos_period, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ z_out,
                                                     T* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int M, int N, DropRng din,
                                                     DropRng dout, const float* __restrict__ rowscale, int rps) {
    din = drop_resolve(din);
    dout = drop_resolve(dout);
    constexpr int EV = vio<T>::EV;
    constexpr int GRP = 64 / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * GRP + ((threadIdx.x & 63) / LPR);
    if (row >= M) return;
    const size_t base = (size_t)row * N;
    const float rsc = rowscale ? rowscale[row / rps] : 1.0f;   // DropPath: per-sample scale of the sub-layer branch
    // every row-sized load first, from unguarded (clamped) addresses, then the arithmetic: see the note in ln_bwd_kernel
    uint4 rx[VPL], rr[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
        rx[i] = vio<T>::load_raw(x + base + cl);
    }
    if (res) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
            rr[i] = vio<T>::load_raw(res + base + cl);
        }
    }
    float v[VPL][EV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * LPR + lane) * EV;
        if (FULL || c < N) {
            vio<T>::unpack(rx[i], v[i]);
            if (bias) {
                float b[EV];
                load_f32v<EV>(bias + c, b);
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] += b[k];
            }
            if (rowscale) {
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] *= rsc;
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
                float r[EV];
                vio<T>::unpack(rr[i], r);
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] += r[k];
            }
            if (pos) {
                float b[EV];
                load_f32v<EV>(pos + (size_t)(row % pos_period) * N + c, b);
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] += b[k];
            }
            if (z_out) {
                vio<T>::store(z_out + base + c, v[i]);
                // the backward pass re-reads z at storage precision: normalise exactly what was stored
#pragma unroll
                for (int k = 0; k < EV; ++k) v[i][k] = io<T>::round(v[i][k]);
            }
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
            if (dout.thresh) {  // dropout on the LayerNorm output (embedding stages: modules.py:93-94, HF BertEmbeddings)
#pragma unroll
                for (int k = 0; k < EV; ++k) o[k] *= dout.inv_keep;
                bool kp[EV];
                drop_keep_vec<EV>(dout, base + c, kp);
#pragma unroll
                for (int k = 0; k < EV; ++k) o[k] = kp[k] ? o[k] : 0.f;
            }
            vio<T>::store(y + base + c, o);
        }
    }
}

// Backward.  Each block owns RPB consecutive rows, one row per wave per trip.  Per-column dgamma / dbeta (and the
// sub-layer bias gradient dbias = column sums of dzd) partials are reduced across the block's waves in LDS and leave
// as ONE atomicAdd per column per block.
template <typename T, int VPL, int LPR = 64, bool FULL = false>
__global__ __launch_bounds__(256, (VPL * vio<T>::EV <= 24 ? 2 : 1)) void ln_bwd_kernel(const T* __restrict__ dy_a, const T* __restrict__ dy_b,
                                                     const T* __restrict__ z, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     T* __restrict__ dz, T* __restrict__ dzd, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dbias, int M, int N,
                                                     int rows_per_block, DropRng din, DropRng dout,
                                                     const T* __restrict__ dres, const float* __restrict__ rowscale, int rps,
                                                     float* __restrict__ det) {
    din = drop_resolve(din);
    dout = drop_resolve(dout);
    constexpr int EV = vio<T>::EV;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int GRP = 64 / LPR, NP = 4 * GRP;     // row groups per wave, column-partial sets per block
    float* sgm = reinterpret_cast<float*>(smem_raw); // [N] gamma (re-read per row: 24 registers cheaper than holding it)
    float* sg = sgm + N;                             // [NP][N] dgamma partials
    float* sb = sg + NP * (size_t)N;                // [NP][N] dbeta partials
    float* sd = sb + NP * (size_t)N;                // [NP][N] dbias partials (only when dbias)
    const int lane = threadIdx.x & (LPR - 1), wave = (threadIdx.x >> 6) * GRP + ((threadIdx.x & 63) / LPR);
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
    // Row loads go through buffer descriptors based at the block's first row: every load of a row is issued back to back
    // and unguarded (a load inside `if (c < N)` / `if (dy_b)` next to its use is its own basic block ending in
    // s_waitcnt vmcnt(0): one memory round trip per vector per operand instead of one per row); a row past the block's
    // last one, or an absent dy_b (zero-extent descriptor), is an out-of-range offset: zeros, no memory traffic.
    // The NEXT row's loads are issued as soon as this row's raw vectors are unpacked -- into the same registers, ahead of the
    // two row reductions, the output arithmetic and the stores -- so the HBM round trip overlaps them.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const int blk_bytes = (r1 - r0) * N * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(dy_a + (size_t)r0 * N), 0, blk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc((void*)(z + (size_t)r0 * N), 0, blk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)((dy_b ? dy_b : dy_a) + (size_t)r0 * N), 0, dy_b ? blk_bytes : 0, 0x00020000);
    uint4 ra[VPL], rz[VPL], rb[VPL];
    auto as_uint4 = [](u32x4_t v) { return make_uint4(v[0], v[1], v[2], v[3]); };
    auto fetch_row = [&](int row) {
        const uint32_t ro = row < r1 ? (uint32_t)((row - r0) * N) * (uint32_t)sizeof(T) : 0x80000000u;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
            ra[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rA, ro + cl * (int)sizeof(T), 0, 0));
            rz[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rZ, ro + cl * (int)sizeof(T), 0, 0));
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
            rb[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rB, ro + cl * (int)sizeof(T), 0, 0));
        }
    };
    fetch_row(r0 + wave);
    for (int row = r0 + wave; row < r1; row += NP) {
        const size_t base = (size_t)row * N;
        const float mu = mean[row], rs = rstd[row];
        const float rsc = (dzd && rowscale) ? rowscale[row / rps] : 1.0f;
        float g[VPL][EV], xh[VPL][EV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV;
            if (FULL || c < N) {
                float d[EV], zz[EV];
                vio<T>::unpack(ra[i], d);
                {
                    float e[EV];
                    vio<T>::unpack(rb[i], e);      // zeros without dy_b
#pragma unroll
                    for (int k = 0; k < EV; ++k) d[k] += e[k];
                }
                if (dout.thresh) {
#pragma unroll
                    for (int k = 0; k < EV; ++k) d[k] *= dout.inv_keep;
                    bool kp[EV];
                    drop_keep_vec<EV>(dout, base + c, kp);
#pragma unroll
                    for (int k = 0; k < EV; ++k) d[k] = kp[k] ? d[k] : 0.f;
                }
                vio<T>::unpack(rz[i], zz);
                float gmv[EV];
                load_f32v<EV>(sgm + c, gmv);
#pragma unroll
                for (int k = 0; k < EV; ++k) {
                    xh[i][k] = (zz[k] - mu) * rs;
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
            // one vector at a time: without the `c < N` blocks the whole row is one basic block and the scheduler interleaves every
            // vector's arithmetic (dropout masks included) -- 38 spilled registers at VPL 3
            if (FULL) __builtin_amdgcn_sched_barrier(0);
        }
        fetch_row(row + NP);
        uint4 rd[VPL];
        if (dres) {      // pre-LN blocks only; requested together, in flight during the two row reductions
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = (i * LPR + lane) * EV, cl = FULL ? c : min(c, N - EV);
                rd[i] = vio<T>::load_raw(dres + base + cl);
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
                if (dres) {   // pre-LN blocks: the residual stream's own gradient joins the LayerNorm-input gradient
                    float e[EV];
                    vio<T>::unpack(rd[i], e);
#pragma unroll
                    for (int k = 0; k < EV; ++k) o[k] += e[k];
                }
                vio<T>::store(dz + base + c, o);
                if (dzd) {   // gradient w.r.t. the dropped-out sub-layer output (the residual branch takes dz itself)
#pragma unroll
                    for (int k = 0; k < EV; ++k) o[k] *= din.inv_keep * rsc;
                    if (din.thresh) {
                        bool kp[EV];
                        drop_keep_vec<EV>(din, base + c, kp);
#pragma unroll
                        for (int k = 0; k < EV; ++k) o[k] = kp[k] ? o[k] : 0.f;
                    }
                    vio<T>::store(dzd + base + c, o);
                }
                if (dbias) {
#pragma unroll
                    for (int k = 0; k < EV; ++k) ad[i][k] += io<T>::round(o[k]);   // what colsum over the stored tensor would see
                }
            }
            if (FULL) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (dgamma || dbias) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * LPR + lane) * EV;
            if (FULL || c < N) {
                store_f32v<EV>(sg + (size_t)wave * N + c, ag[i]);
                store_f32v<EV>(sb + (size_t)wave * N + c, ab[i]);
                if (dbias) store_f32v<EV>(sd + (size_t)wave * N + c, ad[i]);
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
            if (det) {      // deterministic mode: this block's [3][N] partial row; the launcher folds the blocks in order
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

template <typename T>
static int ln_fwd_dispatch(const void* x, const float* bias, const void* res, const float* pos, int pos_period,
                           const float* gamma, const float* beta, float eps, void* z_out, void* y, float* mean,
                           float* rstd, int M, int N, DropRng din, DropRng dout, const float* rowscale, int rps,
                           hipStream_t s) {
    if (N % vio<T>::EV) return MOREC_E_ALIGN;
    const int vpl = (N + 64 * vio<T>::EV - 1) / (64 * vio<T>::EV);
    dim3 grid((M + 3) / 4), block(256);
#define LN_FWD_L(V, L)                                                                                                 \
    hipLaunchKernelGGL((ln_fwd_kernel<T, V, L>), dim3((M + 4 * (64 / L) - 1) / (4 * (64 / L))), block, 0, s, (const T*)x, \
                       bias, (const T*)res, pos, pos_period, gamma, beta, eps, (T*)z_out, (T*)y, mean, rstd, M, N, din,  \
                       dout, rowscale, rps)
#define LN_FWD(V) LN_FWD_L(V, 64)
    if (N <= 16 * vio<T>::EV) LN_FWD_L(1, 16);
    else if (N <= 32 * vio<T>::EV) LN_FWD_L(1, 32);
    else if (N == 96 * vio<T>::EV)   // H = 768 in bf16: 96 vectors = 32 lanes x 3, two rows per wave, no idle lanes, no bounds checks
        hipLaunchKernelGGL((ln_fwd_kernel<T, 3, 32, true>), dim3((M + 7) / 8), block, 0, s, (const T*)x, bias, (const T*)res, pos, pos_period,
                           gamma, beta, eps, (T*)z_out, (T*)y, mean, rstd, M, N, din, dout, rowscale, rps);
    else if (vpl <= 1) LN_FWD(1);
    else if (vpl <= 2) LN_FWD(2);
    else if (vpl <= 3) LN_FWD(3);
    else if (vpl <= 4) LN_FWD(4);
    else if (vpl <= 8) LN_FWD(8);
    else if (vpl <= 16) LN_FWD(16);
    else return MOREC_E_UNSUPPORTED;
#undef LN_FWD
#undef LN_FWD_L
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_layernorm_fwd(const void* x, const float* bias, const void* res, const float* pos,
                                   int pos_period, const float* gamma, const float* beta, float eps, void* z_out,
                                   void* y, float* mean, float* rstd, int M, int N, int dtype, float p_in,
                                   uint64_t seed_in, float p_out, uint64_t seed_out, const float* rowscale,
                                   int rows_per_scale, void* stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (rowscale && rows_per_scale <= 0) return MOREC_E_ARG;
    if (p_in < 0.f || p_in >= 1.f || p_out < 0.f || p_out >= 1.f) return MOREC_E_ARG;
    const DropRng din = make_drop(p_in, seed_in), dout = make_drop(p_out, seed_out);
    if (N % 4 || (dtype == MOREC_BF16 && N % 4)) return MOREC_E_ALIGN;
    if (pos && pos_period <= 0) return MOREC_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        return ln_fwd_dispatch<float>(x, bias, res, pos, pos_period, gamma, beta, eps, z_out, y, mean, rstd, M, N, din, dout, rowscale, rows_per_scale, s);
    if (dtype == MOREC_BF16)
        return ln_fwd_dispatch<bf16>(x, bias, res, pos, pos_period, gamma, beta, eps, z_out, y, mean, rstd, M, N, din, dout, rowscale, rows_per_scale, s);
    if (dtype == MOREC_F16)
        return ln_fwd_dispatch<f16>(x, bias, res, pos, pos_period, gamma, beta, eps, z_out, y, mean, rstd, M, N, din, dout, rowscale, rows_per_scale, s);
    return MOREC_E_DTYPE;
}

// One instantiation's launch.  Rows per block: ONE round of blocks at the occupancy this variant really gets (LDS partial sets and
// registers: 2 blocks per CU at H = 768, 4 at the Swin stage widths -- a grid sized for 2 left half of the wave slots of the
// C = 96 launch empty), never fewer than 64 rows so that the per-block column flush (3 N atomics) stays small.
// (64-row blocks at M = 51200, H = 768: 800 blocks = 1.56 rounds, 12 % slower than one round.)
template <typename T, int V, int L, bool FULL>
static int ln_bwd_launch(const void* dy_a, const void* dy_b, const void* z, const float* mean, const float* rstd, const float* gamma,
                          void* dz, void* dzd, float* dgamma, float* dbeta, float* dbias, int M, int N, DropRng din, DropRng dout,
                          const void* dres, const float* rowscale, int rps, hipStream_t s) {
    const size_t lds = ((dgamma || dbias) ? (size_t)12 * (64 / L) * N : 0) * sizeof(float) + N * sizeof(float);
    static size_t lds_seen = ~(size_t)0;
    static int slots = 0;
    if (lds != lds_seen) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<T, V, L, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, n_cu = 0, nb = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ln_bwd_kernel<T, V, L, FULL>, 256, lds) != hipSuccess || nb <= 0) {
            (void)hipGetLastError();
            nb = 2;
        }
        slots = nb * n_cu;
        lds_seen = lds;
    }
    // (floor of rows per block: 64 keeps the per-block column atomics rare on the encoders' 50 k rows; on the SASRec layers' 2 560 rows it
    // left 40 blocks whose four waves walked 16 rows each, one memory round trip per row: 18 - 24 us per launch -- 16 rows there)
    const int rpb = std::max(M <= 8192 ? 16 : 64, (((M + slots - 1) / slots) + 15) & ~15);
    dim3 grid((M + rpb - 1) / rpb), block(256);
    float* det = nullptr;
    if ((dgamma || dbias) && morec_deterministic()) {      // per-block partial rows instead of one atomic per column per block
        det = morec_det_scratch(s, (size_t)grid.x * 3 * N);
        if (!det) return (int)hipErrorOutOfMemory;
    }
    hipLaunchKernelGGL((ln_bwd_kernel<T, V, L, FULL>), grid, block, lds, s, (const T*)dy_a, (const T*)dy_b, (const T*)z, mean, rstd, gamma,
                       (T*)dz, (T*)dzd, dgamma, dbeta, dbias, M, N, rpb, din, dout, (const T*)dres, rowscale, rps, det);
    if (det) {
        MOREC_CHECK_LAUNCH();
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

template <typename T>
static int ln_bwd_dispatch(const void* dy_a, const void* dy_b, const void* z, const float* mean, const float* rstd,
                           const float* gamma, void* dz, void* dzd, float* dgamma, float* dbeta, float* dbias, int M,
                           int N, DropRng din, DropRng dout, const void* dres, const float* rowscale, int rps, hipStream_t s) {
    if (N % vio<T>::EV) return MOREC_E_ALIGN;
    const int vpl = (N + 64 * vio<T>::EV - 1) / (64 * vio<T>::EV);
#define LN_BWD_L(V, L, F) rc = ln_bwd_launch<T, V, L, F>(dy_a, dy_b, z, mean, rstd, gamma, dz, dzd, dgamma, dbeta, dbias, M, N, din, dout, dres, rowscale, rps, s)
    int rc = MOREC_OK;
    if (N <= 16 * vio<T>::EV) LN_BWD_L(1, 16, false);
    else if (N <= 32 * vio<T>::EV) LN_BWD_L(1, 32, false);
    else if (N == 96 * vio<T>::EV) LN_BWD_L(3, 32, true);
    else if (vpl <= 1) LN_BWD_L(1, 64, false);
    else if (vpl <= 2) LN_BWD_L(2, 64, false);
    else if (vpl <= 3) LN_BWD_L(3, 64, false);
    else if (vpl <= 4) LN_BWD_L(4, 64, false);
    else if (vpl <= 8) LN_BWD_L(8, 64, false);
    else if (vpl <= 16) LN_BWD_L(16, 64, false);
    else return MOREC_E_UNSUPPORTED;
#undef LN_BWD_L
    if (rc != MOREC_OK) return rc;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_layernorm_bwd(const void* dy_a, const void* dy_b, const void* z, const float* mean,
                                   const float* rstd, const float* gamma, void* dz, void* dzd, float* dgamma,
                                   float* dbeta, float* dbias, int M, int N, int dtype, float p_in, uint64_t seed_in,
                                   float p_out, uint64_t seed_out, const void* dres, const float* rowscale,
                                   int rows_per_scale, void* stream) {
    if (!dy_a || !z || !mean || !rstd || !gamma || !dz || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (p_in < 0.f || p_in >= 1.f || p_out < 0.f || p_out >= 1.f) return MOREC_E_ARG;
    if ((p_in > 0.f || rowscale != nullptr) != (dzd != nullptr)) return MOREC_E_ARG;
    if (rowscale && rows_per_scale <= 0) return MOREC_E_ARG;
    const DropRng din = make_drop(p_in, seed_in), dout = make_drop(p_out, seed_out);
    if ((dgamma == nullptr) != (dbeta == nullptr)) return MOREC_E_ARG;
    if (N % 4) return MOREC_E_ALIGN;
    if (N > 4096) return MOREC_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MOREC_F32)
        return ln_bwd_dispatch<float>(dy_a, dy_b, z, mean, rstd, gamma, dz, dzd, dgamma, dbeta, dbias, M, N, din, dout, dres, rowscale, rows_per_scale, s);
    if (dtype == MOREC_BF16)
        return ln_bwd_dispatch<bf16>(dy_a, dy_b, z, mean, rstd, gamma, dz, dzd, dgamma, dbeta, dbias, M, N, din, dout, dres, rowscale, rows_per_scale, s);
    if (dtype == MOREC_F16)
        return ln_bwd_dispatch<f16>(dy_a, dy_b, z, mean, rstd, gamma, dz, dzd, dgamma, dbeta, dbias, M, N, din, dout, dres, rowscale, rows_per_scale, s);
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
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // 16-byte column vectors, several rows in flight (pos_grad.hpp); the scalar kernel keeps the row widths that do not fit
    int rc_pos = 1;
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            rc_pos = pos_type_grad_launch<T>((const T*)dz, dpos, nullptr, M / period, period, N, s);
            if (rc_pos == 0) {      // row width outside the vector layout: the scalar kernel; deterministic mode = ONE block per position (single writer)
                const int spb_ = morec_deterministic() ? M / period : spb;
                hipLaunchKernelGGL((pos_grad_kernel<T>), dim3(period, (M / period + spb_ - 1) / spb_), dim3(256), 0, s, (const T*)dz, dpos, M, N, period, spb_);
            }
        }))
        return MOREC_E_DTYPE;
    if (rc_pos < 0) return (int)hipErrorOutOfMemory;      // deterministic scratch unavailable (it may not grow under graph capture)
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
