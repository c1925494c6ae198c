// gemm.hip -- NT GEMM with fused epilogues (bias, GELU/ReLU, activation-derivative, split-K atomics),
// plus the layout helpers the backward pass needs (transpose with conversion, cast, column sums).
// Reference arithmetic replaced: see include/morec_hip.h (morec_gemm_nt).
#include "gemm_nt_generic.hpp"


extern "C" int morec_gemm_nt(const morec_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                             void* aux_out, const void* dact_in, void* stream) {
    return morec_gemm_nt_colsum(d, A, B, C, bias, aux_out, dact_in, nullptr, nullptr, stream);
}

extern "C" size_t morec_gemm_colsum_workspace_bytes(int M, int N) {
    return (size_t)((M + 31) / 32) * (size_t)N * sizeof(float);      // one partial row per 32-row wave block at most (gemm_small.hip)
}

extern "C" int morec_gemm_nt_colsum(const morec_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                                    void* aux_out, const void* dact_in, float* colsum_out, float* workspace, void* stream) {
    if (!d || !A || !B || !C) return MOREC_E_ARG;
    if (colsum_out) {
        if (!workspace) return MOREC_E_ARG;
        if (d->dact == MOREC_ACT_NONE || d->accumulate != 0 || d->split_k > 1) return MOREC_E_UNSUPPORTED;
        const int os_ = elt_size(d->out_dtype);
        if ((d->N * os_) % 16 || (d->ldc * os_) % 16) return MOREC_E_UNSUPPORTED;     // needs the LDS-staged epilogues
    }
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return MOREC_E_ARG;
    const int es = elt_size(d->in_dtype), os = elt_size(d->out_dtype);
    if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (bias && !aligned16(bias))) return MOREC_E_ALIGN;
    if ((d->lda * es) % 16 || (d->ldb * es) % 16 || (d->K * es) % 16) return MOREC_E_ALIGN;
    if (d->N % 4 || (d->ldc * os) % (4 * os) || d->ldc % 4) return MOREC_E_ALIGN;
    if (d->split_k > 1 && d->accumulate != 2) return MOREC_E_ARG;
    if (d->accumulate == 2 && d->out_dtype != MOREC_F32) return MOREC_E_DTYPE;
    if (d->dact != MOREC_ACT_NONE && !dact_in) return MOREC_E_ARG;
    if (d->dact != MOREC_ACT_NONE && d->act != MOREC_ACT_NONE) return MOREC_E_ARG;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux_out = aux_out; a.dact_in = dact_in;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.act = d->act; a.dact = d->dact; a.accumulate = d->accumulate; a.alpha = d->alpha;
    a.aux_deriv = d->aux_deriv;
    if (d->aux_deriv && (d->act == MOREC_ACT_NONE || !aux_out)) return MOREC_E_ARG;
    a.colsum = colsum_out ? workspace : nullptr;
    a.colsum_dst = colsum_out;
    {   // epilogue form: per-wave LDS slices without workgroup barriers pay off where the epilogue is heavy (GELU + the second
        // output) or the tile row is short (N <= 1024); measured per shape with scripts/gemm_bench.py.  MOREC_GEMM_EPI=w|b forces one.
        static int we = -1;
        if (we < 0) { const char* e = getenv("MOREC_GEMM_EPI"); we = !e ? 2 : (e[0] == 'w' ? 1 : 0); }
        a.wave_epilogue = we == 2 ? ((aux_out != nullptr || d->N <= 1024) ? 1 : 0) : we;
    }
    a.vec_store = ((d->N * os) % 16 == 0) && ((d->ldc * os) % 16 == 0) && (!aux_out || aligned16(aux_out));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    {   // narrow outputs over very many rows (Swin stage 1 / 2): the streaming kernel (gemm_skinny.hip)
        const int rs = gemm_skinny_try_launch(d, a, s);
        if (rs != G8_NOT_TAKEN) return rs;
        const int rw = gemm_skinny_wide_try_launch(d, a, s);      // 288 < N <= 512, K <= 128 (stage-1 fc1 + GELU without a second output)
        if (rw != G8_NOT_TAKEN) return rw;
    }
    {   // epilogue-heavy 16-bit products: 256 x 128 tiles, two workgroups per CU (gemm2w.hip)
        const int r2 = gemm2w_try_launch(d, a, s);
        if (r2 != G8_NOT_TAKEN) return r2;
    }
    {   // bf16, large: the 256 x 256 eight-phase kernel (gemm8p.hip)
        const int r8 = gemm8p_try_launch(d, a, s);
        if (r8 != G8_NOT_TAKEN) return r8;
    }
    {   // latency-class 16-bit products (the SASRec layers): 64 x 64 tiles on a four-stage ring (gemm_small.hip)
        const int rs = gemm_small_try_launch(d, a, s);
        if (rs != G8_NOT_TAKEN) return rs;
    }
    if (d->in_dtype == MOREC_F32 && d->out_dtype == MOREC_F32) return launch_gemm<float, float>(d, a, s);
    if (d->in_dtype == MOREC_BF16 && d->out_dtype == MOREC_BF16) return launch_gemm<bf16, bf16>(d, a, s);
    if (d->in_dtype == MOREC_BF16 && d->out_dtype == MOREC_F32) return launch_gemm<bf16, float>(d, a, s);
    if (d->in_dtype == MOREC_F16) return gemm_nt_f16_launch(d, a, s);       // gemm_f16.hip
    return MOREC_E_DTYPE;
}

// ---------------------------------------------------------------------------------------------------
// transpose (+ conversion): 64 x 64 tiles through LDS, coalesced on both sides
// ---------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, TO* __restrict__ out, int R, int C,
                                                        int ld_in, int ld_out) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + i * 4, c = c0 + tx;
        tile[ty + i * 4][tx] = (r < R && c < C) ? io<TI>::load1(in + (size_t)r * ld_in + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + i * 4, r = r0 + tx;
        if (c < C && r < R) io<TO>::store1(out + (size_t)c * ld_out + r, tile[tx][ty + i * 4]);
    }
}

// vectorised variant (4 elements per global access); needs C % 4 == 0 and both pitches % 4 == 0
template <typename TI, typename TO>
__device__ __forceinline__ void transpose4_tile(const TI* __restrict__ in, TO* __restrict__ out, int R, int C, int ld_in, int ld_out,
                                                int r0, int c0, float (&tile)[64][65]) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 16, c = c0 + tx * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < R && c < C) io<TI>::load4(in + (size_t)r * ld_in + c, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[ty + i * 16][tx * 4 + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 16, r = r0 + tx * 4;
        if (c < C && r < R) {   // rows past R were zero-filled above, so a partial group writes zeros into the pad
            const float v[4] = {tile[tx * 4 + 0][ty + i * 16], tile[tx * 4 + 1][ty + i * 16], tile[tx * 4 + 2][ty + i * 16],
                                tile[tx * 4 + 3][ty + i * 16]};
            io<TO>::store4(out + (size_t)c * ld_out + r, v);
        }
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose4_kernel(const TI* __restrict__ in, TO* __restrict__ out, int R, int C,
                                                         int ld_in, int ld_out) {
    __shared__ float tile[64][65];
    transpose4_tile<TI, TO>(in, out, R, C, ld_in, ld_out, blockIdx.y * 64, blockIdx.x * 64, tile);
}

// Many matrices in one launch (the Linear weights' W^T copies of a step: 60 launches of 5-6 us each before).  Block b belongs
// to the item whose [tile0, tile0 + tiles) range holds it; the table lives in device memory and is built once by the caller.
template <typename T>
__global__ __launch_bounds__(256) void transpose4_batch_kernel(const morec_transpose_item* __restrict__ items, int n_items) {
    __shared__ float tile[64][65];
    const int b = blockIdx.x;
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {        // last item with tile0 <= b (wave-uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const morec_transpose_item it = items[lo];
    const int t = b - it.tile0, tiles_x = (it.cols + 63) / 64;
    transpose4_tile<T, T>(reinterpret_cast<const T*>(it.src), reinterpret_cast<T*>(it.dst), it.rows, it.cols, it.ld_src, it.ld_dst,
                          (t / tiles_x) * 64, (t % tiles_x) * 64, tile);
}

extern "C" int morec_transpose_batch(const morec_transpose_item* items, int n_items, int n_tiles, int dtype, void* stream) {
    if (!items || n_items <= 0 || n_tiles <= 0) return MOREC_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((transpose4_batch_kernel<T>), dim3(n_tiles), dim3(256), 0, s, items, n_items);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_transpose(const void* in, void* out, int R, int C, int ld_in, int ld_out, int in_dtype,
                               int out_dtype, void* stream) {
    if (!in || !out || R <= 0 || C <= 0) return MOREC_E_ARG;
    dim3 grid((C + 63) / 64, (R + 63) / 64);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool vec = (C % 4 == 0) && (ld_in % 4 == 0) && (ld_out % 4 == 0) && (ld_out >= ((R + 3) & ~3)) &&
                     ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    // conversions offered: within a type, fp32 -> 16-bit (weight shadows) and 16-bit -> fp32; not bf16 <-> fp16
    if (in_dtype != out_dtype && in_dtype != MOREC_F32 && out_dtype != MOREC_F32) return MOREC_E_DTYPE;
    bool ok = true;
    if (!by_dtype(in_dtype, [&](auto* ti) {
            using TI = MOREC_TAG_T(ti);
            ok = by_dtype(out_dtype, [&](auto* to) {
                using TO = MOREC_TAG_T(to);
                if constexpr (std::is_same<TI, TO>::value || std::is_same<TI, float>::value || std::is_same<TO, float>::value) {
                    if (vec) hipLaunchKernelGGL((transpose4_kernel<TI, TO>), grid, dim3(256), 0, s, (const TI*)in, (TO*)out, R, C, ld_in, ld_out);
                    else hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, dim3(256), 0, s, (const TI*)in, (TO*)out, R, C, ld_in, ld_out);
                }
            });
        }) || !ok)
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float v[4];
        io<TI>::load4(in + i * 4, v);
        io<TO>::store4(out + i * 4, v);
    }
}

// out[r] = [ hi(in[r]) | s1 | s2 ] (bf16, 3 C columns): hi = bf16(x), lo = bf16(x - hi); slot lo_slot (1 or 2) holds lo, the other one
// hi again.  An NT product of an A-side row block (lo_slot 2: hi | hi | lo) with a B-side one (lo_slot 1: hi | lo | hi) over 3 C is
// hi.hi + hi.lo + lo.hi in the MFMA's fp32 accumulators: the fp32 product up to the lo.lo term (2^-16 relative) -- three bf16 MFMA
// passes instead of the 1/16-rate exact-fp32 MFMA.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ in, bf16* __restrict__ out, int R, int Cc, int ld_in,
                                                           int ld_out, int lo_slot) {
    const int c4 = Cc / 4;
    const size_t total = (size_t)R * c4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4), c = (int)(i - (size_t)r * c4) * 4;
        const float4 x = *reinterpret_cast<const float4*>(in + (size_t)r * ld_in + c);
        const float v[4] = {x.x, x.y, x.z, x.w};
        float hi[4], lo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hi[k] = bf2f(f2bf(v[k]));
            lo[k] = v[k] - hi[k];
        }
        bf16* o = out + (size_t)r * ld_out + c;
        io<bf16>::store4(o, hi);
        io<bf16>::store4(o + (size_t)(lo_slot == 1 ? 2 : 1) * Cc, hi);
        io<bf16>::store4(o + (size_t)lo_slot * Cc, lo);
    }
}

extern "C" int morec_split_bf16x3(const float* in, void* out, int R, int C, int ld_in, int ld_out, int lo_slot, void* stream) {
    if (!in || !out || R <= 0 || C <= 0 || (lo_slot != 1 && lo_slot != 2)) return MOREC_E_ARG;
    if (C % 4 || ld_in % 4 || ld_out % 4 || ld_in < C || ld_out < 3 * C || !aligned16(in) || (reinterpret_cast<uintptr_t>(out) & 7u)) return MOREC_E_ALIGN;
    const size_t total = (size_t)R * (C / 4);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, reinterpret_cast<bf16*>(out), R, C,
                       ld_in, ld_out, lo_slot);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_cast(const void* in, void* out, size_t n, int in_dtype, int out_dtype, void* stream) {
    if (!in || !out) return MOREC_E_ARG;
    if (n == 0) return MOREC_OK;
    if (n % 4 || !aligned16(in) || (reinterpret_cast<uintptr_t>(out) & 7u)) return MOREC_E_ALIGN;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bool ok = true;
    if (!by_dtype(in_dtype, [&](auto* ti) {
            using TI = MOREC_TAG_T(ti);
            ok = by_dtype(out_dtype, [&](auto* to) {
                using TO = MOREC_TAG_T(to);
                hipLaunchKernelGGL((cast_kernel<TI, TO>), dim3(blocks), dim3(256), 0, s, (const TI*)in, (TO*)out, n4);
            });
        }) || !ok)
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// column sums: each block reduces a [rows_per_block x 64-column] slab with 4-element vector loads (16 column
// groups x 16 row lanes), folds the row lanes through LDS and leaves ONE atomicAdd per column per block
// det (deterministic mode): out is a [gridDim.y][N] partial buffer written with plain stores; the launcher folds the rows in order
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, float* __restrict__ out, int M, int N,
                                                     int ld, int rows_per_block, int det = 0) {
    __shared__ float part[16][65];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int n = blockIdx.x * 64 + cg * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 16) {
            float v[4];
            io<T>::load4(in + (size_t)r * ld + n, v);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) part[rl][cg * 4 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += part[r][threadIdx.x];
            if (det) out[(size_t)blockIdx.y * N + c] = t;
            else atomicAdd(out + c, t);
        }
    }
}

int colsum_f32_launch(const float* in, float* out, int rows, int N, hipStream_t s) {
    // folds of per-tile / per-sequence partial rows: a few MB spread over >= 512 blocks (with 512 rows per block the
    // [2560 x 2304] attention-bias partials ran on 180 blocks of 32 dependent loads per lane: 16 us for 24 MB)
    const int cb = (N + 63) / 64;
    int rpb = (int)(((long)rows * cb / 512 + 15) & ~15L);
    rpb = rpb < 16 ? 16 : rpb > 512 ? 512 : rpb;
    if ((long)rows * N <= (1L << 20) && rows <= 512) rpb = 512;     // small folds: one block per column group, i.e. a fixed summation order
    dim3 grid(cb, (rows + rpb - 1) / rpb);
    if (grid.y > 1 && morec_deterministic()) {      // several row blocks per column: their partials through the scratch, folded in order
        float* part = morec_det_scratch(s, (size_t)grid.y * N);
        if (!part) return (int)hipErrorOutOfMemory;
        hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, in, part, rows, N, N, rpb, 1);
        MOREC_CHECK_LAUNCH();
        return morec_det_fold_add(part, out, (int)grid.y, (size_t)N, (size_t)N, s);
    }
    hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, in, out, rows, N, N, rpb);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_colsum(const void* in, float* out, int M, int N, int ld, int dtype, void* stream) {
    if (!in || !out || M <= 0 || N <= 0) return MOREC_E_ARG;
    if (N % 4 || ld % 4) return MOREC_E_ALIGN;
    const int rpb = 512;
    dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* part = nullptr;
    if (grid.y > 1 && morec_deterministic()) {
        part = morec_det_scratch(s, (size_t)grid.y * N);
        if (!part) return (int)hipErrorOutOfMemory;
    }
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, s, (const T*)in, part ? part : out, M, N, ld, rpb, part ? 1 : 0);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    if (part) return morec_det_fold_add(part, out, (int)grid.y, (size_t)N, (size_t)N, s);
    return MOREC_OK;
}

// out = dy * act'(pre)  (backward through the GELU of Text_Encoder, T/model/encoders.py:70, where the
// activation follows the LAST projection and its derivative cannot ride on a GEMM epilogue)
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ out, size_t n4,
                               int act) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float d[4], u[4];
        io<T>::load4(dy + i * 4, d);
        io<T>::load4(pre + i * 4, u);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = (act == MOREC_ACT_GELU) ? d[k] * dgelu_f(u[k]) : (u[k] > 0.f ? d[k] : 0.f);
        io<T>::store4(out + i * 4, d);
    }
}

extern "C" int morec_act_bwd(const void* dy, const void* pre, void* out, size_t n, int act, int dtype, void* stream) {
    if (!dy || !pre || !out) return MOREC_E_ARG;
    if (n == 0) return MOREC_OK;
    if (n % 4) return MOREC_E_ALIGN;
    if (act != MOREC_ACT_GELU && act != MOREC_ACT_RELU) return MOREC_E_ARG;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)dy, (const T*)pre, (T*)out, n4, act);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// out = scale * (x0 + x1 + x2), elementwise with fp32 arithmetic (x1 / x2 may be null): the mean over a news item's text attributes
// (T/model/encoders.py:113-116: torch.mean(torch.stack(text_vectors, dim=1), dim=1)) and, with one input, its backward (d / k).
template <typename T>
__global__ void scaled_sum_kernel(const T* __restrict__ x0, const T* __restrict__ x1, const T* __restrict__ x2, T* __restrict__ out,
                                  size_t n4, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float a[4], b[4];
        io<T>::load4(x0 + i * 4, a);
        if (x1) {
            io<T>::load4(x1 + i * 4, b);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += b[k];
        }
        if (x2) {
            io<T>::load4(x2 + i * 4, b);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += b[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] *= scale;
        io<T>::store4(out + i * 4, a);
    }
}

extern "C" int morec_scaled_sum(const void* x0, const void* x1, const void* x2, void* out, size_t n, float scale, int dtype, void* stream) {
    if (!x0 || !out || (x2 && !x1)) return MOREC_E_ARG;
    if (n == 0) return MOREC_OK;
    if (n % 4) return MOREC_E_ALIGN;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((scaled_sum_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)x0, (const T*)x1, (const T*)x2, (T*)out, n4, scale);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
