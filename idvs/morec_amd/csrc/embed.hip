// embed.hip -- embedding gathers / scatters (HBM-bound integer-indexed row traffic).
//   BERT embeddings (HF BertEmbeddings): word[ids] + position[t] + token_type[0] -> LayerNorm, fused;
//   backward scatter-add into the three tables (word row pad_id skipped: nn.Embedding padding_idx);
//   ID tower (T/model/model.py:27-28,37): row gather / scatter-add with padding_idx = 0;
//   strided row copies for hidden[:, 0] (T/model/encoders.py:69).
#include "common.hpp"
#include "pos_grad.hpp"

template <typename T, int VPL>
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(const int32_t* __restrict__ ids,
                                                             const float* __restrict__ word,
                                                             const float* __restrict__ pos,
                                                             const float* __restrict__ type0,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             T* __restrict__ z_out, T* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             int M, int Tlen, int H, DropRng dout) {
    dout = drop_resolve(dout);
    constexpr int EV = vio<T>::EV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t base = (size_t)row * H;
    const float* w = word + (size_t)ids[row] * H;
    const float* p = pos + (size_t)(row % Tlen) * H;
    float v[VPL][EV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * EV;
        if (c < H) {
            float a[EV], b[EV], t[EV];
            load_f32v<EV>(w + c, a);
            load_f32v<EV>(p + c, b);
            load_f32v<EV>(type0 + c, t);
            // HF order: (inputs_embeds + token_type_embeddings) + position_embeddings
#pragma unroll
            for (int k = 0; k < EV; ++k) v[i][k] = (a[k] + t[k]) + b[k];
            if (z_out) {
                vio<T>::store(z_out + base + c, v[i]);
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
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * EV;
        if (c < H) {
#pragma unroll
            for (int k = 0; k < EV; ++k) {
                const float d = v[i][k] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * EV;
        if (c < H) {
            float g[EV], b[EV], o[EV];
            load_f32v<EV>(gamma + c, g);
            load_f32v<EV>(beta + c, b);
#pragma unroll
            for (int k = 0; k < EV; ++k) o[k] = (v[i][k] - mean) * rstd * g[k] + b[k];
            if (dout.thresh) {   // HF BertEmbeddings: dropout after the LayerNorm
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

extern "C" int morec_bert_embed_fwd(const int32_t* ids, const float* word, const float* pos, const float* type0,
                                    const float* gamma, const float* beta, float eps, void* z_out, void* y,
                                    float* mean, float* rstd, int M, int T, int H, int dtype, float p_out,
                                    uint64_t seed_out, void* stream) {
    if (p_out < 0.f || p_out >= 1.f) return MOREC_E_ARG;
    const DropRng dout = make_drop(p_out, seed_out);
    if (!ids || !word || !pos || !type0 || !gamma || !beta || !y || M <= 0 || T <= 0 || H <= 0) return MOREC_E_ARG;
    const int ev = dtype == MOREC_F32 ? 4 : 8;
    if (H % ev) return MOREC_E_ALIGN;
    const int vpl = (H + 64 * ev - 1) / (64 * ev);
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define EMB(TT, V)                                                                                                  \
    hipLaunchKernelGGL((bert_embed_fwd_kernel<TT, V>), grid, block, 0, s, ids, word, pos, type0, gamma, beta, eps, \
                       (TT*)z_out, (TT*)y, mean, rstd, M, T, H, dout)
#define EMB_DISPATCH(TT)                      \
    do {                                      \
        if (vpl <= 1) EMB(TT, 1);             \
        else if (vpl <= 2) EMB(TT, 2);        \
        else if (vpl <= 3) EMB(TT, 3);        \
        else if (vpl <= 4) EMB(TT, 4);        \
        else if (vpl <= 8) EMB(TT, 8);        \
        else return MOREC_E_UNSUPPORTED;      \
    } while (0)
    if (dtype == MOREC_F32) EMB_DISPATCH(float);
    else if (dtype == MOREC_BF16) EMB_DISPATCH(bf16);
    else if (dtype == MOREC_F16) EMB_DISPATCH(f16);
    else return MOREC_E_DTYPE;
#undef EMB
#undef EMB_DISPATCH
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// word-table scatter: one wave per token row, fp32 atomics (duplicate tokens collide in L2)
template <typename T>
__global__ __launch_bounds__(256) void word_scatter_kernel(const int32_t* __restrict__ ids, const T* __restrict__ dz,
                                                           float* __restrict__ dword, int pad_id, int M, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int id = ids[row];
    if (id == pad_id) return;
    float* dst = dword + (size_t)id * H;
    for (int c = lane * 4; c < H; c += 256) {
        float v[4];
        io<T>::load4(dz + (size_t)row * H + c, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(dst + c + k, v[k]);
    }
}

// Same scatter over rows visited in token-id order (`order` = argsort of ids): equal ids are adjacent, so a wave sums a run of
// rows in registers and issues ONE atomic per element per run instead of one per row -- the [CLS] / [SEP] rows that every title
// contributes to (2688 colliding rows at B = 128) made the plain scatter 0.76 ms per step.
template <typename T, int CPL>   // CPL = float4 column groups per lane: H <= 256 * CPL
__global__ __launch_bounds__(256) void word_scatter_sorted_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ order,
                                                                  const T* __restrict__ dz, float* __restrict__ dword, int pad_id,
                                                                  int M, int H, int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int i0 = w * rows_per_wave, i1 = min(M, i0 + rows_per_wave);
    if (i0 >= M) return;
    // The wave's rows and their ids first (one lane each: no load -> load -> load chain per row), then the rows four at a time with
    // every load issued before the first sum; runs are still summed in `order`.
    const int n = i1 - i0;      // <= 64 (launcher)
    int my_row = 0, my_id = pad_id;
    if (lane < n) {
        my_row = order[i0 + lane];
        my_id = ids[my_row];
    }
    // A run that neither starts before this wave's rows nor continues after them is the ONLY contribution to its table row (equal
    // ids are adjacent in `order`): plain 16-byte read-modify-write instead of four atomics per lane.  Typical titles are mostly
    // such runs of length one; the atomics of the first version (20 M per step) bounded the kernel.
    const int id_before = i0 > 0 ? ids[order[i0 - 1]] : -1, id_after = i1 < M ? ids[order[i1]] : -1;
    float acc[CPL][4];
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0 || cur == pad_id) return;
        float* dst = dword + (size_t)cur * H;
        const bool shared_run = cur == id_before || cur == id_after;      // wave-uniform
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = (q * 64 + lane) * 4;
            if (c < H) {
                if (shared_run) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) atomicAdd(dst + c + k, acc[q][k]);
                } else {
                    float4 o = *reinterpret_cast<const float4*>(dst + c);
                    o.x += acc[q][0]; o.y += acc[q][1]; o.z += acc[q][2]; o.w += acc[q][3];
                    *reinterpret_cast<float4*>(dst + c) = o;
                }
            }
        }
    };
    constexpr int U = 4;
    for (int j = 0; j < n; j += U) {
        float v[U][CPL][4];
        int id_u[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = min(j + u, n - 1);
            id_u[u] = j + u < n ? __builtin_amdgcn_readlane(my_id, jj) : pad_id;
            const int row = __builtin_amdgcn_readlane(my_row, jj);
            if (id_u[u] != pad_id) {      // wave-uniform
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int c = (q * 64 + lane) * 4;
                    if (c < H) io<T>::load4(dz + (size_t)row * H + c, v[u][q]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j + u >= n) break;
            const int id = id_u[u];
            if (id != cur) {
                flush();
                cur = id;
#pragma unroll
                for (int q = 0; q < CPL; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[q][k] = 0.f;
            }
            if (id == pad_id) continue;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                if ((q * 64 + lane) * 4 < H) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[q][k] += v[u][q][k];
                }
            }
        }
    }
    flush();
}

// Deterministic mode: ONE wave owns a table row.  The wave at sorted position i acts only when i starts a run of equal ids; it then sums
// the run's rows in `order` (four rows in flight) and adds the total to the table row with a plain read-modify-write -- no other wave
// touches that row, and the summation order is the (stable) sort order.  Long runs ([CLS] / [SEP]: one row per title) are serial in one
// wave: slower than the atomics (~0.3 ms at B = 128), which is the price of the mode.
template <typename T>
__global__ __launch_bounds__(256) void word_scatter_runs_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ order,
                                                                const T* __restrict__ dz, float* __restrict__ dword, int pad_id, int M, int H) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    const int id = ids[order[i]];
    if (id == pad_id || (i > 0 && ids[order[i - 1]] == id)) return;
    for (int c0 = 0; c0 < H; c0 += 256) {      // 256 columns (4 per lane) per sweep over the run
        const int c = c0 + lane * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int j = i;
        while (j < M) {
            int rows[4];
            int n = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rows[u] = -1;
                if (j + u < M) {
                    const int r = order[j + u];
                    if (ids[r] == id && n == u) { rows[u] = r; ++n; }
                }
            }
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (rows[u] >= 0 && c < H) io<T>::load4(dz + (size_t)rows[u] * H + c, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (rows[u] >= 0 && c < H) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] += v[u][k];
                }
            j += n;
            if (n < 4) break;
        }
        if (c < H) {
            float4 o = *reinterpret_cast<const float4*>(dword + (size_t)id * H + c);
            o.x += acc[0]; o.y += acc[1]; o.z += acc[2]; o.w += acc[3];
            *reinterpret_cast<float4*>(dword + (size_t)id * H + c) = o;
        }
    }
}

template <typename T>      // row width not a multiple of the 16-byte vector: one element per lane per trip
__global__ __launch_bounds__(256) void pos_type_grad_scalar_kernel(const T* __restrict__ dz, float* __restrict__ dpos,
                                                                   float* __restrict__ dtype0, int nseq, int Tlen, int H,
                                                                   int seq_per_block) {
    const int t = blockIdx.x;
    const int s0 = blockIdx.y * seq_per_block, s1 = min(nseq, s0 + seq_per_block);
    for (int c = threadIdx.x; c < H; c += 256) {
        float acc = 0.f;
        for (int sq = s0; sq < s1; ++sq) acc += io<T>::load1(dz + ((size_t)sq * Tlen + t) * H + c);
        atomicAdd(dpos + (size_t)t * H + c, acc);
        if (dtype0) atomicAdd(dtype0 + c, acc);
    }
}

extern "C" int morec_bert_embed_bwd(const int32_t* ids, const void* dz, float* dword, float* dpos, float* dtype0,
                                    int pad_id, int M, int T, int H, int dtype, const int32_t* order, void* stream) {
    if (!ids || !dz || !dword || !dpos || M <= 0 || T <= 0 || H <= 0 || M % T) return MOREC_E_ARG;
    if (H % 4) return MOREC_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int spb = 64;
    dim3 g1((M + 3) / 4), g2(T, (M / T + spb - 1) / spb);
    const int ev = dtype == MOREC_F32 ? 4 : 8;
    const bool vec = H % ev == 0 && H / ev <= 256;          // one 16-byte column vector per thread
    const int rpw = 32;                                   // sorted rows per wave
    dim3 g3((((M + rpw - 1) / rpw) + 3) / 4);
    const bool sorted = order != nullptr && H <= 1024;
    const int T_ = T;      // (the lambda below names its storage type T)
    const bool det = morec_deterministic();
    // deterministic mode has no atomic fall-backs: it needs the token-id order (one writer wave per table row) and a row width the
    // vector kernel takes; anything else is refused instead of silently summing in arrival order
    if (det && (order == nullptr || !vec)) return MOREC_E_UNSUPPORTED;
    int rc_pos = 1;
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            if (det)
                hipLaunchKernelGGL((word_scatter_runs_kernel<T>), g1, dim3(256), 0, s, ids, order, (const T*)dz, dword, pad_id, M, H);
            else if (sorted) hipLaunchKernelGGL((word_scatter_sorted_kernel<T, 4>), g3, dim3(256), 0, s, ids, order, (const T*)dz, dword, pad_id, M, H, rpw);
            else hipLaunchKernelGGL((word_scatter_kernel<T>), g1, dim3(256), 0, s, ids, (const T*)dz, dword, pad_id, M, H);
            if (vec) rc_pos = pos_type_grad_launch<T>((const T*)dz, dpos, dtype0, M / T_, T_, H, s);
            else hipLaunchKernelGGL((pos_type_grad_scalar_kernel<T>), g2, dim3(256), 0, s, (const T*)dz, dpos, dtype0, M / T_, T_, H, spb);
        }))
        return MOREC_E_DTYPE;
    if (rc_pos < 0) return (int)hipErrorOutOfMemory;      // deterministic scratch unavailable (e.g. it would have to grow under graph capture)
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx,
                                                          T* __restrict__ out, int R, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* src = table + (size_t)idx[row] * D;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        const float o[4] = {v.x, v.y, v.z, v.w};
        io<T>::store4(out + (size_t)row * D + c, o);
    }
}

extern "C" int morec_gather_rows(const float* table, const int32_t* idx, void* out, int R, int D, int dtype,
                                 void* stream) {
    if (!table || !idx || !out || R <= 0 || D <= 0) return MOREC_E_ARG;
    if (D % 4) return MOREC_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((R + 3) / 4);
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((gather_rows_kernel<T>), grid, dim3(256), 0, s, table, idx, (T*)out, R, D);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const T* __restrict__ d, const int32_t* __restrict__ idx,
                                                               float* __restrict__ dtable, int R, int D, int pad_id) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int id = idx[row];
    if (id == pad_id) return;
    float* dst = dtable + (size_t)id * D;
    for (int c = lane * 4; c < D; c += 256) {
        float v[4];
        io<T>::load4(d + (size_t)row * D + c, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(dst + c + k, v[k]);
    }
}

// Deterministic mode: the wave of source row r acts only when no EARLIER row carries the same index (it scans idx[0 .. r): R is the
// B (S + 1) slots of a batch, a few thousand); it then adds the rows r, r' > r, ... with that index in row order -- one writer per table row.
template <typename T>
__global__ __launch_bounds__(256) void scatter_add_rows_det_kernel(const T* __restrict__ d, const int32_t* __restrict__ idx,
                                                                   float* __restrict__ dtable, int R, int D, int pad_id) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int id = idx[row];
    if (id == pad_id) return;
    bool dup = false;
    for (int r0 = 0; r0 < row && !dup; r0 += 64) {
        const int r = r0 + lane;
        dup = __ballot(r < row && idx[r] == id) != 0ull;
    }
    if (dup) return;
    float* dst = dtable + (size_t)id * D;
    for (int c0 = 0; c0 < D; c0 += 256) {           // wave-uniform trip count: every lane takes part in the ballots below
        const int c = c0 + lane * 4;
        const bool mine = c < D;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r0 = row; r0 < R; r0 += 64) {      // 64 candidate rows at a time: ALL lanes test, the wave walks the hits in order
            const int r = r0 + lane;
            unsigned long long hit = __ballot(r < R && idx[r] == id);
            while (hit) {
                const int b = __builtin_ctzll(hit);
                hit &= hit - 1;
                if (mine) {
                    float v[4];
                    io<T>::load4(d + (size_t)(r0 + b) * D + c, v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] += v[k];
                }
            }
        }
        if (mine) {
            float4 o = *reinterpret_cast<const float4*>(dst + c);
            o.x += acc[0]; o.y += acc[1]; o.z += acc[2]; o.w += acc[3];
            *reinterpret_cast<float4*>(dst + c) = o;
        }
    }
}

extern "C" int morec_scatter_add_rows(const void* d, const int32_t* idx, float* dtable, int R, int D, int pad_id,
                                      int dtype, void* stream) {
    if (!d || !idx || !dtable || R <= 0 || D <= 0) return MOREC_E_ARG;
    if (D % 4) return MOREC_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((R + 3) / 4);
    const bool det = morec_deterministic();
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            if (det) hipLaunchKernelGGL((scatter_add_rows_det_kernel<T>), grid, dim3(256), 0, s, (const T*)d, idx, dtable, R, D, pad_id);
            else hipLaunchKernelGGL((scatter_add_rows_kernel<T>), grid, dim3(256), 0, s, (const T*)d, idx, dtable, R, D, pad_id);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void strided_rows_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int D,
                                                           int in_stride, int out_stride) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    for (int c = lane * 4; c < D; c += 256) {
        float v[4];
        io<T>::load4(in + (size_t)row * in_stride * D + c, v);
        io<T>::store4(out + (size_t)row * out_stride * D + c, v);
    }
}

extern "C" int morec_strided_rows_copy(const void* in, void* out, int R, int D, int in_row_stride, int out_row_stride,
                                       int dtype, void* stream) {
    if (!in || !out || R <= 0 || D <= 0 || in_row_stride <= 0 || out_row_stride <= 0) return MOREC_E_ARG;
    if (D % 4) return MOREC_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((R + 3) / 4);
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((strided_rows_kernel<T>), grid, dim3(256), 0, s, (const T*)in, (T*)out, R, D, in_row_stride, out_row_stride);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// row gather / scatter by int32 index arrays (either side optional): unpadded token layouts, [CLS] rows of packed sequences
template <typename T>
__global__ __launch_bounds__(256) void indexed_rows_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                           const int32_t* __restrict__ in_idx, const int32_t* __restrict__ out_idx,
                                                           int R, int D) {
    constexpr int EV = vio<T>::EV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int si = in_idx ? in_idx[row] : row;         // negative gather index: a zero row (the [PAD] rows of an unpacked layout)
    const size_t src = (size_t)(si < 0 ? 0 : si), dst = out_idx ? (size_t)out_idx[row] : (size_t)row;
    for (int c = lane * EV; c < D; c += 64 * EV) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (si >= 0) v = vio<T>::load_raw(in + src * D + c);
        *reinterpret_cast<uint4*>(out + dst * D + c) = v;
    }
}

extern "C" int morec_indexed_rows_copy(const void* in, void* out, const int32_t* in_idx, const int32_t* out_idx, int R, int D,
                                       int dtype, void* stream) {
    if (!in || !out || R <= 0 || D <= 0) return MOREC_E_ARG;
    if (D % 8) return MOREC_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((R + 3) / 4);
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((indexed_rows_kernel<T>), grid, dim3(256), 0, s, (const T*)in, (T*)out, in_idx, out_idx, R, D);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
