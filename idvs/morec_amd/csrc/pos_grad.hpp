// Position-row (and type-row) gradients of the embedding stages: dpos[m % T, :] += dz[m, :], dtype0 += sum of all rows.
// Shared by morec_bert_embed_bwd (embed.hip) and morec_pos_grad (layernorm.hip).
#pragma once
#include "common.hpp"

// dpos[t] = sum over sequences of dz[seq*T + t]; dtype0 = sum over all rows.  A block owns 4 * groups sequences and walks the T
// positions: a thread owns one 16-byte column vector of the row and four of the block's sequences, with the rows of the next TWO
// positions in flight while this position's partials are folded through LDS into ONE coalesced atomic per column (the type-row
// sum stays in registers over all positions: one atomic per column per block).  The first version read two bytes per lane per
// dependent trip and sent every block's type-row sums to the same H addresses: 118 us for 124 MB.  Few sequences (the recommender's
// [B, S] rows): the positions are spread over blockIdx.y so that the launch still has a few hundred blocks.
template <typename T>
__global__ __launch_bounds__(256) void pos_type_grad_kernel(const T* __restrict__ dz, float* __restrict__ dpos,
                                                            float* __restrict__ dtype0, int nseq, int Tlen, int H,
                                                            int t_per_block, int det = 0) {
    constexpr int EV = vio<T>::EV, R = 4;
    extern __shared__ __attribute__((aligned(16))) float sm_pt[];      // [groups][H]
    const int nv = H / EV;                                   // <= 256, H % EV == 0 (launcher)
    const int groups = 256 / nv;
    const int grp = (int)threadIdx.x / nv, cv = (int)threadIdx.x - grp * nv;
    const bool active = grp < groups;
    const int s0 = blockIdx.x * (R * groups);
    const int t0 = blockIdx.y * t_per_block, t1 = min(Tlen, t0 + t_per_block);
    // this thread's sequences: s0 + grp + u * groups, u < n_ok
    const int n_ok = active ? max(0, min(R, (nseq - s0 - grp + groups - 1) / groups)) : 0;
    const T* base = dz + ((size_t)(n_ok ? s0 + grp : 0) * Tlen) * H + cv * EV;
    const size_t ustride = (size_t)groups * Tlen * H;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    uint4 ra[R], rb[R];
#define PT_FETCH(r, t)                                                                                              \
    _Pragma("unroll") for (int u = 0; u < R; ++u)                                                                    \
        r[u] = (u < n_ok && (t) < t1) ? vio<T>::load_raw(base + u * ustride + (size_t)(t) * H) : zero4
    float ty[EV];
#pragma unroll
    for (int k = 0; k < EV; ++k) ty[k] = 0.f;
    auto fold = [&](const float (&acc)[EV], float* dst) {      // the groups' partials -> ONE coalesced atomic per column
        if (active) store_f32v<EV>(sm_pt + (size_t)grp * H + cv * EV, acc);
        __syncthreads();
        for (int c = threadIdx.x; c < H; c += 256) {
            float v = 0.f;
            for (int g = 0; g < groups; ++g) v += sm_pt[(size_t)g * H + c];
            if (det) dst[c] = v;      // deterministic mode: dst is this block's own partial row (launcher folds the blocks in order)
            else atomicAdd(dst + c, v);
        }
        __syncthreads();
    };
    // consume position t from r, refill r with position t + 2, fold and emit
#define PT_POSITION(r, t)                                                                     \
    {                                                                                         \
        float acc[EV];                                                                        \
        _Pragma("unroll") for (int k = 0; k < EV; ++k) acc[k] = 0.f;                          \
        _Pragma("unroll") for (int u = 0; u < R; ++u) {                                       \
            float e[EV];                                                                      \
            vio<T>::unpack(r[u], e);                                                          \
            _Pragma("unroll") for (int k = 0; k < EV; ++k) acc[k] += e[k];                    \
        }                                                                                     \
        PT_FETCH(r, (t) + 2);                                                                 \
        _Pragma("unroll") for (int k = 0; k < EV; ++k) ty[k] += acc[k];                       \
        fold(acc, dpos + ((size_t)(det ? blockIdx.x * Tlen : 0) + (t)) * H);                  \
    }
    PT_FETCH(ra, t0);
    PT_FETCH(rb, t0 + 1);
    for (int t = t0; t < t1; t += 2) {
        PT_POSITION(ra, t);
        if (t + 1 < t1) PT_POSITION(rb, t + 1);
    }
    if (dtype0) fold(ty, dtype0 + (det ? ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * H : 0));
#undef PT_FETCH
#undef PT_POSITION
}

// 0: the row width does not fit the one-vector-per-thread layout (the callers keep a scalar kernel for that); 1: launched; -1: the
// deterministic mode's scratch is not available (callers return an error: never a silent fall-back to the atomic kernel)
template <typename T>
inline int pos_type_grad_launch(const T* dz, float* dpos, float* dtype0, int nseq, int Tlen, int H, hipStream_t s) {
    constexpr int EV = vio<T>::EV;
    if (H % EV || H / EV > 256) return 0;
    const int grp = 256 / (H / EV);
    const int seq_blocks = (nseq + 4 * grp - 1) / (4 * grp);
    int t_chunks = 512 / seq_blocks;
    t_chunks = t_chunks < 1 ? 1 : t_chunks > Tlen ? Tlen : t_chunks;
    const int tpb = (Tlen + t_chunks - 1) / t_chunks;
    const int ty_blocks = (Tlen + tpb - 1) / tpb;
    if (morec_deterministic()) {      // per-block partial rows [seq_blocks][Tlen][H] (+ [seq_blocks * ty_blocks][H] for the type row), folded in block order
        const size_t n_pos = (size_t)seq_blocks * Tlen * H, n_ty = dtype0 ? (size_t)seq_blocks * ty_blocks * H : 0;
        float* part = morec_det_scratch(s, n_pos + n_ty);
        if (!part) return -1;
        hipLaunchKernelGGL((pos_type_grad_kernel<T>), dim3(seq_blocks, ty_blocks), dim3(256), (size_t)grp * H * sizeof(float), s, dz,
                           part, dtype0 ? part + n_pos : nullptr, nseq, Tlen, H, tpb, 1);
        (void)morec_det_fold_add(part, dpos, seq_blocks, (size_t)Tlen * H, (size_t)Tlen * H, s);
        if (dtype0) (void)morec_det_fold_add(part + n_pos, dtype0, seq_blocks * ty_blocks, (size_t)H, (size_t)H, s);
        return 1;
    }
    hipLaunchKernelGGL((pos_type_grad_kernel<T>), dim3(seq_blocks, ty_blocks), dim3(256), (size_t)grp * H * sizeof(float), s, dz,
                       dpos, dtype0, nseq, Tlen, H, tpb);
    return 1;
}
