// eval.hip -- HR@10 / nDCG@10 ranking without an argsort (T/data_utils/metrics.py:96-102,49-57).
// The reference scores every user against every item, sets history scores to -inf, drops column 0 and
// runs a full descending argsort per user in a Python loop to find the 1-based position of the target.
// Equivalent for tie-free scores: rank = 1 + #{items i >= 1, i not in history(u), i != target :
// score(u, i) > score(u, target)}.  The score tiles come out of the exact-fp32 MFMA main loop and are
// consumed in registers -- the [U, item_num+1] score matrix is never written.
#include "gemm_core.hpp"

namespace {
struct EvalArgs {
    const float* prec;
    const float* emb;
    const int32_t* hist;
    const int32_t* target;
    const float* tscore;
    int32_t* rank;
    int U, I, D, Hmax, tiles_m, tiles_n;
};

// tscore[u] = prec[u] . emb[target[u]], or -inf when the target is itself in the (masked) history;
// also initialises rank[u] = 1.  One wave per user.
__global__ __launch_bounds__(256) void eval_target_kernel(const float* __restrict__ prec, const float* __restrict__ emb,
                                                          const int32_t* __restrict__ hist,
                                                          const int32_t* __restrict__ target, float* __restrict__ tscore,
                                                          int32_t* __restrict__ rank, int U, int D, int Hmax) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= U) return;
    const int t = target[u];
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) acc += prec[(size_t)u * D + c] * emb[(size_t)t * D + c];
    acc = wave_sum(acc);
    bool inhist = false;
    for (int h = lane; h < Hmax; h += 64) inhist |= (hist[(size_t)u * Hmax + h] == t);
    inhist = __any(inhist);
    if (lane == 0) {
        tscore[u] = inhist ? -INFINITY : acc;
        rank[u] = 1;
    }
}

__global__ __launch_bounds__(256) void eval_rank_kernel(EvalArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int m0 = (wg / p.tiles_n) * 128, n0 = (wg % p.tiles_n) * 128;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    gemm_mainloop<float, 2>(p.prec, p.emb, p.U, p.I, p.D, p.D, m0, n0, 0, p.D, smem, acc);
    int32_t* s_hist = reinterpret_cast<int32_t*>(smem);   // [128][Hmax]
    for (int i = threadIdx.x; i < 128 * p.Hmax; i += 256) {
        const int u = m0 + i / p.Hmax;
        s_hist[i] = (u < p.U) ? p.hist[(size_t)u * p.Hmax + (i % p.Hmax)] : -1;
    }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int u = acc_row(m0, mi);
        int cnt = 0;
        if (u < p.U) {
            const float ts = p.tscore[u];
            const int tgt = p.target[u];
            const int32_t* hu = s_hist + (u - m0) * p.Hmax;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = acc_col(n0, ni);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int item = n + r;
                    if (item >= 1 && item < p.I && item != tgt && acc[mi][ni][r] > ts) {
                        bool inh = false;
                        for (int h = 0; h < p.Hmax; ++h) inh |= (hu[h] == item);
                        cnt += inh ? 0 : 1;
                    }
                }
            }
        }
        cnt += __shfl_xor(cnt, 16, 64);
        cnt += __shfl_xor(cnt, 32, 64);
        if (u < p.U && ((threadIdx.x & 63) >> 4) == 0 && cnt) atomicAdd(p.rank + u, cnt);
    }
}
}  // namespace

extern "C" int morec_eval_rank(const float* prec, const float* item_emb, const int32_t* hist, int Hmax,
                               const int32_t* target, int32_t* rank, float* tscore_ws, int U, int n_items_plus1,
                               int D, void* stream) {
    if (!prec || !item_emb || !hist || !target || !rank || !tscore_ws || U <= 0 || n_items_plus1 <= 1 || D <= 0 ||
        Hmax <= 0)
        return MOREC_E_ARG;
    if (D % 4 || !aligned16(prec) || !aligned16(item_emb)) return MOREC_E_ALIGN;
    if (128 * Hmax * 4 > GemmTile<float, 2>::LDS_BYTES) return MOREC_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(eval_target_kernel, dim3((U + 3) / 4), dim3(256), 0, s, prec, item_emb, hist, target, tscore_ws,
                       rank, U, D, Hmax);
    MOREC_CHECK_LAUNCH();
    EvalArgs a{prec, item_emb, hist, target, tscore_ws, rank, U, n_items_plus1, D, Hmax, (U + 127) / 128,
               (n_items_plus1 + 127) / 128};
    using G = GemmTile<float, 2>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              G::LDS_BYTES);
    hipLaunchKernelGGL(eval_rank_kernel, dim3(a.tiles_m * a.tiles_n), dim3(256), G::LDS_BYTES, s, a);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
