// gemm_tn.hip -- weight-gradient GEMM without transposed copies (bf16):
//      C[n, k] (+)= sum_m DY[m, n] * X[m, k]           (dW = dY^T X of every nn.Linear on the path)
// Both operands are contracted over their ROW index (tokens), i.e. the MFMA fragments are columns of the
// row-major global tiles.  The tiles are staged as they lie in memory ([64 tokens][256 columns], LDS-DMA, one
// wave-instruction = two 512-byte rows) and the fragments are fetched with ds_read_b64_tr_b16, the hardware
// transpose read: lane (c = lane & 15, g = lane >> 4) receives, for column block*16 + c, the tokens
// 4g .. 4g+3 and 16+4g .. 16+4g+3 of a 32-token step.  DY and X use the same token permutation, which a
// dot product does not see.  Round 1 ran this product as an NT GEMM over explicitly transposed copies of DY
// and X: 8.3 ms of pure transposition per step (rocprof r01c).
// Bank conflicts: the 8 rows a 32-lane half touches would share banks at a 512-byte pitch; the 32-byte column
// block index is XOR-ed with (token & 7) on the DMA source side and on the read side (destination stays linear).
#include "gemm_core.hpp"
#include "gemm_args.hpp"

int gemm_tn8p_try_launch(const bf16* DY, const bf16* X, float* out, size_t slab_stride, int M, int N, int K, int ldy, int ldx, int ldo,
                         int mchunk, int zs, int dtype, hipStream_t s);

namespace {
constexpr int TT = 64;                 // tokens per stage
constexpr int TC = 256;                // columns per operand tile
constexpr int ROWB = TC * 2;           // 512 bytes per staged row
constexpr int OPB = TT * ROWB;         // 32 KiB per operand per stage
constexpr int STAGE = 2 * OPB;
constexpr int LDS_TN = 2 * STAGE;      // 128 KiB
constexpr int NWAVE = 16, NTHREADS = 1024;
constexpr int WN = 4, WK = 4;          // wave grid: 4 (n) x 4 (k); wave tile 64 n x 64 k (16 waves = 4 per SIMD cover each
constexpr int NI = 4, KI = 4;          //   other's LDS / barrier waits: +3..14 % over 8 waves of 128 x 64 on the NT kernel)
constexpr int DPW = (TT * ROWB / 1024) / NWAVE;   // LDS-DMA instructions per operand per wave per stage
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

struct TnArgs {
    const bf16* DY;
    const bf16* X;
    float* C;
    int M, N, K, ldy, ldx, ldc, mchunk, tiles_n, tiles_k, atomic;
    float* slabs;     // split-m partials [gridDim.z][N][K] (plain stores) -- reduced by reduce_slabs_kernel
};

__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int ks, int blk) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int row = ks * 32 + 4 * g + (c >> 2);
    const char* p0 = tile + row * ROWB + ((blk ^ (row & 7)) * 32) + 8 * (c & 3);
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 16 * ROWB));   // (row + 16) & 7 == row & 7
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

template <typename T16>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(TnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int wg = xcd_remap(blockIdx.x, p.tiles_n * p.tiles_k);
    const int n0 = (wg / p.tiles_k) * TC, k0 = (wg % p.tiles_k) * TC;
    const int mbeg = blockIdx.z * p.mchunk, mend = min(p.M, mbeg + p.mchunk);
    const int nt = (mend - mbeg + TT - 1) / TT;
    if (nt <= 0) return;
    const bool tail = ((mend - mbeg) % TT) != 0;

    f32x4_t acc[NI][KI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < KI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // DMA geometry: instruction q of an operand covers token rows 2q, 2q+1; lane -> (row, 16-byte physical slot)
    const int drow = lane >> 5, pslot = lane & 31;
    const bf16* ysrc[DPW];
    const bf16* xsrc[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int row = (wave * DPW + i) * 2 + drow;                     // token row within the stage
        const int lslot = (((pslot >> 1) ^ (row & 7)) << 1) | (pslot & 1); // logical 16-byte column slot
        const int yc = min(n0 + lslot * 8, p.N - 8), xc = min(k0 + lslot * 8, p.K - 8);   // clamp: unused output columns
        ysrc[i] = p.DY + (size_t)row * p.ldy + yc;
        xsrc[i] = p.X + (size_t)row * p.ldx + xc;
    }
    auto dma = [&](int stage, int m) {
        char* base = smem + stage * STAGE + (wave * DPW) * 1024;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ysrc[i] + (size_t)m * p.ldy), (lptr_t)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + (size_t)m * p.ldx), (lptr_t)(base + OPB + i * 1024), 16, 0, 0);
        }
    };
    auto stage_tail = [&](int stage, int m) {   // token tail: zero-filled, register staged, same swizzled image
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < (TT * 32) / NTHREADS; ++i) {
            const int v = tid + NTHREADS * i;
            const int row = v >> 5, ls = v & 31;
            const int yc = min(n0 + ls * 8, p.N - 8), xc = min(k0 + ls * 8, p.K - 8);
            const bool in = (m + row) < mend;
            const uint4 ry = in ? *reinterpret_cast<const uint4*>(p.DY + (size_t)(m + row) * p.ldy + yc) : make_uint4(0, 0, 0, 0);
            const uint4 rx = in ? *reinterpret_cast<const uint4*>(p.X + (size_t)(m + row) * p.ldx + xc) : make_uint4(0, 0, 0, 0);
            const int off = row * ROWB + ((((ls >> 1) ^ (row & 7)) << 1) | (ls & 1)) * 16;
            *reinterpret_cast<uint4*>(base + off) = ry;
            *reinterpret_cast<uint4*>(base + OPB + off) = rx;
        }
    };
    auto stage = [&](int st, int t) {
        if (tail && t == nt - 1) stage_tail(st, mbeg + t * TT);
        else dma(st, mbeg + t * TT);
    };

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(cur ^ 1, t + 1);
        const char* ty = smem + cur * STAGE;
        const char* tx = ty + OPB;
#pragma unroll
        for (int ks = 0; ks < TT / 32; ++ks) {
            bf16x8_t fy[NI], fx[KI];
#pragma unroll
            for (int i = 0; i < NI; ++i) fy[i] = tr_frag(ty, ks, wn * NI + i);
#pragma unroll
            for (int j = 0; j < KI; ++j) fx[j] = tr_frag(tx, ks, wk * KI + j);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < KI; ++j) acc[i][j] = h16<T16>::mma16(fx[j], fy[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // acc[i][j][r] = C[n0 + (wn*NI + i)*16 + (lane & 15)][k0 + (wk*KI + j)*16 + 4*(lane >> 4) + r]
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n0 + (wn * NI + i) * 16 + c;
        if (n >= p.N) continue;
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int k = k0 + (wk * KI + j) * 16 + 4 * g;
            if (k >= p.K) continue;
            if (p.slabs) {
                float* sl = p.slabs + ((size_t)blockIdx.z * p.N + n) * p.K + k;
                *reinterpret_cast<float4*>(sl) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                continue;
            }
            float* dst = p.C + (size_t)n * p.ldc + k;
            if (p.atomic && gridDim.z == 1) {      // one token chunk: this lane is the only writer of its four elements -- read-modify-write, no atomics
                float4 cur = *reinterpret_cast<const float4*>(dst);   // (SASRec weights at D = 2048, 640 rows: 16.8 M scalar atomics took 150 us per launch)
                cur.x += acc[i][j][0]; cur.y += acc[i][j][1]; cur.z += acc[i][j][2]; cur.w += acc[i][j][3];
                *reinterpret_cast<float4*>(dst) = cur;
            } else if (p.atomic) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dst + r, acc[i][j][r]);
            } else {
                *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
    }
}
// C[n, k] (+)= sum_s slabs[s][n][k]   (deterministic split-m reduction; replaces 65k fp32 atomics per workgroup).  TO = bf16: the sum
// is rounded once on the way out (overwrite only) -- the scoring backward's dP, which nothing reads in fp32.
template <typename TO>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, TO* __restrict__ C, int N, int K,
                                                           int ldc, int nslab, int accumulate) {
    const size_t total4 = (size_t)N * K / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int n = (int)(e / K), k = (int)(e % K);
        float4 s = reinterpret_cast<const float4*>(slabs)[i];
        // eight slabs' loads in flight per thread, added in slab order (the sum is the sequential one): with one load per trip a fold over 40 - 64 slabs of a
        // small weight (few blocks) was a chain of memory round trips -- 35 - 48 us for the BERT-tiny weights, 23 us on average in the Swin-T step
        int t = 1;
        for (; t + 8 <= nslab; t += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(slabs + (size_t)(t + u) * N * K)[i];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; t < nslab; ++t) {
            const float4 v = reinterpret_cast<const float4*>(slabs + (size_t)t * N * K)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if constexpr (sizeof(TO) == 4) {
            float4* dst = reinterpret_cast<float4*>(C + (size_t)n * ldc + k);
            if (accumulate) {
                const float4 c = *dst;
                s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
            }
            *dst = s;
        } else {
            const float o[4] = {s.x, s.y, s.z, s.w};
            io<TO>::store4(C + (size_t)n * ldc + k, o);
        }
    }
}
}  // namespace

extern "C" size_t morec_gemm_tn_workspace_bytes(int N, int K, int split_m) {
    return split_m > 1 ? (size_t)split_m * N * K * sizeof(float) : 0;
}

// c_dtype = MOREC_BF16: only with slabs (split_m > 1 + workspace) and accumulate == 0 -- the fold writes the rounded sum.
int gemm_tn_launch(const void* DY, const void* X, void* C, int c_dtype, int M, int N, int K, int ldy, int ldx, int ldc, int dtype, int split_m,
                   int accumulate, float* workspace, void* stream) {
    if (!DY || !X || !C || M <= 0 || N <= 0 || K <= 0) return MOREC_E_ARG;
    if (!is_h16(dtype)) return MOREC_E_UNSUPPORTED;   // the exact-fp32 path uses transposed copies + morec_gemm_nt
    if (c_dtype != MOREC_F32 && c_dtype != dtype) return MOREC_E_DTYPE;
    if (N % 8 || K % 8 || ldy % 8 || ldx % 8 || ldc % 4 || !aligned16(DY) || !aligned16(X) || !aligned16(C)) return MOREC_E_ALIGN;
    if (split_m > 1 && !accumulate && !workspace) return MOREC_E_ARG;      // the atomic path needs a caller-zeroed C
    TnArgs a;
    a.DY = reinterpret_cast<const bf16*>(DY); a.X = reinterpret_cast<const bf16*>(X); a.C = reinterpret_cast<float*>(C);
    a.M = M; a.N = N; a.K = K; a.ldy = ldy; a.ldx = ldx; a.ldc = ldc; a.atomic = accumulate ? 1 : 0;
    const int split = split_m < 1 ? 1 : split_m;
    int mchunk = (M + split - 1) / split;
    mchunk = ((mchunk + TT - 1) / TT) * TT;
    a.mchunk = mchunk;
    const int zs = (M + mchunk - 1) / mchunk;
    a.tiles_n = (N + TC - 1) / TC; a.tiles_k = (K + TC - 1) / TC;
    a.slabs = (zs > 1 && workspace) ? workspace : nullptr;
    if (a.slabs && (K % 4 || !aligned16(workspace))) return MOREC_E_ALIGN;
    if (c_dtype != MOREC_F32 && (!a.slabs || accumulate)) return MOREC_E_UNSUPPORTED;
    if (zs > 1 && !a.slabs && !accumulate) return MOREC_E_ARG;
    int r8 = G8_NOT_TAKEN;
    if (a.slabs)                    // split-m partials -> slabs: the eight-phase kernel (gemm_tn8p.hip) when the launch is large enough
        r8 = gemm_tn8p_try_launch(a.DY, a.X, a.slabs, (size_t)N * K, M, N, K, ldy, ldx, K, mchunk, zs, dtype, reinterpret_cast<hipStream_t>(stream));
    else if (zs == 1 && !accumulate)
        r8 = gemm_tn8p_try_launch(a.DY, a.X, a.C, 0, M, N, K, ldy, ldx, ldc, mchunk, 1, dtype, reinterpret_cast<hipStream_t>(stream));
    if (r8 != G8_NOT_TAKEN && r8 != MOREC_OK) return r8;
    if (r8 == G8_NOT_TAKEN) {
        by_h16(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TN);   // thread-safe one-time set-up
            (void)attr_rc;
            hipLaunchKernelGGL(gemm_tn_kernel<T>, dim3(a.tiles_n * a.tiles_k, 1, zs), dim3(NTHREADS), LDS_TN, reinterpret_cast<hipStream_t>(stream), a);
        });
        MOREC_CHECK_LAUNCH();
    }
    if (a.slabs) {
        const size_t total4 = (size_t)N * K / 4;
        const unsigned blocks = (unsigned)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
        if (c_dtype == MOREC_F32)
            hipLaunchKernelGGL(reduce_slabs_kernel<float>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a.slabs, a.C, N,
                               K, ldc, zs, accumulate ? 1 : 0);
        else if (c_dtype == MOREC_BF16)
            hipLaunchKernelGGL(reduce_slabs_kernel<bf16>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a.slabs,
                               reinterpret_cast<bf16*>(C), N, K, ldc, zs, 0);
        else
            hipLaunchKernelGGL(reduce_slabs_kernel<f16>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a.slabs,
                               reinterpret_cast<f16*>(C), N, K, ldc, zs, 0);
        MOREC_CHECK_LAUNCH();
    }
    return MOREC_OK;
}

extern "C" int morec_gemm_tn(const void* DY, const void* X, float* C, int M, int N, int K, int ldy, int ldx, int ldc,
                             int dtype, int split_m, int accumulate, float* workspace, void* stream) {
    return gemm_tn_launch(DY, X, C, MOREC_F32, M, N, K, ldy, ldx, ldc, dtype, split_m, accumulate, workspace, stream);
}
