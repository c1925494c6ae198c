// gemm_nt_generic.hpp -- the two-buffer NT GEMM kernel + its tile-shape / epilogue dispatch, shared by gemm.hip (fp32 and bf16
// operands) and gemm_f16.hip (fp16 operands: a translation unit of its own so that the two sets of instantiations compile in parallel).
#pragma once
#include <stdlib.h>
#include "gemm_core.hpp"
#include "gemm_args.hpp"

// ACT is a compile-time epilogue selector (0 linear, 1 GELU, 2 ReLU, 3 x GELU'(dact_in), 4 x ReLU'(dact_in)): with the
// activation chosen at run time every unrolled accumulator block carried the erf / exp expansions and the kernel grew
// to ~42k instructions (330 KB of code against a 64 KB instruction cache) -- the epilogue then took as long as the
// K = 768 main loop purely on instruction fetch.
// CS: also leave the column sums of the finished tile in p.colsum (only instantiated for the activation-derivative
// epilogues: d(bias) of the layer whose pre-activation gradient this GEMM produces).
template <typename G, typename TI, typename TO, int ACT, bool CS = false>
__global__ __launch_bounds__(G::THREADS) void gemm_nt_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tm = wg / p.tiles_n, tn = wg % p.tiles_n;
    const int m0 = tm * G::TM, n0 = tn * G::TN;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    f32x4_t acc[G::MI][G::NI];
#pragma unroll
    for (int i = 0; i < G::MI; ++i)
#pragma unroll
        for (int j = 0; j < G::NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if constexpr (G::NST > 2)
        gemm_mainloop_ring<G, TI>(reinterpret_cast<const TI*>(p.A), reinterpret_cast<const TI*>(p.B), p.M, p.N, p.lda, p.ldb, m0,
                                  n0, kbeg, kend, smem, acc);
    else
        gemm_mainloop_cfg<G, TI>(reinterpret_cast<const TI*>(p.A), reinterpret_cast<const TI*>(p.B), p.M, p.N, p.lda, p.ldb, m0,
                                 n0, kbeg, kend, smem, acc);

    TO* C = reinterpret_cast<TO*>(p.C);
    TO* aux = reinterpret_cast<TO*>(p.aux_out);
    const TO* din = reinterpret_cast<const TO*>(p.dact_in);
    const int lane = threadIdx.x & 63, c16 = lane & 15, g4 = lane >> 4;
    const int wave = threadIdx.x >> 6, wm = wave / G::WN, wn = wave % G::WN;

    // value of accumulator element block (mi, ni) after the fused epilogue; `pre` receives acc*alpha + bias
    auto finish = [&](int mi, int ni, int m, int n, float (&v)[4], float (&pre)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r] * p.alpha;
        if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pre[r] = v[r];
        if constexpr (ACT == 1) {
            gelu4(v);
        } else if constexpr (ACT == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if constexpr (ACT == 3 || ACT == 4 || ACT == 5) {
            float u[4];
            io<TO>::load4(din + (size_t)m * p.ldc + n, u);
            if constexpr (ACT == 3) {
                dgelu4_mul(v, u);
            } else if constexpr (ACT == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (u[r] > 0.f) ? v[r] : 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= u[r];
            }
        }
        if constexpr (ACT == 1 || ACT == 2) {       // aux_out = act'(pre) instead of pre (morec_gemm_desc.aux_deriv)
            if (p.aux_deriv) {
                if constexpr (ACT == 1) {
                    dgelu4(pre);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre[r] = pre[r] > 0.f ? 1.f : 0.f;
                }
            }
        }
    };

    // CS: column sums of the finished tile, taken from the LDS staging of the output (no extra accumulator registers per
    // block): one partial row of p.colsum per wave row block (wave epilogue: MI * 16 rows -- 64, or 32 on the 64 x 64 tiles of gemm_small.hip) or per tile (block epilogue), each written
    // by exactly one wave / workgroup -- the launcher folds the partial rows with one small column-sum launch.
    float csum0 = 0.f, csum1 = 0.f;
    // Fast path: the output tile goes through LDS (free after the main loop) so that every global store is a
    // full 16-byte lane write along a row -- the direct form (8-byte pieces, 16 different rows per wave
    // instruction) is store-issue bound and cost more than the K = 768 main loop itself.
    constexpr int EPV_O = 16 / (int)sizeof(TO);
    if (p.accumulate == 0 && p.vec_store && p.wave_epilogue) {
        // Wave-local form of the same idea: every wave transposes its own 16-row blocks through a private LDS slice (no
        // workgroup barriers in the epilogue) and stores 16-byte lanes along rows of its NI * 16 columns.
        constexpr int WROWB = G::NI * 16 * (int)sizeof(TO) + 16;
        constexpr int VPRW = G::NI * 16 * (int)sizeof(TO) / 16;
        static_assert(G::NWAVES * 16 * WROWB <= G::LDS_BYTES, "wave slices do not fit");
        char* ws = smem + wave * 16 * WROWB;
        auto wfence = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        constexpr int WCOLS = G::NI * 16, LG = 64 / WCOLS;      // the wave's columns; lane groups sharing a column
        auto flushw = [&](TO* dst, int mi) {
            wfence();
            if constexpr (CS) {
                if (dst == C) {
#pragma unroll
                    for (int r = lane / WCOLS; r < 16; r += LG)
                        csum0 += io<TO>::load1(reinterpret_cast<const TO*>(ws + r * WROWB) + (lane % WCOLS));
                }
            }
#pragma unroll
            for (int v = lane; v < 16 * VPRW; v += 64) {
                const int lrow = v / VPRW, cv = v % VPRW;
                const int m = m0 + wm * G::MI * 16 + mi * 16 + lrow;
                const int n = n0 + wn * G::NI * 16 + cv * EPV_O;
                if (m < p.M && n < p.N)
                    *reinterpret_cast<uint4*>(dst + (size_t)m * p.ldc + n) = *reinterpret_cast<const uint4*>(ws + lrow * WROWB + cv * 16);
            }
            wfence();
        };
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) {
            const int m = acc_row_cfg<G>(m0, mi);
            float vv[G::NI][4];
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) {
                const int n = acc_col_cfg<G>(n0, ni);
                float pre[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[ni][r] = 0.f;
                if (m < p.M && n < p.N) finish(mi, ni, m, n, vv[ni], pre);
                if (aux) io<TO>::store4(reinterpret_cast<TO*>(ws + c16 * WROWB + (ni * 16 + g4 * 4) * (int)sizeof(TO)), pre);
            }
            if (aux) flushw(aux, mi);
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
                io<TO>::store4(reinterpret_cast<TO*>(ws + c16 * WROWB + (ni * 16 + g4 * 4) * (int)sizeof(TO)), vv[ni]);
            flushw(C, mi);
        }
        if constexpr (CS) {
            if constexpr (LG == 2) csum0 += __shfl_xor(csum0, 32, 64);
            const int n = n0 + wn * WCOLS + lane;
            if (lane < WCOLS && n < p.N) p.colsum[(size_t)(tm * G::WM + wm) * p.N + n] = csum0;
        }
        return;
    }
    if (p.accumulate == 0 && p.vec_store) {
        constexpr int ROWB = G::TN * (int)sizeof(TO) + 16;                    // LDS pitch of a staged output row
        constexpr int NP = (G::TM * ROWB + G::LDS_BYTES - 1) / G::LDS_BYTES;  // passes needed
        constexpr int NPASS = NP <= 1 ? 1 : (NP <= 2 ? 2 : (NP <= 4 ? 4 : 8));
        constexpr int MIP = G::MI / NPASS;                                    // mi blocks per pass
        static_assert(MIP >= 1 && G::WM * MIP * 16 * ROWB <= G::LDS_BYTES, "output staging does not fit");
        constexpr int RPP = G::WM * MIP * 16;                                 // rows per pass
        constexpr int VPR = G::TN * (int)sizeof(TO) / 16;                     // 16-byte vectors per row
        constexpr int NCP = G::TN / 2, NRG = G::THREADS / NCP;                // column pairs; row groups of the column-sum pass
        const int cp = threadIdx.x % NCP, rg = threadIdx.x / NCP;
        auto flush = [&](TO* dst, int pass) {          // staged rows -> global, full 16-byte lanes along each row
            __syncthreads();
            if constexpr (CS) {
                if (dst == C) {
                    for (int r = rg; r < RPP; r += NRG) {
                        const TO* q = reinterpret_cast<const TO*>(smem + r * ROWB) + cp * 2;
                        csum0 += io<TO>::load1(q);
                        csum1 += io<TO>::load1(q + 1);
                    }
                }
            }
            for (int v = threadIdx.x; v < RPP * VPR; v += G::THREADS) {
                const int lrow = v / VPR, cv = v % VPR;
                const int m = m0 + (lrow / (MIP * 16)) * (G::MI * 16) + pass * MIP * 16 + (lrow % (MIP * 16));
                const int n = n0 + cv * EPV_O;
                if (m < p.M && n < p.N)
                    *reinterpret_cast<uint4*>(dst + (size_t)m * p.ldc + n) = *reinterpret_cast<const uint4*>(smem + lrow * ROWB + cv * 16);
            }
            __syncthreads();
        };
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            __syncthreads();
            if (aux) {      // pre-activation first (cheap: acc * alpha + bias), then the activated values reuse the LDS window
#pragma unroll
                for (int ml = 0; ml < MIP; ++ml) {
                    const int mi = pass * MIP + ml;
                    const int m = acc_row_cfg<G>(m0, mi);
#pragma unroll
                    for (int ni = 0; ni < G::NI; ++ni) {
                        const int n = acc_col_cfg<G>(n0, ni);
                        float pre[4] = {0.f, 0.f, 0.f, 0.f};
                        if (m < p.M && n < p.N) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) pre[r] = acc[mi][ni][r] * p.alpha;
                            if (p.bias) {
                                const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                                pre[0] += b.x; pre[1] += b.y; pre[2] += b.z; pre[3] += b.w;
                            }
                        }
                        if constexpr (ACT == 1 || ACT == 2) {
                            if (p.aux_deriv) {
                                if constexpr (ACT == 1) {
                                    dgelu4(pre);
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) pre[r] = pre[r] > 0.f ? 1.f : 0.f;
                                }
                            }
                        }
                        char* l = smem + (wm * MIP * 16 + ml * 16 + c16) * ROWB + (wn * G::NI * 16 + ni * 16 + g4 * 4) * (int)sizeof(TO);
                        io<TO>::store4(reinterpret_cast<TO*>(l), pre);
                    }
                }
                flush(aux, pass);
            }
#pragma unroll
            for (int ml = 0; ml < MIP; ++ml) {
                const int mi = pass * MIP + ml;
                const int m = acc_row_cfg<G>(m0, mi);
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni) {
                    const int n = acc_col_cfg<G>(n0, ni);
                    float v[4] = {0.f, 0.f, 0.f, 0.f}, pre[4];
                    if (m < p.M && n < p.N) finish(mi, ni, m, n, v, pre);
                    char* l = smem + (wm * MIP * 16 + ml * 16 + c16) * ROWB + (wn * G::NI * 16 + ni * 16 + g4 * 4) * (int)sizeof(TO);
                    io<TO>::store4(reinterpret_cast<TO*>(l), v);
                }
            }
            flush(C, pass);
        }
        if constexpr (CS) {      // fold the NRG row groups through LDS (free again after the last flush's barrier)
            static_assert(NRG * G::TN * 4 <= G::LDS_BYTES, "column-sum scratch does not fit");
            float* red = reinterpret_cast<float*>(smem);
            red[rg * G::TN + cp * 2] = csum0;
            red[rg * G::TN + cp * 2 + 1] = csum1;
            __syncthreads();
            const int n = n0 + threadIdx.x;
            if (threadIdx.x < G::TN && n < p.N) {
                float t = 0.f;
#pragma unroll
                for (int g = 0; g < NRG; ++g) t += red[g * G::TN + threadIdx.x];
                p.colsum[(size_t)tm * p.N + n] = t;
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi) {
        const int m = acc_row_cfg<G>(m0, mi);
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) {
            const int n = acc_col_cfg<G>(n0, ni);
            if (n >= p.N) continue;  // N % 4 == 0 is enforced by the launcher
            float v[4], pre[4];
            finish(mi, ni, m, n, v, pre);
            const size_t off = (size_t)m * p.ldc + n;
            if (aux) io<TO>::store4(aux + off, pre);
            if (p.accumulate == 0) {
                io<TO>::store4(C + off, v);
            } else if (p.accumulate == 1) {
                float c[4];
                io<TO>::load4(C + off, c);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += c[r];
                io<TO>::store4(C + off, v);
            } else {
                if constexpr (sizeof(TO) == 4) {
                    float* cf = reinterpret_cast<float*>(C) + off;
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(cf + r, v[r]);
                }
            }
        }
    }
}


template <typename G, typename TI, typename TO, int ACT, bool CS = false>
inline int launch_gemm_act(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    const int split = d->split_k < 1 ? 1 : d->split_k;
    int kchunk = (d->K + split - 1) / split;
    kchunk = ((kchunk + G::KE - 1) / G::KE) * G::KE;
    a.kchunk = kchunk;
    const int zs = (d->K + kchunk - 1) / kchunk;
    a.tiles_m = (d->M + G::TM - 1) / G::TM;
    a.tiles_n = (d->N + G::TN - 1) / G::TN;
    // one-time attribute set-up behind a function-local static: thread-safe first call (C++11), re-entrant afterwards
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<G, TI, TO, ACT, CS>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    (void)attr_rc;
    dim3 grid(a.tiles_m * a.tiles_n, 1, zs);
    hipLaunchKernelGGL((gemm_nt_kernel<G, TI, TO, ACT, CS>), grid, dim3(G::THREADS), G::LDS_BYTES, s, a);
    MOREC_CHECK_LAUNCH();
    if constexpr (CS) {   // fold the partial rows: [tiles_m * WM (wave epilogue) | tiles_m (block epilogue)] x N -> colsum_dst +=
        const int rows = a.wave_epilogue ? a.tiles_m * G::WM : a.tiles_m;
        return colsum_f32_launch(a.colsum, a.colsum_dst, rows, d->N, s);
    }
    return MOREC_OK;
}

template <typename G, typename TI, typename TO>
inline int launch_gemm_cfg(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    const int mode = d->dact == MOREC_ACT_GELU ? 3 : d->dact == MOREC_ACT_RELU ? 4 : d->dact == MOREC_DACT_MUL ? 5
                     : d->act == MOREC_ACT_GELU ? 1 : d->act == MOREC_ACT_RELU ? 2 : 0;
    {   // (bf16 operands -> fp32 output carries every epilogue as well: the fp32x3 mode's GEMMs, include/morec_hip.h morec_split_bf16x3)
        if (a.colsum) {     // fused bias gradient: only behind the activation-derivative epilogues
            if (mode == 3) return launch_gemm_act<G, TI, TO, 3, true>(d, a, s);
            if (mode == 4) return launch_gemm_act<G, TI, TO, 4, true>(d, a, s);
            if (mode == 5) return launch_gemm_act<G, TI, TO, 5, true>(d, a, s);
            return MOREC_E_UNSUPPORTED;
        }
        switch (mode) {
            case 1: return launch_gemm_act<G, TI, TO, 1>(d, a, s);
            case 2: return launch_gemm_act<G, TI, TO, 2>(d, a, s);
            case 3: return launch_gemm_act<G, TI, TO, 3>(d, a, s);
            case 4: return launch_gemm_act<G, TI, TO, 4>(d, a, s);
            case 5: return launch_gemm_act<G, TI, TO, 5>(d, a, s);
            default: return launch_gemm_act<G, TI, TO, 0>(d, a, s);
        }
    }
}

// 256 x 256 tiles when the problem is large enough to fill the 256 CUs with them, 128 x 128 tiles otherwise
template <typename TI, typename TO>
inline int launch_gemm(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    const long big_tiles = (long)((d->M + 255) / 256) * ((d->N + 255) / 256) * (d->split_k < 1 ? 1 : d->split_k);
    static int force = -1;
    if (force < 0) { const char* e = getenv("MOREC_GEMM_TILE"); force = e ? atoi(e) : 0; }   // 128 / 256: tuning override
    if (force == 128) return launch_gemm_cfg<GemmTile<TI, 2>, TI, TO>(d, a, s);
    if (force == 256) return launch_gemm_cfg<GemmTileCfg<TI, 2, 4, 8, 4>, TI, TO>(d, a, s);
    if (force == 129) return launch_gemm_cfg<GemmTileCfg<TI, 2, 4, 4, 2>, TI, TO>(d, a, s);    // 128 x 128, 8 waves of 64 x 32
    if (force == 2128) return launch_gemm_cfg<GemmTileCfg<TI, 4, 2, 4, 4, 1>, TI, TO>(d, a, s);  // 256 x 128, 8 waves, 2 workgroups per CU
    if (force == 1024) return launch_gemm_cfg<GemmTileCfg<TI, 4, 4, 4, 4>, TI, TO>(d, a, s);   // 256 x 256, 16 waves of 64 x 64
    // narrow outputs (N <= 128: the Swin stage-1 projections) and very short K (<= 96) are streaming problems: a 256-wide
    // tile would be mostly padding / a 3-step main loop, and two independent 128 x 128 workgroups per CU overlap one's loads
    // with the other's stores (measured per shape in scripts/swin_gemm_shapes.py)
    if (d->N <= 128 || d->K <= 96) return launch_gemm_cfg<GemmTileCfg<TI, 2, 4, 4, 2>, TI, TO>(d, a, s);
    // 256 x 256 as 16 waves of 64 x 64 (4 waves per SIMD, 118 VGPRs): 3 - 14 % faster than 8 waves of 128 x 64 on every encoder
    // shape (scripts/gemm_bench.py) -- the extra waves cover each other's LDS / barrier waits
    if (big_tiles >= 192) return launch_gemm_cfg<GemmTileCfg<TI, 4, 4, 4, 4>, TI, TO>(d, a, s);
    return launch_gemm_cfg<GemmTileCfg<TI, 2, 4, 4, 2>, TI, TO>(d, a, s);   // 128 x 128 as 8 waves of 64 x 32 (same reasoning)
}

