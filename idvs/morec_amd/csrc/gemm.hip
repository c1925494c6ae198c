// gemm.hip -- NT GEMM with fused epilogues (bias, GELU/ReLU, activation-derivative, split-K atomics),
// plus the layout helpers the backward pass needs (transpose with conversion, cast, column sums).
// Reference arithmetic replaced: see include/morec_hip.h (morec_gemm_nt).
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
    // block): one partial row of p.colsum per 64-row wave block (wave epilogue) or per tile (block epilogue), each written
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
static int launch_gemm_act(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
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
static int launch_gemm_cfg(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
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
static int launch_gemm(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
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

extern "C" int morec_gemm_nt(const morec_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                             void* aux_out, const void* dact_in, void* stream) {
    return morec_gemm_nt_colsum(d, A, B, C, bias, aux_out, dact_in, nullptr, nullptr, stream);
}

extern "C" size_t morec_gemm_colsum_workspace_bytes(int M, int N) {
    return (size_t)((M + 63) / 64) * (size_t)N * sizeof(float);      // one partial row per 64-row wave block at most
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
    {   // bf16, large: the 256 x 256 eight-phase kernel (gemm8p.hip)
        const int r8 = gemm8p_try_launch(d, a, s);
        if (r8 != G8_NOT_TAKEN) return r8;
    }
    if (d->in_dtype == MOREC_F32 && d->out_dtype == MOREC_F32) return launch_gemm<float, float>(d, a, s);
    if (d->in_dtype == MOREC_BF16 && d->out_dtype == MOREC_BF16) return launch_gemm<bf16, bf16>(d, a, s);
    if (d->in_dtype == MOREC_BF16 && d->out_dtype == MOREC_F32) return launch_gemm<bf16, float>(d, a, s);
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
    if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((transpose4_batch_kernel<bf16>), dim3(n_tiles), dim3(256), 0, s, items, n_items);
    else if (dtype == MOREC_F32)
        hipLaunchKernelGGL((transpose4_batch_kernel<float>), dim3(n_tiles), dim3(256), 0, s, items, n_items);
    else
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
    if (vec) {
        if (in_dtype == MOREC_F32 && out_dtype == MOREC_F32)
            hipLaunchKernelGGL((transpose4_kernel<float, float>), grid, dim3(256), 0, s, (const float*)in, (float*)out, R, C, ld_in, ld_out);
        else if (in_dtype == MOREC_F32 && out_dtype == MOREC_BF16)
            hipLaunchKernelGGL((transpose4_kernel<float, bf16>), grid, dim3(256), 0, s, (const float*)in, (bf16*)out, R, C, ld_in, ld_out);
        else if (in_dtype == MOREC_BF16 && out_dtype == MOREC_BF16)
            hipLaunchKernelGGL((transpose4_kernel<bf16, bf16>), grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, R, C, ld_in, ld_out);
        else if (in_dtype == MOREC_BF16 && out_dtype == MOREC_F32)
            hipLaunchKernelGGL((transpose4_kernel<bf16, float>), grid, dim3(256), 0, s, (const bf16*)in, (float*)out, R, C, ld_in, ld_out);
        else
            return MOREC_E_DTYPE;
        MOREC_CHECK_LAUNCH();
        return MOREC_OK;
    }
    if (in_dtype == MOREC_F32 && out_dtype == MOREC_F32)
        hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, s, (const float*)in, (float*)out, R, C, ld_in, ld_out);
    else if (in_dtype == MOREC_F32 && out_dtype == MOREC_BF16)
        hipLaunchKernelGGL((transpose_kernel<float, bf16>), grid, dim3(256), 0, s, (const float*)in, (bf16*)out, R, C, ld_in, ld_out);
    else if (in_dtype == MOREC_BF16 && out_dtype == MOREC_BF16)
        hipLaunchKernelGGL((transpose_kernel<bf16, bf16>), grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, R, C, ld_in, ld_out);
    else if (in_dtype == MOREC_BF16 && out_dtype == MOREC_F32)
        hipLaunchKernelGGL((transpose_kernel<bf16, float>), grid, dim3(256), 0, s, (const bf16*)in, (float*)out, R, C, ld_in, ld_out);
    else
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
    if (in_dtype == MOREC_F32 && out_dtype == MOREC_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(blocks), dim3(256), 0, s, (const float*)in, (bf16*)out, n4);
    else if (in_dtype == MOREC_BF16 && out_dtype == MOREC_F32)
        hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(blocks), dim3(256), 0, s, (const bf16*)in, (float*)out, n4);
    else if (in_dtype == MOREC_F32 && out_dtype == MOREC_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, s, (const float*)in, (float*)out, n4);
    else if (in_dtype == MOREC_BF16 && out_dtype == MOREC_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(blocks), dim3(256), 0, s, (const bf16*)in, (bf16*)out, n4);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// column sums: each block reduces a [rows_per_block x 64-column] slab with 4-element vector loads (16 column
// groups x 16 row lanes), folds the row lanes through LDS and leaves ONE atomicAdd per column per block
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, float* __restrict__ out, int M, int N,
                                                     int ld, int rows_per_block) {
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
            atomicAdd(out + c, t);
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
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)in, out, M, N, ld, rpb);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((colsum_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)in, out, M, N, ld, rpb);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
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
    if (dtype == MOREC_F32)
        hipLaunchKernelGGL((act_bwd_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)dy, (const float*)pre, (float*)out, n4, act);
    else if (dtype == MOREC_BF16)
        hipLaunchKernelGGL((act_bwd_kernel<bf16>), dim3(blocks), dim3(256), 0, s, (const bf16*)dy, (const bf16*)pre, (bf16*)out, n4, act);
    else
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
