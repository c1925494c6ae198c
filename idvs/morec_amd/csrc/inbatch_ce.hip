// inbatch_ce.hip -- fused in-batch debiased sampled-softmax cross-entropy (the "scoring kernel").
//
// Reference (T/model/model.py:32-33,45-67): logits = P.E^T - log(pop[ids]) are materialised as an
// [B*S, B*(S+1)] fp32 matrix, masked by a Python double loop over users (one boolean compare of size
// (S+1) x S*Nc per user), then fed to nn.CrossEntropyLoss.  Here each 128 x 128 logit tile lives only in
// MFMA accumulators: the tile is produced by the shared GEMM main loop, the column-validity mask, the
// popularity correction and the per-user id-membership reject mask (positive restored) are applied in
// registers, and only per-(row, 64-column) softmax partials (max, sum-exp) leave the workgroup.
//
// Masking rules, bit-exact with the reference's bookkeeping:
//   column c invalid (padding slot)                          -> -1e4          (model.py:51-52)
//   ids[c] in {ids of user u's S+1 slots} and c != label(r)  -> -1e4          (model.py:54-63)
//   label(r = u*S + j) = col_offset + u*(S+1) + j + 1                         (model.py:45-48)
// Masked cells keep the VALUE -1e4 inside the softmax (exactly as the reference) and get zero gradient.
#include "gemm_core.hpp"
#include "ce_args.hpp"

namespace {
constexpr float MASKED_LOGIT = -1e4f;
constexpr int MAX_TILE_USERS = 130;  // users overlapped by a 128-row tile when S >= 1

struct CeArgs {
    const void* P;
    const void* E;
    const int32_t* row_ids;
    const int32_t* col_ids;
    const float* col_logpop;
    const uint8_t* col_valid;
    const uint8_t* row_valid;
    float* pmax;      // fwd: [Nr][K2] partial maxima         bwd: unused
    float* psum;      // fwd: [Nr][K2] partial sum-exp
    float* pos;       // fwd: [Nr] positive logit
    const float* row_lse;   // bwd
    void* dl;         // bwd: [Nr][ld_dl] dlogits (dtype T)
    const float* gscale_dev;
    float gscale;
    int B, S, D, Nr, Nc, col_offset, K2, ld_dl, ldp, lde;
    int tiles_m, tiles_n;
};

// Turns the accumulator tile into masked logits in place.  Uses LDS (free after the main loop) for the
// tile's user id lists, column ids and the [user][column] membership bytes.
// Returns through `lab[mi]` the label column of each of the lane's 4 rows (or -1 when the row is out of range).
template <typename T>
__device__ __forceinline__ void masked_logits(const CeArgs& p, int m0, int n0, char* smem, f32x4_t (&acc)[4][4],
                                              int (&lab)[4]) {
    int32_t* s_uid = reinterpret_cast<int32_t*>(smem);                       // [nu][S+1]
    const int S1 = p.S + 1;
    const int u_lo = m0 / p.S;
    const int u_hi = min(p.B - 1, (min(m0 + 127, p.Nr - 1)) / p.S);
    const int nu = u_hi - u_lo + 1;
    int32_t* s_cid = s_uid + ((nu * S1 + 3) & ~3);                           // [128]
    uint8_t* s_mem = reinterpret_cast<uint8_t*>(s_cid + 128);               // [nu][128]
    // the tile's 128 columns' log-popularity and validity, staged once (coalesced) instead of two guarded global loads
    // per logit element -- 128 dependent L2 round trips per lane, most of this kernel's time at these sizes
    float* s_lp = reinterpret_cast<float*>(s_mem + ((nu * 128 + 15) & ~15));   // [128]
    uint8_t* s_cv = reinterpret_cast<uint8_t*>(s_lp + 128);                   // [128]
    const int tid = threadIdx.x;
    for (int i = tid; i < nu * S1; i += 256) s_uid[i] = p.row_ids[u_lo * S1 + i];
    if (tid < 128) {
        const int c = min(n0 + tid, p.Nc - 1);          // clamped: the three loads need no guard
        const bool in = n0 + tid < p.Nc;
        const int32_t id = p.col_ids[c];
        const float lp = p.col_logpop[c];
        const uint8_t cv = p.col_valid[c];
        s_cid[tid] = in ? id : -1;
        s_lp[tid] = in ? lp : 0.f;
        s_cv[tid] = in ? cv : 0;
    }
    __syncthreads();
    for (int i = tid; i < nu * 128; i += 256) {
        const int u = i >> 7, c = i & 127;
        const int32_t id = s_cid[c];
        bool hit = false;
        for (int k = 0; k < S1; ++k) hit |= (s_uid[u * S1 + k] == id);
        s_mem[i] = (hit && n0 + c < p.Nc) ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = acc_row(m0, mi);
        lab[mi] = -1;
        if (m >= p.Nr) continue;
        const int u = m / p.S, j = m - u * p.S;
        lab[mi] = p.col_offset + u * S1 + j + 1;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = acc_col(n0, ni);
            const uint32_t mem4 = *reinterpret_cast<const uint32_t*>(s_mem + (u - u_lo) * 128 + (n - n0));
            const uint32_t cv4 = *reinterpret_cast<const uint32_t*>(s_cv + (n - n0));
            const float4 lp4 = *reinterpret_cast<const float4*>(s_lp + (n - n0));
            const float lpr[4] = {lp4.x, lp4.y, lp4.z, lp4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = n + r;
                float v = -INFINITY;   // columns past the pool do not exist
                if (c < p.Nc) {
                    const bool rejected = ((mem4 >> (8 * r)) & 1u) && (c != lab[mi]);
                    v = (((cv4 >> (8 * r)) & 0xffu) == 0 || rejected) ? MASKED_LOGIT : acc[mi][ni][r] - lpr[r];
                }
                acc[mi][ni][r] = v;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(CeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int m0 = (wg / p.tiles_n) * 128, n0 = (wg % p.tiles_n) * 128;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    gemm_mainloop<T, 2>(reinterpret_cast<const T*>(p.P), reinterpret_cast<const T*>(p.E), p.Nr, p.Nc, p.ldp, p.lde, m0,
                        n0, 0, p.D, smem, acc);
    int lab[4];
    masked_logits<T>(p, m0, n0, smem, acc, lab);
    const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
    const int half = (wg % p.tiles_n) * 2 + wn;   // which 64-column slice of the pool this wave covered
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = acc_row(m0, mi);
        float mx = -INFINITY;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[mi][ni][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sm = 0.f;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = acc_col(n0, ni);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[mi][ni][r];
                if (mx > -INFINITY) sm += expf(v - mx);   // v == -inf (no such column) contributes 0
                if (m < p.Nr && n + r == lab[mi]) p.pos[m] = v;
            }
        }
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        if (m < p.Nr && (lane >> 4) == 0) {
            p.pmax[(size_t)m * p.K2 + half] = mx;
            p.psum[(size_t)m * p.K2 + half] = sm;
        }
    }
}

// one wave per row: merge the K2 partials -> lse, loss; the valid rows' losses of a block (4 rows) go to block_part[block], and
// ce_loss_total adds them up IN A FIXED ORDER (the first version did one fp32 atomicAdd per block into loss_sum: 640 same-address
// atomics at B = 128 -- most of the kernel's time -- and a total that changed in the last bit from run to run)
__global__ __launch_bounds__(256) void ce_combine_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                         const float* __restrict__ pos,
                                                         const uint8_t* __restrict__ row_valid,
                                                         float* __restrict__ row_lse, float* __restrict__ row_loss,
                                                         float* __restrict__ block_part, int Nr, int K2, int log2_domain) {
    __shared__ float s_part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    float loss = 0.f;
    if (row < Nr) {
        float mx = -INFINITY;
        for (int k = lane; k < K2; k += 64) mx = fmaxf(mx, pmax[(size_t)row * K2 + k]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int k = lane; k < K2; k += 64) {
            const float pm = pmax[(size_t)row * K2 + k];
            if (pm > -INFINITY) sm += psum[(size_t)row * K2 + k] * (log2_domain ? exp2f(pm - mx) : expf(pm - mx));
        }
        sm = wave_sum(sm);
        // log2_domain (the eight-phase scoring kernels): maxima and sums of 2^(x log2 e) -> lse = ln 2 (max2 + log2 sum)
        const float lse = log2_domain ? 0.6931471805599453f * (mx + log2f(sm)) : mx + logf(sm);
        loss = row_valid[row] ? (lse - pos[row]) : 0.f;
        if (lane == 0) {
            row_lse[row] = lse;
            row_loss[row] = loss;
        }
    }
    if (lane == 0) s_part[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) block_part[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}
// loss_sum += sum of the block partials: one block, fixed summation tree -> bit-identical totals from run to run
__global__ __launch_bounds__(256) void ce_loss_total_kernel(const float* __restrict__ block_part, int n, float* __restrict__ loss_sum) {
    __shared__ float s[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += block_part[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_sum += s[0];
}

// backward, stage 1: recompute the logit tile and emit dlogits = g * (softmax - onehot) for the unmasked
// cells of valid rows (0 elsewhere), written once as dtype T; the two products dP = dl.E and dE = dl^T.P
// then run on the shared GEMM kernel.
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_dl_kernel(CeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int m0 = (wg / p.tiles_n) * 128, n0 = (wg % p.tiles_n) * 128;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    gemm_mainloop<T, 2>(reinterpret_cast<const T*>(p.P), reinterpret_cast<const T*>(p.E), p.Nr, p.Nc, p.ldp, p.lde, m0,
                        n0, 0, p.D, smem, acc);
    int lab[4];
    masked_logits<T>(p, m0, n0, smem, acc, lab);
    const float g = p.gscale * (p.gscale_dev ? *p.gscale_dev : 1.0f);
    T* dl = reinterpret_cast<T*>(p.dl);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = acc_row(m0, mi);
        if (m >= p.Nr) continue;
        const float w = p.row_valid[m] ? g : 0.f;
        const float lse = p.row_lse[m];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = acc_col(n0, ni);
            if (n >= p.ld_dl) continue;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[mi][ni][r];
                float d = 0.f;
                // cells overwritten with -1e4 receive no gradient (index_put semantics); their softmax mass is
                // exp(-1e4 - lse) == 0 in fp32 whenever the row has a finite positive
                if (n + r < p.Nc && v != MASKED_LOGIT) d = expf(v - lse) - ((n + r == lab[mi]) ? 1.f : 0.f);
                o[r] = w * d;
            }
            io<T>::store4(dl + (size_t)m * p.ld_dl + n, o);
        }
    }
}

inline int pad8(int x) { return (x + 7) & ~7; }
}  // namespace

// workspace layout:
//   fwd:  pmax f32[Nr*K2] | psum f32[Nr*K2] | pos f32[Nr]
//   bwd (fp32 mode):  dl T[Nr*ldc] | dlt T[Nc*ldr] | Pt T[D*ldr] | Et T[D*ldc]          (ldc = pad8(Nc), ldr = pad8(Nr))
//   bwd (bf16 mode):  dl T[Nr*ldc] | Et T[D*ldc] | dE32 f32[Nc*D] (unless the caller takes dE in fp32) | slabs f32[split*Nc*D]
static int ce_tn_split(int Nr, int Nc, int D) {     // token (row) split of dE = dl^T P so that the launch has ~256 workgroups
    const int tiles = ((Nc + 255) / 256) * ((D + 255) / 256);
    int s = (256 + tiles - 1) / tiles;
    const int max_s = Nr / 256 > 0 ? Nr / 256 : 1;   // at least 4 stages of 64 rows per chunk
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}
extern "C" size_t morec_inbatch_ce_workspace_bytes(const morec_ce_desc* d) {
    if (!d) return 0;
    const size_t Nr = (size_t)d->B * d->S, Nc = d->Nc, D = d->D;
    const size_t K2 = 2 * ((Nc + 127) / 128);
    const size_t fwd = (2 * Nr * K2 + Nr + (Nr + 3) / 4 + 4) * sizeof(float);       // pmax | psum | pos | per-block loss partials
    const size_t es = elt_size(d->dtype);
    const size_t ldc = pad8((int)Nc), ldr = pad8((int)Nr);
    size_t bwd = (Nr * ldc + Nc * ldr + D * ldr + D * ldc) * es + 256;
    if (is_h16(d->dtype) && Nc % 8 == 0 && D % 8 == 0) {
        const size_t tn = (Nr * ldc + D * ldc) * es + Nc * D * sizeof(float) + (size_t)ce_tn_split((int)Nr, (int)Nc, (int)D) * Nc * D * sizeof(float) + 1024;
        bwd = tn > bwd ? tn : bwd;
    }
    size_t need = (fwd > bwd ? fwd : bwd) + 256;
    if (ce8p_eligible(d)) {      // the eight-phase scoring kernels (inbatch_ce8p.hip) lay the workspace out differently
        Ce8Layout L;
        ce8p_layout(d, L);
        if (L.fwd_bytes + 256 > need) need = L.fwd_bytes + 256;
        if (L.bwd_bytes + 256 > need) need = L.bwd_bytes + 256;
    }
    return need;
}

static int ce_fill(const morec_ce_desc* d, CeArgs& a) {
    if (!d || d->B <= 0 || d->S <= 0 || d->D <= 0 || d->Nc <= 0) return MOREC_E_ARG;
    if ((d->D * elt_size(d->dtype)) % 16) return MOREC_E_ALIGN;
    if (d->dtype != MOREC_F32 && !is_h16(d->dtype)) return MOREC_E_DTYPE;
    if (d->col_offset < 0 || d->col_offset + d->B * (d->S + 1) > d->Nc) return MOREC_E_ARG;
    a.B = d->B; a.S = d->S; a.D = d->D; a.Nr = d->B * d->S; a.Nc = d->Nc; a.col_offset = d->col_offset;
    a.tiles_m = (a.Nr + 127) / 128; a.tiles_n = (a.Nc + 127) / 128; a.K2 = 2 * a.tiles_n;
    a.ldp = d->D; a.lde = d->D;
    return MOREC_OK;
}

extern "C" int morec_inbatch_ce_fwd(const morec_ce_desc* d, const void* P, const void* E, const int32_t* row_ids,
                                    const int32_t* col_ids, const float* col_logpop, const uint8_t* col_valid,
                                    const uint8_t* row_valid, float* row_lse, float* row_loss, float* loss_sum,
                                    void* workspace, void* stream) {
    CeArgs a{};
    int rc = ce_fill(d, a);
    if (rc) return rc;
    if (!P || !E || !row_ids || !col_ids || !col_logpop || !col_valid || !row_valid || !row_lse || !row_loss ||
        !loss_sum || !workspace)
        return MOREC_E_ARG;
    if (!aligned16(P) || !aligned16(E) || !aligned16(workspace)) return MOREC_E_ALIGN;
    a.P = P; a.E = E; a.row_ids = row_ids; a.col_ids = col_ids; a.col_logpop = col_logpop; a.col_valid = col_valid;
    a.row_valid = row_valid;
    float* ws = reinterpret_cast<float*>(workspace);
    a.pmax = ws; a.psum = ws + (size_t)a.Nr * a.K2; a.pos = ws + 2 * (size_t)a.Nr * a.K2;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(a.tiles_m * a.tiles_n);
    int log2_domain = 0;
    float* part8 = nullptr;
    if (ce8p_eligible(d)) {      // 256 x 256 tiles on the eight-phase main loop; same partial format (one entry per 64-column slice)
        if (!aligned16(col_ids) || !aligned16(col_logpop) || (reinterpret_cast<uintptr_t>(col_valid) & 3u)) return MOREC_E_ALIGN;
        rc = ce8p_fwd(d, P, E, row_ids, col_ids, col_logpop, col_valid, row_valid, workspace, &a.pmax, &a.psum, &a.pos, &part8, &a.K2, s);
        if (rc) return rc;
        log2_domain = 1;
    } else {
        by_dtype(d->dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            using G = GemmTile<T, 2>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ce_fwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
            hipLaunchKernelGGL((ce_fwd_kernel<T>), grid, dim3(256), G::LDS_BYTES, s, a);
        });
    }
    MOREC_CHECK_LAUNCH();
    const int n_blocks = (a.Nr + 3) / 4;
    float* block_part = part8 ? part8 : a.pos + a.Nr;
    hipLaunchKernelGGL(ce_combine_kernel, dim3(n_blocks), dim3(256), 0, s, a.pmax, a.psum, a.pos, row_valid, row_lse, row_loss, block_part,
                       a.Nr, a.K2, log2_domain);
    MOREC_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_loss_total_kernel, dim3(1), dim3(256), 0, s, block_part, n_blocks, loss_sum);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_inbatch_ce_bwd(const morec_ce_desc* d, const void* P, const void* E, const int32_t* row_ids,
                                    const int32_t* col_ids, const float* col_logpop, const uint8_t* col_valid,
                                    const uint8_t* row_valid, const float* row_lse, const float* gscale_dev,
                                    float gscale, void* dP, void* dE, void* workspace, void* stream) {
    CeArgs a{};
    int rc = ce_fill(d, a);
    if (rc) return rc;
    if (!P || !E || !row_ids || !col_ids || !col_logpop || !col_valid || !row_valid || !row_lse || !dP || !dE ||
        !workspace)
        return MOREC_E_ARG;
    if (!aligned16(P) || !aligned16(E) || !aligned16(workspace) || !aligned16(dP) || !aligned16(dE))
        return MOREC_E_ALIGN;
    a.P = P; a.E = E; a.row_ids = row_ids; a.col_ids = col_ids; a.col_logpop = col_logpop; a.col_valid = col_valid;
    a.row_valid = row_valid; a.row_lse = row_lse; a.gscale_dev = gscale_dev; a.gscale = gscale;
    if (ce8p_eligible(d)) {
        if (!aligned16(col_ids) || !aligned16(col_logpop) || (reinterpret_cast<uintptr_t>(col_valid) & 3u)) return MOREC_E_ALIGN;
        return ce8p_bwd(d, P, E, row_ids, col_ids, col_logpop, col_valid, row_valid, row_lse, gscale_dev, gscale, dP, dE, workspace,
                        reinterpret_cast<hipStream_t>(stream));
    }
    const int es = elt_size(d->dtype);
    const int ldc = pad8(a.Nc), ldr = pad8(a.Nr);
    char* ws = reinterpret_cast<char*>(workspace);
    char* dl = ws;
    char* dlt = dl + (((size_t)a.Nr * ldc * es + 15) & ~(size_t)15);
    char* Pt = dlt + (((size_t)a.Nc * ldr * es + 15) & ~(size_t)15);
    char* Et = Pt + (((size_t)a.D * ldr * es + 15) & ~(size_t)15);
    a.dl = dl; a.ld_dl = ldc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(a.tiles_m * a.tiles_n);
    by_dtype(d->dtype, [&](auto* t) {
        using T = MOREC_TAG_T(t);
        using G = GemmTile<T, 2>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ce_bwd_dl_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        hipLaunchKernelGGL((ce_bwd_dl_kernel<T>), grid, dim3(256), G::LDS_BYTES, s, a);
    });
    MOREC_CHECK_LAUNCH();
    morec_gemm_desc g{};
    g.in_dtype = d->dtype; g.out_dtype = d->dtype; g.alpha = 1.0f; g.split_k = 1;
    if (is_h16(d->dtype) && a.Nc % 8 == 0 && a.D % 8 == 0) {
        // dE[Nc, D] = dl^T . P straight from the row-major dl and P (transposing LDS reads, morec_gemm_tn): no transposed copies of the
        // Nr x Nc matrix, fp32 result -- handed out as it is when the caller reduces it over ranks (dE_fp32), else cast
        char* Et2 = dl + (((size_t)a.Nr * ldc * es + 15) & ~(size_t)15);
        float* dE32 = reinterpret_cast<float*>(Et2 + (((size_t)a.D * ldc * es + 15) & ~(size_t)15));
        float* slabs = dE32 + (((size_t)a.Nc * a.D + 3) & ~(size_t)3);
        float* dEo = d->dE_fp32 ? reinterpret_cast<float*>(dE) : dE32;
        const int split = ce_tn_split(a.Nr, a.Nc, a.D);
        if (split > 1) (void)hipMemsetAsync(dEo, 0, (size_t)a.Nc * a.D * sizeof(float), s);
        rc = morec_gemm_tn(dl, P, dEo, a.Nr, a.Nc, a.D, ldc, a.D, a.D, d->dtype, split, split > 1 ? 1 : 0, split > 1 ? slabs : nullptr, stream);
        if (rc) return rc;
        if (!d->dE_fp32) {
            rc = morec_cast(dE32, dE, (size_t)a.Nc * a.D, MOREC_F32, d->dtype, stream);
            if (rc) return rc;
        }
        // dP[Nr, D] = dl[Nr, Nc] . Et[D, Nc]^T
        if (ldc != a.Nc) (void)hipMemsetAsync(Et2, 0, (size_t)a.D * ldc * es, s);
        rc = morec_transpose(E, Et2, a.Nc, a.D, a.D, ldc, d->dtype, d->dtype, stream);
        if (rc) return rc;
        g.M = a.Nr; g.N = a.D; g.K = ldc; g.lda = ldc; g.ldb = ldc; g.ldc = a.D;
        return morec_gemm_nt(&g, dl, Et2, dP, nullptr, nullptr, nullptr, stream);
    }
    if (d->dE_fp32) return MOREC_E_UNSUPPORTED;
    if (ldr != a.Nr) {  // zero the pad columns the transposes do not touch
        (void)hipMemsetAsync(dlt, 0, (size_t)a.Nc * ldr * es, s);
        (void)hipMemsetAsync(Pt, 0, (size_t)a.D * ldr * es, s);
    }
    if (ldc != a.Nc) (void)hipMemsetAsync(Et, 0, (size_t)a.D * ldc * es, s);
    rc = morec_transpose(dl, dlt, a.Nr, a.Nc, ldc, ldr, d->dtype, d->dtype, stream);
    if (rc) return rc;
    rc = morec_transpose(P, Pt, a.Nr, a.D, a.D, ldr, d->dtype, d->dtype, stream);
    if (rc) return rc;
    rc = morec_transpose(E, Et, a.Nc, a.D, a.D, ldc, d->dtype, d->dtype, stream);
    if (rc) return rc;
    // dP[Nr, D] = dl[Nr, Nc] . Et[D, Nc]^T
    g.M = a.Nr; g.N = a.D; g.K = ldc; g.lda = ldc; g.ldb = ldc; g.ldc = a.D;
    rc = morec_gemm_nt(&g, dl, Et, dP, nullptr, nullptr, nullptr, stream);
    if (rc) return rc;
    // dE[Nc, D] = dlt[Nc, Nr] . Pt[D, Nr]^T
    g.M = a.Nc; g.N = a.D; g.K = ldr; g.lda = ldr; g.ldb = ldr; g.ldc = a.D;
    return morec_gemm_nt(&g, dlt, Pt, dE, nullptr, nullptr, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// BCE variant (bce_text/main-end2end/model/model.py:30-51): one sampled negative per position.
//   E [B, S+1, 2, D]: item vectors, pos at [:, :, 0], neg at [:, :, 1];  P [B, S, D] user states.
//   row (b, j): pos = P . E[b, j+1, 0], neg = P . E[b, j, 1];
//   loss = mean_valid(softplus(-pos)) + mean_valid(softplus(neg))          (two BCEWithLogitsLoss means over the same rows)
// One wavefront per row; HBM-bound (reads P, two E rows; backward writes dP and every dE row exactly once, no atomics).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

template <typename T>
__global__ __launch_bounds__(256) void bce_fwd_kernel(const T* __restrict__ P, const T* __restrict__ E,
                                                      const uint8_t* __restrict__ row_valid, float* __restrict__ scores,
                                                      float* __restrict__ loss_sum, int B, int S, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * S) return;
    const int b = row / S, j = row - b * S;
    const T* p = P + (size_t)row * D;
    const T* ep = E + ((size_t)(b * (S + 1) + j + 1) * 2 + 0) * D;
    const T* en = E + ((size_t)(b * (S + 1) + j) * 2 + 1) * D;
    float sp = 0.f, sn = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float x[4], y[4], z[4];
        io<T>::load4(p + c, x); io<T>::load4(ep + c, y); io<T>::load4(en + c, z);
#pragma unroll
        for (int k = 0; k < 4; ++k) { sp = fmaf(x[k], y[k], sp); sn = fmaf(x[k], z[k], sn); }
    }
    sp = wave_sum(sp); sn = wave_sum(sn);
    if (lane == 0) {
        scores[row] = sp;
        scores[B * S + row] = sn;
        if (row_valid[row]) atomicAdd(loss_sum, softplus_f(-sp) + softplus_f(sn));
    }
}

// dpos = -sigmoid(-pos) g, dneg = sigmoid(neg) g on valid rows (g = dloss / n_valid), 0 elsewhere
__device__ __forceinline__ void bce_coeffs(const float* scores, const uint8_t* row_valid, int BS, int row, float g, float& dpos, float& dneg) {
    if (row < 0 || !row_valid[row]) { dpos = 0.f; dneg = 0.f; return; }
    dpos = -g / (1.f + expf(scores[row]));
    dneg = g / (1.f + expf(-scores[BS + row]));
}

template <typename T>
__global__ __launch_bounds__(256) void bce_bwd_kernel(const T* __restrict__ P, const T* __restrict__ E, const uint8_t* __restrict__ row_valid,
                                                      const float* __restrict__ scores, const float* __restrict__ gscale,
                                                      T* __restrict__ dP, T* __restrict__ dE, int B, int S, int D) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave per (b, slot j in 0..S): writes dE[b, j, 0], dE[b, j, 1] and dP[b, j]
    if (w >= B * (S + 1)) return;
    const int b = w / (S + 1), j = w - b * (S + 1);
    const float g = gscale[0];
    const int BS = B * S;
    float dp_prev, dn_prev, dp_cur, dn_cur;
    bce_coeffs(scores, row_valid, BS, j >= 1 ? b * S + j - 1 : -1, g, dp_prev, dn_prev);   // row (b, j-1) scores E[b, j, 0] as its positive
    bce_coeffs(scores, row_valid, BS, j < S ? b * S + j : -1, g, dp_cur, dn_cur);          // row (b, j) scores E[b, j, 1] as its negative
    T* de_pos = dE + ((size_t)w * 2 + 0) * D;
    T* de_neg = dE + ((size_t)w * 2 + 1) * D;
    const T* p_prev = P + (size_t)(b * S + (j >= 1 ? j - 1 : 0)) * D;
    const T* p_cur = P + (size_t)(b * S + (j < S ? j : 0)) * D;
    const T* ep = E + ((size_t)(b * (S + 1) + (j < S ? j + 1 : 0)) * 2 + 0) * D;           // positive of row (b, j)
    const T* en = E + ((size_t)w * 2 + 1) * D;
    for (int c = lane * 4; c < D; c += 256) {
        float a[4], q[4], y[4], z[4], o[4];
        io<T>::load4(p_prev + c, a); io<T>::load4(p_cur + c, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = dp_prev * a[k];
        io<T>::store4(de_pos + c, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = dn_cur * q[k];
        io<T>::store4(de_neg + c, o);
        if (j < S) {
            io<T>::load4(ep + c, y); io<T>::load4(en + c, z);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = dp_cur * y[k] + dn_cur * z[k];
            io<T>::store4(dP + (size_t)(b * S + j) * D + c, o);
        }
    }
}
}  // namespace

extern "C" int morec_bce_fwd(const void* P, const void* E, const uint8_t* row_valid, float* scores, float* loss_sum, int B, int S, int D,
                             int dtype, void* stream) {
    if (!P || !E || !row_valid || !scores || !loss_sum || B <= 0 || S <= 0 || D <= 0) return MOREC_E_ARG;
    if (D % 4) return MOREC_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((B * S + 3) / 4);
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((bce_fwd_kernel<T>), grid, dim3(256), 0, s, (const T*)P, (const T*)E, row_valid, scores, loss_sum, B, S, D);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_bce_bwd(const void* P, const void* E, const uint8_t* row_valid, const float* scores, const float* gscale, void* dP,
                             void* dE, int B, int S, int D, int dtype, void* stream) {
    if (!P || !E || !row_valid || !scores || !gscale || !dP || !dE || B <= 0 || S <= 0 || D <= 0) return MOREC_E_ARG;
    if (D % 4) return MOREC_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((B * (S + 1) + 3) / 4);
    if (!by_dtype(dtype, [&](auto* t) {
            using T = MOREC_TAG_T(t);
            hipLaunchKernelGGL((bce_bwd_kernel<T>), grid, dim3(256), 0, s, (const T*)P, (const T*)E, row_valid, scores, gscale, (T*)dP, (T*)dE, B, S, D);
        }))
        return MOREC_E_DTYPE;
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
