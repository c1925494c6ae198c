// ce_args.hpp -- argument block, workspace layout and entry points of the eight-phase scoring kernels (inbatch_ce8p.hip), shared with
// the C-ABI launchers in inbatch_ce.hip.
#pragma once
#include "common.hpp"

struct Ce8Args {
    const bf16* P;
    const bf16* E;
    const uint8_t* tab;      // [B][Ncp] cell flags in lane order (inbatch_ce8p.hip: ce8p_prep_kernel)
    const float* lpp;        // [Ncp] log-popularity in lane order
    const uint8_t* row_valid;
    float* pmax;             // fwd: [Nr][K2] partial maxima (base-2 domain), one per 256-column tile
    float* psum;             // fwd: [Nr][K2] partial sum-exp
    float* pos;              // fwd: [Nr] positive logit
    const float* row_lse;    // bwd
    bf16* dlt;               // bwd: dlogit^T [Nc][ldr]
    const float* gscale_dev;
    float gscale;
    int B, S, D, Nr, Nc, col_offset, K2, Ncp, ldr;
    int tiles_m, tiles_n;
};

struct Ce8Layout {           // byte offsets into the caller's workspace
    size_t off_tab, off_lpp, off_pmax, off_psum, off_pos, off_part, fwd_bytes;
    size_t off_dlt, off_pt, off_dp32, off_de32, off_slabs, bwd_bytes;
    int tiles_m, tiles_n, Ncp, K2, ldr, tn_split;
};

// 0: automatic (large enough bf16 problems), 1: never, 2: wherever the shape rules allow (tests)
extern int g_ce8p_mode;
bool ce8p_eligible(const morec_ce_desc* d);
void ce8p_layout(const morec_ce_desc* d, Ce8Layout& L);
int ce8p_fwd(const morec_ce_desc* d, const void* P, const void* E, const int32_t* row_ids, const int32_t* col_ids, const float* col_logpop,
             const uint8_t* col_valid, const uint8_t* row_valid, void* workspace, float** pmax, float** psum, float** pos, float** part, int* K2, hipStream_t s);
// morec_gemm_tn with a choice of output type for the slab fold (gemm_tn.hip)
int gemm_tn_launch(const void* DY, const void* X, void* C, int c_dtype, int M, int N, int K, int ldy, int ldx, int ldc, int dtype, int split_m,
                   int accumulate, float* workspace, void* stream);
int ce8p_bwd(const morec_ce_desc* d, const void* P, const void* E, const int32_t* row_ids, const int32_t* col_ids, const float* col_logpop,
             const uint8_t* col_valid, const uint8_t* row_valid, const float* row_lse, const float* gscale_dev, float gscale, void* dP, void* dE,
             void* workspace, hipStream_t s);
