// gemm_args.hpp -- kernel argument block shared by the NT GEMM kernels (gemm.hip: two-buffer main loop;
// gemm8p.hip: the 256 x 256 eight-phase main loop).
#pragma once
#include "common.hpp"

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    const float* bias;
    void* aux_out;
    const void* dact_in;
    int M, N, K, lda, ldb, ldc;
    int act, dact, accumulate;
    int aux_deriv;   // aux_out receives act'(pre) instead of pre
    int kchunk;
    int tiles_m, tiles_n;
    float alpha;
    int vec_store;   // output rows are 16-byte addressable: stage the tile through LDS and store full lines
    int wave_epilogue;
    float* colsum_dst;   // host side only: fp32 [N] the partial rows are folded into after the launch
    unsigned long long* stamps;   // gemm8p: per-workgroup s_memtime stamps (debug bit 8), else null
    int debug;       // gemm8p ablation bits (tuning key "gemm8p_debug"; 0 in production)
    float* colsum;   // CS kernels: fp32 workspace [partial rows][N] of per-wave-block / per-tile column sums of C
    float* tail_ws;  // gemm8p tail split: fp32 partial tiles [tail tiles][2 parts][256 x 256], or null (no split)
    int tail_split;  // gemm8p: 1 = split the tail round along K (opt-in)
    int tail_bias;   // gemm8p tail split: the consumer takes (nk + tail_bias) / 2 of the nk K-tiles
    int* tail_cnt;   // gemm8p tail split: one arrival counter per tail tile (zero between launches)
    int ngroup;      // gemm8p: N-tiles per column group of the tile order (0 / >= tiles_n: plain row-major order)
    int tmr;         // gemm8p: rows of an output tile (256, or 224 / 192: the PART instantiations -- see gemm8p.hip "tile height")
};

// gemm8p.hip: runs the launch on the eight-phase kernel when the problem is eligible (returns MOREC_OK / an error) or
// reports "not eligible" with G8_NOT_TAKEN so that the caller falls through to the generic kernels.
#define G8_NOT_TAKEN 0x7fff0001
int gemm8p_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s);
int gemm2w_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s);      // gemm2w.hip: 256 x 128 tiles, two workgroups per CU (epilogue-heavy products)
int gemm_small_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s);       // gemm_small.hip: 64 x 64 tiles, four-stage ring (latency-class products)
int gemm_skinny_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s);      // gemm_skinny.hip: N <= 128, K <= 384, plain product
int gemm_skinny_wide_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s); // gemm_skinny_wide.hip: 288 < N <= 512, K <= 128, bias / GELU
extern int g_skinny_mode;
int gemm_nt_f16_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s);      // gemm_f16.hip
int colsum_f32_launch(const float* in, float* out, int rows, int N, hipStream_t s);
// morec_gemm_tn with a choice of output type for the slab fold (gemm_tn.hip)
int gemm_tn_launch(const void* DY, const void* X, void* C, int c_dtype, int M, int N, int K, int ldy, int ldx, int ldc, int dtype, int split_m,
                   int accumulate, float* workspace, void* stream);
