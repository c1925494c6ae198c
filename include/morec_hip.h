/*
 * morec_hip.h -- C-ABI of libmorec_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * end-to-end MoRec in-batch training step (westlake-repl/IDvs.MoRec, inbatch_sasrec_e2e_text).
 *
 * The reference is pure Python over PyTorch/HuggingFace and has NO FFI of its own:
 * every entry point below therefore cites the reference Python call site whose arithmetic it
 * replaces (paths relative to /root/reference, T/ = inbatch_sasrec_e2e_text/), and INTEGRATION.md
 * shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless named h_*;
 *   - the caller owns every buffer (PyTorch caching allocator); every call is asynchronous on `stream` (a hipStream_t passed as
 *     void*) and does no host synchronisation;
 *   - return 0 on success, a negative MOREC_E_* for bad arguments, a positive value = hipError_t;
 *   - dtype codes: MOREC_F32 = 0 (exact-fp32 MFMA, v_mfma_f32_16x16x4_f32), MOREC_BF16 = 1 (bf16 operands, fp32 accumulate:
 *     v_mfma_f32_32x32x16_bf16 in the 256 x 256 eight-phase GEMMs, v_mfma_f32_16x16x32_bf16 in attention, scoring and the
 *     small-problem GEMMs), MOREC_F16 = 2 (IEEE fp16 operands, fp32 accumulate: v_mfma_f32_32x32x16_f16 / 16x16x32_f16 in the same
 *     kernels -- the arithmetic of the reference's `torch.cuda.amp.autocast()` step, T/run.py:242-247, three more significand bits
 *     than bf16 at the same MFMA rate; gradients need the loss scaling of morec_step_params below).  "bf16" in an entry point's
 *     description means either 16-bit type unless it says otherwise (the Swin kernels too, V/run.py's autocast step); morec_split_bf16x3
 *     takes MOREC_BF16 only;
 *   - row-major matrices with explicit leading dimensions in ELEMENTS; every base pointer and
 *     every row pitch must be 16-byte aligned (MOREC_E_ALIGN otherwise);
 *   - item ids are int32 on the device (the host narrows the int64 ids PyTorch provides).
 *
 * State the library keeps (everything else is stateless and re-entrant; first-call set-up of a kernel's launch attributes is
 * behind thread-safe function-local statics).  One process drives one GPU (the reference's process model, T/run.py:305-321):
 *   1. process-wide kernel-SELECTION knobs, morec_tuning_set() and the environment variables read once at the first GEMM
 *      launch: MOREC_GEMM8P (0 automatic | 1 never | 2 always the eight-phase kernel), MOREC_GEMM8P_TAIL_SPLIT (0 | 1),
 *      MOREC_GEMM8P_NGROUP (tile order), MOREC_GEMM8P_RESERVE_CUS, MOREC_GEMM_SKINNY, MOREC_CE8P, MOREC_GEMM8P_DEBUG (ablation bits), MOREC_GEMM_TILE /
 *      MOREC_GEMM_EPI (two-buffer kernel variants), MOREC_SWIN_BWD_WIDE.  Every selectable kernel computes the same function
 *      to the same accuracy; "gemm8p_tail_split" = 1 is the only knob that changes the fp32 SUMMATION ORDER (and with it the
 *      last-bit rounding pattern of bf16 outputs), which is why it is off unless asked for.  Set knobs before the first
 *      launch or between steps, not concurrently with launches;
 *   2. with "gemm8p_tail_split" = 1 only: one device scratch allocation per stream (the only memory the library ever allocates);
 *   3. morec_comm handles (below): created and destroyed by the caller;
 *   4. the dropout seed source (morec_dropout_seed_source): one device pointer, NULL by default;
 *   5. the ring of 256 events behind morec_stream_wait_stream (created on first use, never destroyed);
 *   6. deterministic mode only (morec_tuning_set("deterministic", 1) / MOREC_DETERMINISTIC=1; the reference sets torch's
 *      deterministic flags, T/run.py:313-314): one partial-sum scratch per (device, stream), grown on demand OUTSIDE graph capture
 *      (morec_det_scratch_reserve).  In this mode no kernel falls back to atomics: a call that cannot run its fixed-order form
 *      (scratch unavailable, no token order, row width outside the vector layout) returns an error instead.
 */
#ifndef MOREC_HIP_H
#define MOREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOREC_F32 0
#define MOREC_BF16 1
#define MOREC_F16 2 /* IEEE half operands / activations, fp32 accumulate: the reference's own GPU arithmetic (fp16 autocast, T/run.py:242) */

#define MOREC_OK 0
#define MOREC_E_ARG (-1)         /* null pointer / non-positive size */
#define MOREC_E_ALIGN (-2)       /* pointer or pitch not 16-byte aligned / size not a multiple of the vector width */
#define MOREC_E_UNSUPPORTED (-3) /* shape outside what the kernel was written for */
#define MOREC_E_DTYPE (-4)
#define MOREC_E_COMM (-5)        /* an RCCL call failed: morec_comm_last_error() */

#define MOREC_ACT_NONE 0
#define MOREC_ACT_GELU 1 /* exact erf GELU (HF BertIntermediate; T/model/encoders.py:59 nn.GELU) */
#define MOREC_ACT_RELU 2 /* T/model/modules.py:12 */
#define MOREC_DACT_MUL 3 /* morec_gemm_desc.dact only: C = acc * dact_in[m,n] (dact_in already holds act'(pre), see aux_deriv) */

const char* morec_strerror(int code);
int morec_version(void);
/* Process-wide kernel-selection knobs (measurement / A-B aid; see "State the library keeps" above).  Keys:
 *   "gemm8p"             0 = automatic, 1 = never use the 256 x 256 eight-phase GEMM, 2 = use it wherever it is eligible;
 *   "gemm2w"             0 = automatic, 1 = never use the 256 x 128 two-workgroups-per-CU GEMM (gemm2w.hip: the epilogue-heavy products --
 *                        GELU + act' outputs, x act' + column sums -- with >= 1024 tiles), 2 = every eligible 16-bit product; env MOREC_GEMM2W;
 *   "gemm_small"         0 = automatic, 1 = never use the 64 x 64 four-stage-ring GEMM (gemm_small.hip: the narrow long-K products of the SASRec
 *                        layers, N-tiles of 128 x 128 < 128 and K >= 1024), 2 = every eligible 16-bit product; outputs bit-identical; env MOREC_GEMM_SMALL;
 *   "gemm8p_tail_split"  1 = split the last, partly filled round of tiles along K between two workgroups (K >= 1536);
 *   "gemm8p_tail_bias"   share of K the first part takes in that split;
 *   "gemm8p_ngroup"      tile order: -1 = automatic column groups (default), 0 = row-major, n = column groups of n N-tiles;
 *   "gemm8p_reserve_cus" CUs (multiple of 8) left out of the persistent grid for a concurrent RCCL kernel;
 *   "gemm_skinny"        0 = automatic, 2 = gemm_skinny only (no gemm_skinny_wide / recompute), 1 = never use the streaming kernels for narrow outputs (gemm_skinny: N <= 288, K <= 384; gemm_skinny_wide:
 *                        N <= 768, K <= 192, GELU(product + bias); M >= 8192) -- morec_mlp_dact_recompute_supported() then answers 0;
 *   "ce8p"               scoring kernels: 0 = automatic, 1 = always the 128 x 128 kernels, 2 = the 256 x 256 eight-phase kernels
 *                        wherever the shape rules allow (bf16, D > 64, D % 8 == Nc % 8 == (B S) % 8 == 0);
 *   "deterministic"      1 = fixed summation order everywhere (state item 6), 0 = off; overrides MOREC_DETERMINISTIC;
 *   "gemm8p_debug", "gemm8p_stamps_lo/hi"  ablation bits / device address of a cycle-stamp buffer (diagnostics).
 * Returns MOREC_E_UNSUPPORTED for an unknown key.  No reference counterpart (the reference has no kernels). */
int morec_tuning_set(const char* key, int value);
/* Deterministic mode: make the partial-sum scratch of `stream` hold at least n_floats (call outside graph capture; under capture the
 * scratch cannot grow and a launch that needs more returns hipErrorOutOfMemory).  The largest user of the encoder step is the LayerNorm
 * backward: blocks x 3 x N floats.  No reference counterpart (torch.use_deterministic_algorithms, T/run.py:313-314, has no workspace API). */
int morec_det_scratch_reserve(size_t n_floats, void* stream);

/* ------------------------------------------------------------------------------------------
 * GEMM  C[M,N] (+)= alpha * A[M,K] . B[N,K]^T  (both operands K-contiguous, "NT")
 * Replaces every nn.Linear / torch.matmul on the path (T/model/modules.py:14,56-61;
 * HF modeling_bert.py BertSelfAttention/BertSelfOutput/BertIntermediate/BertOutput;
 * T/model/encoders.py:69; T/model/model.py:49) and, fed with transposed operands, their
 * autograd backward (dX = dY.W, dW = dY^T.X).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int M, N, K;
    int lda, ldb, ldc;
    int in_dtype;      /* dtype of A and B */
    int out_dtype;     /* dtype of C (and aux_out / dact_in) */
    int act;           /* MOREC_ACT_*: C = act(acc + bias); aux_out (optional) receives acc + bias */
    int dact;          /* MOREC_ACT_*: C = (acc) * act'(dact_in[m,n]) -- backward through an activation */
    int accumulate;    /* 0: C = v   1: C += v (non-atomic)   2: atomicAdd (fp32 C only; used with split_k) */
    int split_k;       /* >= 1: K is cut into split_k chunks over blockIdx.z (requires accumulate == 2 when > 1) */
    float alpha;
    int aux_deriv;     /* 1: aux_out receives act'(acc + bias) instead of acc + bias -- the backward GEMM then runs with
                        * dact = MOREC_DACT_MUL (one multiply per element instead of re-evaluating erf / exp for GELU') */
} morec_gemm_desc;

int morec_gemm_nt(const morec_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                  void* aux_out, const void* dact_in, void* stream);

/* Same, and colsum_out[n] += sum_m C[m, n]: with a dact epilogue C is the pre-activation gradient of the layer below, so
 * this is that layer's bias gradient (autograd of nn.Linear: db = dY.sum(0)) without a second pass over C.  The sums are
 * taken from the output tile while it is staged in LDS (the values as rounded to the output dtype), one partial row per
 * row block of the kernel's epilogue (32 ... 256 rows) into `workspace` (morec_gemm_colsum_workspace_bytes(M, N) bytes, fp32: sized for the
 * smallest block, 32 rows), folded by a second small launch.
 * colsum_out == NULL: plain morec_gemm_nt.  MOREC_E_UNSUPPORTED unless dact != NONE, accumulate == 0, split_k <= 1 and
 * the output rows are 16-byte addressable. */
size_t morec_gemm_colsum_workspace_bytes(int M, int N);
int morec_gemm_nt_colsum(const morec_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                         void* aux_out, const void* dact_in, float* colsum_out, float* workspace, void* stream);

/* Backward of  g = act(X . W1^T + b1) -> fc2  WITHOUT a stored act'(pre) tensor (GELU only):
 *   dU[M, N] = (dY[M, K] . W2t[N, K]^T) * act'(X[M, K] . W1[N, K]^T + b1[N]),   colsum_out[n] += sum_m dU[m, n]  (= d b1)
 * The pre-activation is recomputed from X (K = C wide) instead of being read back 4 C wide; the matching forward is morec_gemm_nt
 * with act = MOREC_ACT_GELU and aux_out == NULL.  All operands contiguous (pitches K, K, K, K, N), 16-bit (bf16 / f16), 16-byte aligned.
 * Written for the Swin stage-1 / stage-2 MLP (HF modeling_swin.py SwinIntermediate / SwinOutput: 288 < N <= 768, K <= 192 --
 * N <= 384 for K <= 96, N <= 512 for K <= 128 --, M >= 8192);
 * morec_mlp_dact_recompute_supported() says for which of those shapes dropping act' pays (K <= 128: the caller then skips the aux
 * output of the forward GEMM); the call itself also takes the K <= 192 class; anything else returns MOREC_E_UNSUPPORTED.  workspace: morec_mlp_dact_recompute_workspace_bytes(N) bytes of fp32 (per-workgroup
 * column sums, folded in a fixed order), needed when colsum_out != NULL.  autograd reference: torch.nn.functional.gelu backward +
 * nn.Linear backward (db = dY.sum(0)). */
int morec_mlp_dact_recompute_supported(int M, int N, int K, int dtype);
size_t morec_mlp_dact_recompute_workspace_bytes(int N);
int morec_mlp_dact_recompute(const void* dY, const void* W2t, const void* X, const void* W1, const float* b1, void* dU,
                             float* colsum_out, float* workspace, int M, int N, int K, int dtype, void* stream);

/* Weight-gradient GEMM without transposed copies (16-bit operands only; MOREC_E_UNSUPPORTED otherwise):
 * C[N, K] (+)= sum_m DY[m, n] * X[m, k], fp32 C.  split_m > 1 cuts the token range over blockIdx.z; without a workspace that
 * requires accumulate != 0 (fp32 atomicAdd into a caller-zeroed C), with one the partial tiles are folded by a second kernel and
 * accumulate == 0 overwrites C.  Autograd backward of nn.Linear: dW = dY^T X.
 * SINGLE WRITER: with accumulate != 0 the update of C is a plain read-modify-write (the slab fold; with one token chunk the GEMM's own
 * 16-byte lanes), not an atomic.  Every kernel that adds into the same C must therefore be ordered with this call -- the same stream, or
 * an event between the streams.  (The engines issue all weight-gradient GEMMs of a step on ONE stream, engine.WgradStream; a tied weight
 * accumulated from two streams at once would lose updates.) */
int morec_gemm_tn(const void* DY, const void* X, float* C, int M, int N, int K, int ldy, int ldx, int ldc, int dtype,
                  int split_m, int accumulate, float* workspace, void* stream);
/* workspace (optional, fp32): morec_gemm_tn_workspace_bytes(N, K, split_m).  With it the split-m partial tiles are written
 * as plain slabs and summed by a second kernel (deterministic, ~4x cheaper than the atomics); without it fp32 atomicAdd. */
size_t morec_gemm_tn_workspace_bytes(int N, int K, int split_m);

/* out[c, r] = in[r, c]; in is [R, C] with pitch ld_in, out is [C, R] with pitch ld_out.
 * in_dtype/out_dtype select a fused conversion (f32 -> bf16 weight shadows). */
int morec_transpose(const void* in, void* out, int R, int C, int ld_in, int ld_out, int in_dtype, int out_dtype,
                    void* stream);

/* Many transposes in ONE launch -- the W^T copies of every Linear weight that dX = dY . W needs once per optimisation step
 * (T/run.py:246: the weights change once per step; HF Linear backward reads W itself through cuBLAS' transposed operand).
 * items: DEVICE array of n_items entries sorted by tile0; entry i owns the 64 x 64 tiles [tile0_i, tile0_{i+1}) of the launch's
 * n_tiles blocks, tiles of one matrix row-major over ceil(rows / 64) x ceil(cols / 64).  dst[c][r] = src[r][c]; every matrix:
 * cols % 4 == 0, ld_src % 4 == 0, ld_dst % 4 == 0, ld_dst >= rows rounded up to 4, 16-byte aligned bases (same rules as the
 * vectorised path of morec_transpose; the CALLER checks them -- the table is not readable from the host side of this call). */
typedef struct {
    const void* src;
    void* dst;
    int rows, cols;       /* of src */
    int ld_src, ld_dst;   /* elements */
    int tile0;            /* first block of this matrix */
    int reserved;
} morec_transpose_item;
int morec_transpose_batch(const morec_transpose_item* items, int n_items, int n_tiles, int dtype, void* stream);
/* elementwise convert n elements */
int morec_cast(const void* in, void* out, size_t n, int in_dtype, int out_dtype, void* stream);
/* fp32 rows -> bf16 rows of three column blocks for an fp32-accurate product on the bf16 matrix cores ("bf16x3"):
 * out[r, 0:C] = hi = bf16(in[r, :]); block lo_slot (1 or 2) = lo = bf16(in - hi); the remaining block = hi again.  With
 * A3 = split(A, lo_slot 2) = [hi | hi | lo] and B3 = split(B, lo_slot 1) = [hi | lo | hi], morec_gemm_nt(A3, B3) over K' = 3 C is
 * A.B^T up to the lo.lo term (2^-16 relative), accumulated in fp32 -- the numerics of torch's fp32 nn.Linear (T/model/modules.py:14)
 * to ~1e-5 at three bf16 MFMA passes instead of the 1/16-rate exact-fp32 MFMA.  C % 4 == 0, ld_in >= C, ld_out >= 3 C. */
int morec_split_bf16x3(const float* in, void* out, int R, int C, int ld_in, int ld_out, int lo_slot, void* stream);
/* out = dy * act'(pre), elementwise (act = MOREC_ACT_GELU | MOREC_ACT_RELU); T/model/encoders.py:70 backward */
int morec_act_bwd(const void* dy, const void* pre, void* out, size_t n, int act, int dtype, void* stream);
/* out = scale * (x0 + x1 + x2), elementwise, fp32 arithmetic, one rounding to `dtype` (x1, x2 may be NULL; x2 needs x1): the mean over the text
 * attributes of a news item -- T/model/encoders.py:107-116, torch.mean(torch.stack(text_vectors, dim=1), dim=1) -- and, with one input and
 * scale = 1 / k, the gradient each attribute's encoder pass receives.  n % 4 == 0. */
int morec_scaled_sum(const void* x0, const void* x1, const void* x2, void* out, size_t n, float scale, int dtype, void* stream);
/* out[n] = sum_m in[m, n]  (bias gradients), atomically accumulated into fp32 out */
int morec_colsum(const void* in, float* out, int M, int N, int ld, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm with fused pre-add:  z = drop_in(x (+ bias[n])) (+ res[m,n]) (+ pos[m % pos_period, n]);
 * y = drop_out((z - mean) * rstd * gamma + beta).   (T/model/modules.py:14-17,61-63,93-94; HF BertSelfOutput /
 * BertOutput / BertEmbeddings).  z_out may be NULL (not needed) or alias x.
 * Dropout is counter-based: element e = m*N + n is kept iff the (e & 1)-th 16-bit half of hash(seed, e >> 1) >= floor(p * 2^16),
 * kept values scaled by 1 / (1 - floor(p * 2^16) / 2^16); p = 0 disables.  morec_dropout_keep_mask exports the same stream.
 * drop_in  = the Dropout the reference applies to the sub-layer output before the residual add;
 * drop_out = the Dropout applied to the LayerNorm output of the embedding stages.
 * rowscale (may be NULL): per-sample DropPath scale of the sub-layer branch (HF SwinDropPath, modeling_swin.py:42-60):
 * (x + bias) is multiplied by rowscale[m / rows_per_scale] before the residual add.
 * ------------------------------------------------------------------------------------------ */
int morec_layernorm_fwd(const void* x, const float* bias, const void* res, const float* pos, int pos_period,
                        const float* gamma, const float* beta, float eps, void* z_out, void* y, float* mean,
                        float* rstd, int M, int N, int dtype, float p_in, uint64_t seed_in, float p_out,
                        uint64_t seed_out, const float* rowscale, int rows_per_scale, void* stream);
/* dz = LN'(drop_out'(dy_a + dy_b); z) (+ dres); dzd = rowscale * drop_in'(dz) (required iff p_in > 0 or rowscale, else
 * NULL).  dres (may be NULL) is the gradient that reaches z directly along the residual stream of a pre-LN block
 * (HF SwinLayer, modeling_swin.py:536,560-566).  dgamma/dbeta (and, if given, dbias = column sums of dzd: the gradient of
 * the fused pre-add bias) are atomically accumulated (fp32, [N]).  dy_b, dgamma/dbeta and dbias may be NULL. */
int morec_layernorm_bwd(const void* dy_a, const void* dy_b, const void* z, const float* mean, const float* rstd,
                        const float* gamma, void* dz, void* dzd, float* dgamma, float* dbeta, float* dbias, int M, int N,
                        int dtype, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, const void* dres,
                        const float* rowscale, int rows_per_scale, void* stream);

/* LayerNorm of the reference's AUTOCAST data flow -- the `*_res32` compute modes.  Under `torch.cuda.amp.autocast()` (T/run.py:242) nn.Linear
 * takes and returns 16-bit tensors while LayerNorm runs and RETURNS fp32 (HF modeling_bert.py BertSelfOutput / BertOutput: LayerNorm(dropout(
 * dense(h)) + input_tensor); T/model/modules.py:14-17,61-63,93-94): the residual stream is fp32, only GEMM operands are rounded.
 *   forward:  z32 = drop_in(x16 (+ bias)) (+ res32) (+ pos[m % pos_period]);  y32 = drop_out(LN(z32) * gamma + beta);  y16 = round(y32).
 *             x16: 16-bit output of the sub-layer's GEMM (dtype16 = MOREC_BF16 | MOREC_F16); z32 (saved for the backward), y32 (the residual
 *             stream: the next LayerNorm's res32) and y16 (the next GEMM's operand) may each be NULL, not all of y32 / y16.
 *   backward: dz32 = LN'(drop_out'(dy16 + dy32); z32)  (dy16: gradient through the GEMM that read y16; dy32: gradient along the residual
 *             stream; either may be NULL);  dzd16 = round(drop_in'(dz32)): the 16-bit gradient of the sub-layer output, what its weight- /
 *             input-gradient GEMMs read.  dgamma / dbeta / dbias (= column sums of dzd16 as stored) are accumulated into, as in
 *             morec_layernorm_bwd.  dz32 or dzd16 may be NULL.
 * N % 8 == 0, N <= 4096, all tensors contiguous and 16-byte aligned.  Dropout streams: as morec_layernorm_fwd. */
int morec_layernorm_fwd_res32(const void* x16, const float* bias, const float* res32, const float* pos, int pos_period, const float* gamma,
                              const float* beta, float eps, float* z32, float* y32, void* y16, float* mean, float* rstd, int M, int N,
                              int dtype16, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, void* stream);
/* The same forward with the residual stream handed over in PRE-LayerNorm form: res_z32 is the z32 the PREVIOUS LayerNorm saved, and the stream
 * value is (res_z32 - res_mean[m]) * res_rstd[m] * res_gamma + res_beta, recomputed in registers -- the previous call then passes y32 = NULL and
 * never writes the stream (12 instead of 16 bytes per element and call; same numbers as reading y32 back up to fp32 contraction).  No pos, no
 * output dropout (the LayerNorms inside an encoder layer have neither: HF BertSelfOutput / BertOutput, T/model/modules.py:17,63). */
int morec_layernorm_fwd_res32_pre(const void* x16, const float* bias, const float* res_z32, const float* res_mean, const float* res_rstd,
                                  const float* res_gamma, const float* res_beta, const float* gamma, const float* beta, float eps, float* z32,
                                  void* y16, float* mean, float* rstd, int M, int N, int dtype16, float p_in, uint64_t seed_in, void* stream);
int morec_layernorm_bwd_res32(const void* dy16, const float* dy32, const float* z32, const float* mean, const float* rstd, const float* gamma,
                              float* dz32, void* dzd16, float* dgamma, float* dbeta, float* dbias, int M, int N, int dtype16, float p_in,
                              uint64_t seed_in, float p_out, uint64_t seed_out, void* stream);
/* dpos[m % period, n] += dz[m, n]  (position-embedding gradient, fp32 atomics) */
int morec_pos_grad(const void* dz, float* dpos, int M, int N, int period, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small-tile multi-head attention on packed projections qkv[M, 3*H] = [Q | K | V], M = n_seq*T,
 * H = n_heads*dh, T <= 256.  16-bit storage, head width % 32 == 0: the matrix-core kernels for T <= 32 and for 32 < T <= 64 (abstracts /
 * bodies of 50 tokens, T/parameters.py:43-44); fp32 storage, T <= 32, head width % 16 == 0: exact-fp32 MFMA (attention_f32mfma.hip: the
 * parity and fp32x3 modes); other head widths and fp32 up to T = 64: the exact-fp32 VALU tile kernels; 64 < T <= 256 (longer than any
 * launcher of the reference sets, accepted by its command line): the row-strip VALU kernels, every dtype; T > 256: MOREC_E_UNSUPPORTED.
 * scores = (q.k) * scale + (key masked ? mask_value : 0), causal option,
 * softmax, ctx = P.V.   SASRec: T/model/encoders.py:24-27 + T/model/modules.py:27-31 (causal,
 * mask_value -1e9, scale 1/sqrt(d_k)).  BERT: HF BertSelfAttention eager (mask_value finfo.min).
 * key_keep: float [n_seq, T], nonzero = attend.  One wavefront per (sequence, head).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int n_seq, T, n_heads, dh;
    int causal;
    float scale, mask_value;
    int dtype;
    float p_drop;      /* dropout on the attention probabilities (0 = off) */
    uint64_t seed;     /* element index = ((seq * n_heads + head) * TP + i) * TP + j, TP = 32 for T <= 32, 64 for T <= 64, 32 * ceil(T / 32) above */
    const int32_t* cu_seqlens;   /* NULL: every sequence owns T rows.  Otherwise int32[n_seq + 1] (device): sequence s owns rows
                                    cu_seqlens[s] .. cu_seqlens[s+1]-1 (<= T of them) -- the unpadded ("varlen") token layout in
                                    which [PAD] positions are not materialised at all; key_keep is then indexed by packed row */
    int total_rows;              /* with cu_seqlens: rows of qkv / ctx / dqkv (>= cu_seqlens[n_seq]), or 0.  The packed buffers may carry SPARE
                                    rows behind the last sequence (a token count padded up to a bucket size, so that one captured graph
                                    serves every batch of the bucket); */
    int spare_rows_max;          /* ... upper bound of their number: the launch appends ceil(spare_rows_max / 16) blocks that write zeros into
                                    the rows cu_seqlens[n_seq] .. total_rows-1 of ctx (forward) / dqkv (backward).  0 = no spare rows. */
} morec_attn_desc;

int morec_attn_fwd(const morec_attn_desc* d, const void* qkv, const float* key_keep, void* ctx, void* stream);
int morec_attn_bwd(const morec_attn_desc* d, const void* qkv, const float* key_keep, const void* dctx, void* dqkv,
                   void* stream);
/* The same backward plus the bias gradient of the fused q|k|v projection (HF BertSelfAttention query/key/value biases;
 * T/model/modules.py:30-32 has none): dbias[3 H] (fp32) += column sums of the `rows` dqkv rows (the bf16 MFMA path sums them in fp32
 * before the rows are rounded to bf16; otherwise the stored rows are summed).  ws: optional
 * scratch of n_seq * 3 H floats -- with it the bf16 path sums inside the attention kernel instead of re-reading dqkv. */
int morec_attn_bwd_dbias(const morec_attn_desc* d, const void* qkv, const float* key_keep, const void* dctx, void* dqkv,
                         int rows, float* dbias, float* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Swin vision tower (V/model/encoders.py:24-31 -> HF SwinForImageClassification, built at V/run.py:47-54;
 * transformers/models/swin/modeling_swin.py).  Token rows are kept in natural (image, y, x) order throughout:
 * window partition (:486-495), cyclic shift (:609-618) and window reverse (:498-505) are index arithmetic
 * inside the attention kernel, never copies.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int n_img, H, W;   /* token grid per image */
    int window;        /* 7 (49 tokens); H, W multiples of it */
    int shift;         /* 0 | window/2 (SW-MSA blocks): adds the -100 region mask of :584-607 */
    int heads, dh;     /* dh = 32 */
    float scale;       /* dh^-0.5 */
    int dtype;
} morec_swin_attn_desc;
/* ctx[row, h*dh..] = softmax(q.k*scale + bias_t[h][j][i] + shift mask) v over the row's window (SwinAttention :401-468).
 * qkv [rows, 3*heads*dh] = [q | k | v]; bias_t fp32 [heads][window^2 (key j)][window^2 (query i)]. */
int morec_swin_attn_fwd(const morec_swin_attn_desc* d, const void* qkv, const float* bias_t, void* ctx, void* stream);
/* dqkv from dctx (P recomputed; ctx = saved forward output); dbias_t (may be NULL) += dS summed over windows. */
int morec_swin_attn_bwd(const morec_swin_attn_desc* d, const void* qkv, const float* bias_t, const void* ctx,
                        const void* dctx, void* dqkv, float* dbias_t, void* stream);
/* The same backward plus the bias gradient of the fused q|k|v projection (modeling_swin.py:407-409): dbqkv[3 heads dh] (fp32) +=
 * column sums of dqkv.  ws: optional scratch (n_windows x 3 heads dh floats is always enough) -- with it the bf16 path sums
 * inside the attention kernel (fp32 sums of the rows BEFORE their bf16 rounding) instead of re-reading dqkv. */
int morec_swin_attn_bwd_dbias(const morec_swin_attn_desc* d, const void* qkv, const float* bias_t, const void* ctx,
                              const void* dctx, void* dqkv, float* dbias_t, float* dbqkv, float* ws, size_t ws_bytes, void* stream);
/* bias_t[h][j][i] = table[rel_index(i, j)][h] (SwinRelativePositionBias :329-370) and its transpose-scatter gradient */
int morec_swin_bias_expand(const float* table, float* bias_t, int window, int heads, void* stream);
int morec_swin_bias_reduce(const float* dbias_t, float* dtable, int window, int heads, void* stream);
/* im2col of the patch-embedding conv (SwinPatchEmbeddings :247-286): out[(n,py,px), (c,i,j)] = pixels[n,c,py*p+i,px*p+j];
 * pixels fp32 NCHW (what V/run.py:201-204 puts on the device), out dtype rows of pitch ld_out. */
int morec_swin_patchify(const float* pixels, void* out, int n_img, int channels, int R, int patch, int ld_out, int dtype,
                        void* stream);
/* the same rows from decoded uint8 HWC images [n, R, R, channels]: ToTensor + Normalize(mean, std) of
 * V/data_utils/dataset.py:69-73 applied in flight (fp32, reference operation order: (x / 255 - mean) / std). */
int morec_swin_patchify_u8(const uint8_t* pixels_hwc, void* out, int n_img, int channels, int R, int patch, int ld_out,
                           float mean, float std, int dtype, void* stream);
/* SwinPatchMerging gather (:309-320): [n,H,W,C] -> [n,H/2,W/2,4C]; reverse != 0 is the inverse copy (backward). */
int morec_swin_merge(const void* in, void* out, int n_img, int H, int W, int C, int reverse, int dtype, void* stream);
/* mean over each image's tokens (AdaptiveAvgPool1d(1), :876-879) and its backward */
int morec_swin_pool_fwd(const void* x, void* out, int n_img, int tokens, int C, int dtype, void* stream);
int morec_swin_pool_bwd(const void* dout, void* dx, int n_img, int tokens, int C, int dtype, void* stream);
/* out = res + rowscale[m / rows_per_scale] * (a + bias)  (bias, rowscale may be NULL) */
int morec_bias_residual(const void* a, const float* bias, const void* res, const float* rowscale, int rows_per_scale,
                        void* out, int M, int N, int dtype, void* stream);
/* DropPath per-sample scales: out[i] = hash(seed, i) kept ? 1/(1-p) : 0  (SwinDropPath :51-57) */
int morec_droppath_scale(float* out, int n, float p, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------------
 * BCE variant, one sampled negative per position (bce_text/main-end2end/model/model.py:30-51).
 * E [B, S+1, 2, D] item vectors (pos at [:, :, 0], neg at [:, :, 1]), P [B, S, D] user states, row_valid u8 [B*S].
 * scores fp32 [2][B*S] (pos | neg) are saved for the backward; loss_sum[0] += sum_valid softplus(-pos) + softplus(neg).
 * Backward: gscale[0] = dloss / n_valid on the device; writes dP [B, S, D] and EVERY row of dE [B, S+1, 2, D] (no atomics).
 * ------------------------------------------------------------------------------------------ */
int morec_bce_fwd(const void* P, const void* E, const uint8_t* row_valid, float* scores, float* loss_sum, int B, int S, int D,
                  int dtype, void* stream);
int morec_bce_bwd(const void* P, const void* E, const uint8_t* row_valid, const float* scores, const float* gscale, void* dP,
                  void* dE, int B, int S, int D, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embeddings
 * ------------------------------------------------------------------------------------------ */
/* BERT embeddings: z = word[ids[m]] + pos[m % T] + type0;  y = LN(z)  (HF BertEmbeddings). */
int morec_bert_embed_fwd(const int32_t* ids, const float* word, const float* pos, const float* type0,
                         const float* gamma, const float* beta, float eps, void* z_out, void* y, float* mean,
                         float* rstd, int M, int T, int H, int dtype, float p_out, uint64_t seed_out, void* stream);
/* scatter dz into dword[ids[m]] (skipping pad_id: nn.Embedding padding_idx), dpos[m % T], dtype0.
 * order (may be NULL): int32[M] permutation of the rows in which EQUAL token ids are adjacent (argsort of ids); a run of equal
 * ids is summed in registers, and a run that lies inside one wavefront's 32 rows is added to its table row by a plain
 * read-modify-write (it is that row's only contribution), the others atomically.  An order that is not grouped by id is a
 * caller error (lost updates); without order every row is added atomically. */
int morec_bert_embed_bwd(const int32_t* ids, const void* dz, float* dword, float* dpos, float* dtype0, int pad_id,
                         int M, int T, int H, int dtype, const int32_t* order, void* stream);
/* out[r, :] = table[idx[r], :]  (fp32 table -> dtype out).  T/model/model.py:37 nn.Embedding lookup. */
int morec_gather_rows(const float* table, const int32_t* idx, void* out, int R, int D, int dtype, void* stream);
/* dtable[idx[r], :] += d[r, :] unless idx[r] == pad_id (padding_idx=0, T/model/model.py:27) */
int morec_scatter_add_rows(const void* d, const int32_t* idx, float* dtable, int R, int D, int pad_id, int dtype,
                           void* stream);
/* strided row copy: out[r, :] = in[r*stride_rows, :]  (hidden[:, 0], T/model/encoders.py:69) and its
 * backward (scatter into a zero-filled [R*stride_rows, D]) */
/* out[out_idx ? out_idx[r] : r, :] = in[in_idx ? in_idx[r] : r, :] for r < R (row gather / scatter; indices int32, device);
 * in_idx[r] < 0 writes a zero row (the [PAD] rows when a packed token layout is spread back over the padded one) */
int morec_indexed_rows_copy(const void* in, void* out, const int32_t* in_idx, const int32_t* out_idx, int R, int D, int dtype,
                            void* stream);
int morec_strided_rows_copy(const void* in, void* out, int R, int D, int in_row_stride, int out_row_stride,
                            int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * In-batch debiased sampled-softmax cross-entropy (T/model/model.py:32-33,45-67), fused:
 * logits never reach HBM.  P [Nr, D] user states (Nr = B*S rows), E [Nc, D] item vectors of the
 * column pool (local: Nc = B*(S+1); pooled over ranks: SURVEY.md §8e), both `dtype`.
 *   row_ids   int32 [B*(S+1)]  ids of this rank's users (reject sets)
 *   col_ids   int32 [Nc]       ids of the pool columns
 *   col_logpop float [Nc]      log(pop_prob[col_ids])
 *   col_valid uint8 [Nc]       slot is not padding (cat(log_mask, 1) != 0)
 *   row_valid uint8 [Nr]       log_mask != 0
 *   col_offset                 column of this rank's slot 0 in the pool
 * fwd: per-(row, 64-column slice) partial (max, sumexp) + positive logit -> row_lse[Nr], row_loss[Nr]
 *      and loss_sum (fp32 scalar, += the sum of the valid rows' losses, added in a fixed order: bit-reproducible; caller zeroes it).
 * bwd: dP[Nr, D] and dE[Nc, D] (dtype, overwritten):
 *      dlogit = gscale * (*gscale_dev if given) * (softmax - onehot) on unmasked cells of valid rows,
 *      0 elsewhere; dP = dlogit . E, dE = dlogit^T . P.
 * workspace (both): morec_inbatch_ce_workspace_bytes().
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int B, S, D;
    int Nc;
    int col_offset;
    int dtype;
    int dE_fp32;   /* backward only: dE is written as fp32 [Nc, D] whatever the compute dtype (the pooled-negative step reduce-scatters it
                    * over ranks in fp32); bf16 compute with Nc % 8 == 0 only, MOREC_E_UNSUPPORTED otherwise */
    int ws_from_fwd; /* backward only: 1 = `workspace` is the buffer the matching morec_inbatch_ce_fwd call (same descriptor fields, same
                      * P / E / ids / log-pop / validity) was given and nothing has written to it since: the backward then reuses the
                      * (user, column) flag table and the positive logits the forward left there instead of rebuilding them (the
                      * 256 x 256 kernels; ignored by the 128 x 128 ones; the "ce8p" knob must not change between the two calls).
                      * 0 = the workspace is plain scratch. */
} morec_ce_desc;

size_t morec_inbatch_ce_workspace_bytes(const morec_ce_desc* d);
int morec_inbatch_ce_fwd(const morec_ce_desc* d, const void* P, const void* E, const int32_t* row_ids,
                         const int32_t* col_ids, const float* col_logpop, const uint8_t* col_valid,
                         const uint8_t* row_valid, float* row_lse, float* row_loss, float* loss_sum,
                         void* workspace, void* stream);
int morec_inbatch_ce_bwd(const morec_ce_desc* d, const void* P, const void* E, const int32_t* row_ids,
                         const int32_t* col_ids, const float* col_logpop, const uint8_t* col_valid,
                         const uint8_t* row_valid, const float* row_lse, const float* gscale_dev, float gscale,
                         void* dP, void* dE, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused AdamW over a flat fp32 parameter arena (torch.optim.AdamW semantics, T/run.py:159-162,246).
 * One launch per hyper-parameter group (contiguous range).  Optionally refreshes the bf16 shadow.
 * step = 1-based step count after increment.
 * ------------------------------------------------------------------------------------------ */
int morec_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, size_t n,
                float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                void* stream);

/* ------------------------------------------------------------------------------------------
 * Device-resident step state: what the reference keeps in `torch.cuda.amp.GradScaler()` and in AdamW's per-parameter `step`
 * (T/run.py:210: scaler = GradScaler(); :243-247: scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()), held in
 * one 64-byte DEVICE block so that a step needs no host round trip and no per-step host scalar among its kernel arguments
 * (a captured graph of the step can be replayed).  Protocol of one step in the fp16 mode:
 *   forward; backward with the loss gradient multiplied by sp->loss_scale (morec_inbatch_ce_bwd: gscale_dev = loss_scale / n_valid);
 *   gradient reduction over ranks; morec_grad_check_finite over every gradient arena; morec_step_decide; morec_adamw_sp per group.
 * A step with a non-finite gradient leaves parameters, moments, shadows and the step count untouched and halves the scale
 * (GradScaler defaults: init 65536, growth 2 every 2000 clean steps, backoff 0.5 -- passed in by the caller).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t step;           /* optimizer steps APPLIED so far (a skipped step does not count, as torch never calls optimizer.step() for it) */
    int32_t found_inf;      /* != 0: a non-finite gradient has been seen since the last morec_step_decide */
    int32_t growth_tracker; /* consecutive clean steps since the scale last changed */
    int32_t skipped;        /* steps skipped so far (diagnostics) */
    float loss_scale;       /* S of the NEXT backward pass */
    float inv_scale;        /* 1 / S of the step being applied: what morec_adamw_sp multiplies the gradients with */
    float bc1, bc2;         /* 1 - beta1^step, 1 - beta2^step of the step being applied */
    int32_t apply;          /* decision of the last morec_step_decide: 1 = update, 0 = skip */
    int32_t pad0;
    uint64_t drop_seed;       /* splitmix64 state, advanced by every morec_step_decide */
    uint64_t drop_seed_mixed; /* its output: the word morec_dropout_seed_source(&sp->drop_seed_mixed) feeds to the dropout kernels */
    int32_t reserved[2];
} morec_step_params;
/* Process-wide dropout seed source (state item 4 of the list at the top of this file): dev_u64 = a DEVICE uint64 (8-byte aligned) that every
 * dropout / DropPath kernel XORs into the two halves of its seed argument at kernel entry, NULL = none (the default).  The mask of a launch
 * is then a function of (seed argument, *dev_u64): a captured graph of the step -- whose seed arguments are frozen at capture -- draws
 * fresh masks at every replay when the word changes between replays (morec_step_decide advances sp->drop_seed_mixed).  The word must not
 * change between the forward and the backward launch that regenerate the same mask.  Stands where torch's global CUDA RNG state stands in
 * the reference (nn.Dropout under T/run.py:307-314 `setup_seed`). */
int morec_dropout_seed_source(const void* dev_u64);
/* Stream `waiting` runs nothing issued after this call before the work issued to stream `signal` so far has finished (event record + stream
 * wait on an internal ring of 256 events; capturable; one device per process).  torch.cuda.Stream.wait_stream in one foreign call: orders the
 * weight-gradient stream behind the backward chain per dW launch.  No reference counterpart (autograd runs its backward on one stream). */
int morec_stream_wait_stream(void* waiting, void* signal);
/* *sp = {step, loss_scale = init_scale, everything else clear} (init_scale = 1: no scaling, bf16 / fp32 modes) */
int morec_step_params_init(morec_step_params* sp, float init_scale, int step, void* stream);
/* sp->found_inf |= any element of grad[0 .. n) is inf or NaN   (GradScaler.unscale_'s found_inf; the division by S is left to AdamW) */
int morec_grad_check_finite(const float* grad, size_t n, morec_step_params* sp, void* stream);
/* GradScaler.step + update on the device: apply = !found_inf; on apply: ++step, bias corrections (formed in double), growth
 * bookkeeping; on skip: ++skipped, scale *= backoff_factor; found_inf is cleared.  dynamic == 0 keeps the scale fixed. */
int morec_step_decide(morec_step_params* sp, float beta1, float beta2, float growth_factor, float backoff_factor,
                      int growth_interval, int dynamic, void* stream);
/* morec_adamw with step count / bias corrections / gradient scale (1 / S) read from *sp; does nothing when sp->apply == 0.
 * shadow (may be NULL): 16-bit copy of the updated parameters in shadow_dtype (MOREC_BF16 | MOREC_F16). */
int morec_adamw_sp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow, int shadow_dtype, size_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, const morec_step_params* sp, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation (T/data_utils/metrics.py:96-102,49-57): rank of the target among all items after
 * masking the history -- count-greater instead of a full argsort.
 * scores = prec[U, D] . item_emb[item_num+1, D]^T are formed tile by tile; rank[u] = 1 + #{items i>=1,
 * i not in history(u), score[u,i] > score[u,target[u]]}.
 * hist: int32 [U, Hmax] padded with -1.
 * ------------------------------------------------------------------------------------------ */
int morec_eval_rank(const float* prec, const float* item_emb, const int32_t* hist, int Hmax, const int32_t* target,
                    int32_t* rank, float* tscore_ws /* float[U] scratch */, int U, int n_items_plus1, int D,
                    void* stream);

/* test hook: out[i] = 1 iff the dropout hash keeps element index i for (p, seed) */
int morec_dropout_keep_mask(uint8_t* out, size_t n, float p, uint64_t seed, void* stream);

/* diagnostics: dumps MFMA fragment layouts and ds_read_b64_tr_b16 semantics into out (int32[4096]) */
int morec_probe(int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Vision input pipeline, device half: n decoded uint8 HWC images of arbitrary sizes, packed back to back
 * in `src`, -> uint8 [n, R, R, 3], bit for bit the reference's host-side tv.transforms.Resize((R, R)) on the PIL image
 * (V/data_utils/dataset.py:68-73,91-98 = Pillow's two-pass BILINEAR resampler).  meta: int64 [n][5] = byte offset in src, H, W,
 * offset (in int32 words) of the horizontal and of the vertical tap table in `tables`; a table = ksize, then R rows of
 * (first input index, tap count, taps[ksize]) in Pillow's 2^22 fixed point (host: data_utils/images.py::resize_table).
 * ToTensor + Normalize(0.5, 0.5) are fused into morec_swin_patchify_u8.
 * ------------------------------------------------------------------------------------------ */
int morec_image_resize_u8(const uint8_t* src, const int64_t* meta, const int32_t* tables, uint8_t* out, int n, int R, void* stream);

/* ------------------------------------------------------------------------------------------
 * Collectives of the data-parallel step on the CALLER's stream (RCCL over xGMI, resolved with dlopen from the
 * librccl.so.1 already mapped in the process -- PyTorch's -- and never linked).  They replace, stream-ordered with the kernels
 * around them, what the reference gets from DistributedDataParallel's NCCL hooks (T/run.py:148,321) plus the exchange the
 * pooled-negative step adds:  all-gather of the encoded item vectors E [Nc, D] and of the packed (ids | log-pop | validity |
 * n_valid) record, reduce-scatter(SUM) of dE_pool [world Nc, D] in fp32, all-reduce(SUM) of a gradient bucket in fp32.
 * h_id128: HOST buffer of 128 bytes (ncclUniqueId): rank 0 fills it with morec_comm_unique_id and hands it to the other ranks
 * out of band (torch.distributed store / broadcast, MPI, a file); morec_comm_create is collective over the `world` ranks and
 * needs this process's device to be current.  One handle per (process, device); calls on one handle must not overlap in time
 * on different streams.  Errors: MOREC_E_UNSUPPORTED = no RCCL library found, MOREC_E_COMM = RCCL reported a failure.
 * ------------------------------------------------------------------------------------------ */
typedef struct morec_comm morec_comm;
int morec_comm_available(void);                       /* 1 when librccl.so.1 and every entry point needed were found */
int morec_comm_unique_id(void* h_id128);
int morec_comm_create(morec_comm** out, const void* h_id128, int rank, int world);
int morec_comm_destroy(morec_comm* comm);
const char* morec_comm_last_error(const morec_comm* comm);
/* recv[r * bytes_per_rank ...] = rank r's send[0 .. bytes_per_rank)   (ncclAllGather, byte-typed) */
int morec_comm_all_gather(morec_comm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
/* recv[i] = sum_r send_r[rank * count_per_rank + i]                   (ncclReduceScatter, fp32 SUM) */
int morec_comm_reduce_scatter_f32(morec_comm* comm, const float* send, float* recv, size_t count_per_rank, void* stream);
/* buf[i] = sum_r buf_r[i], in place                                   (ncclAllReduce, fp32 SUM) */
int morec_comm_all_reduce_f32(morec_comm* comm, float* buf, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOREC_HIP_H */
