// gemm_small.hip -- NT GEMM for the LATENCY-CLASS 16-bit products: the SASRec user encoder's Linear layers over B S = 2 560 rows
// (T/model/modules.py:41-44,8-9: w_Q / w_K / w_V / fc, w_1 / w_2 at d_model = 512) and the products of the same size around them.
// On 128 x 128 tiles such a product is 80 - 320 workgroups whose two-buffer main loop exposes one global-load latency per 64-wide K
// stage: 10 - 33 us per launch at 60 - 170 TFLOP/s (profiles/r06_id_tower_step_before.txt).  Here: 64 x 64 tiles (four waves of 32 x 32),
// a four-stage LDS-DMA ring with counted waits (gemm_core.hpp, gemm_mainloop_ring), 64 KiB of LDS so that two workgroups share a CU.
// The epilogues are the generic kernel's (gemm_nt_generic.hpp): bias, ReLU / GELU (+ act'), x act' + column sums.  Outputs are
// bit-identical to the 128 x 128 kernel's: the same MFMA, the same K order per output element.
#include "gemm_nt_generic.hpp"

// tuning key "gemm_small" / MOREC_GEMM_SMALL: 0 = automatic, 1 = never, 2 = every eligible product
int g_mode_small = -1;

int gemm_small_try_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    if (g_mode_small < 0) {
        const char* e = getenv("MOREC_GEMM_SMALL");
        g_mode_small = e ? atoi(e) : 0;
    }
    if (g_mode_small == 1) return G8_NOT_TAKEN;
    if (!is_h16(d->in_dtype) || d->out_dtype != d->in_dtype || a.accumulate != 0 || !a.vec_store || d->split_k > 1) return G8_NOT_TAKEN;
    if (d->K < 64 || d->M < 1) return G8_NOT_TAKEN;
    if (g_mode_small != 2) {
        // automatic: where it measured faster than the 128 x 128 kernel in isolation (scripts/small_gemm_check.py, profiles/r06_small_gemm.txt):
        // narrow outputs with a long contraction -- fewer than 128 tiles of 128 x 128 and K >= 1024 (f2 / dx1 at K = 2048: 19.7 -> 14.0 us,
        // dx0 at K = 1536: 15.6 -> 11.3, the one-GPU scoring backward's dE at K = 2560: 30 -> 21).  With N >= 1536 the wider tile's
        // half LDS-fill traffic per flop wins (qkv 9.0 vs 11.0 us, f1 + ReLU 14.7 vs 17.6); at K = 512 the two are level.
        const long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
        if (t128 > 128 || d->K < 1024 || d->N < 256 || a.aux_out || a.colsum) return G8_NOT_TAKEN;
    }
    a.wave_epilogue = 1;      // the 64-wide tile row: per-wave LDS slices, no workgroup barriers in the epilogue
    if (d->in_dtype == MOREC_F16) return launch_gemm_cfg<GemmTileCfg<f16, 2, 2, 2, 2, 2, 4>, f16, f16>(d, a, s);
    return launch_gemm_cfg<GemmTileCfg<bf16, 2, 2, 2, 2, 2, 4>, bf16, bf16>(d, a, s);
}
