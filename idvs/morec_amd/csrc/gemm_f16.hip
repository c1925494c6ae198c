// gemm_f16.hip -- the two-buffer NT GEMM kernels of gemm_nt_generic.hpp for fp16 operands (MOREC_F16: v_mfma_f32_16x16x32_f16):
// small problems, split-K accumulation and narrow N of the fp16 mode; large ones run on the eight-phase kernel (gemm8p.hip).
#include "gemm_nt_generic.hpp"

int gemm_nt_f16_launch(const morec_gemm_desc* d, GemmArgs& a, hipStream_t s) {
    if (d->out_dtype == MOREC_F16) return launch_gemm<f16, f16>(d, a, s);
    if (d->out_dtype == MOREC_F32) return launch_gemm<f16, float>(d, a, s);
    return MOREC_E_DTYPE;
}
