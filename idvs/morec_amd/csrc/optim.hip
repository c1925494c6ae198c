// optim.hip -- fused AdamW over a flat fp32 arena (one launch per hyper-parameter group) with the bf16
// shadow refresh folded in.  torch.optim.AdamW semantics (T/run.py:159-162,246):
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// HBM-bound: 16 B read + 12 B written per parameter (+ 2 B shadow).
#include "common.hpp"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    unsigned short* __restrict__ shadow, size_t n4, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, float gscale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float pa[4] = {pp.x, pp.y, pp.z, pp.w};
        const float ga[4] = {gg.x * gscale, gg.y * gscale, gg.z * gscale, gg.w * gscale};
        float ma[4] = {mm.x, mm.y, mm.z, mm.w};
        float va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pa[k] *= (1.0f - lr * wd);
            ma[k] = b1 * ma[k] + (1.0f - b1) * ga[k];
            va[k] = b2 * va[k] + (1.0f - b2) * ga[k] * ga[k];
            const float denom = sqrtf(va[k]) * inv_sqrt_bc2 + eps;
            pa[k] -= (lr * inv_bc1) * (ma[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
        if (shadow) {
            uint2 s;
            s.x = pack_bf16x2(pa[0], pa[1]);
            s.y = pack_bf16x2(pa[2], pa[3]);
            reinterpret_cast<uint2*>(shadow)[i] = s;
        }
    }
}

extern "C" int morec_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                           size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                           float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return MOREC_E_ARG;
    if (n == 0) return MOREC_OK;
    if (n % 4 || !aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq) ||
        (shadow_bf16 && (reinterpret_cast<uintptr_t>(shadow_bf16) & 7u)))
        return MOREC_E_ALIGN;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param,
                       grad, exp_avg, exp_avg_sq, reinterpret_cast<unsigned short*>(shadow_bf16), n4, lr, beta1, beta2,
                       eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
