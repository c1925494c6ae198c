// optim.hip -- fused AdamW over a flat fp32 arena (one launch per hyper-parameter group) with the bf16
// shadow refresh folded in.  torch.optim.AdamW semantics (T/run.py:159-162,246):
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// HBM-bound: 16 B read + 12 B written per parameter (+ 2 B shadow).
//
// morec_step_params / morec_adamw_sp: the same update driven by a DEVICE-resident block (step count, bias corrections, loss scale,
// overflow flag) -- the GradScaler protocol of the reference's fp16 step (T/run.py:210,243-247) without a host round trip, and the
// form a captured graph can replay (no per-step host scalars among the kernel arguments).
#include "common.hpp"

// SH16: storage type of the shadow (bf16 | f16).  sp != nullptr: step state from the device block (skip when sp->apply == 0).
template <typename SH16>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    unsigned short* __restrict__ shadow, size_t n4, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, float gscale, const morec_step_params* __restrict__ sp) {
    if (sp) {
        if (!sp->apply) return;           // a non-finite gradient somewhere in this step: parameters, moments and shadow stay as they are
        inv_bc1 = 1.0f / sp->bc1;
        inv_sqrt_bc2 = rsqrtf(sp->bc2);
        gscale = sp->inv_scale;
    }
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
            s.x = h16<SH16>::pack2(pa[0], pa[1]);
            s.y = h16<SH16>::pack2(pa[2], pa[3]);
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
    hipLaunchKernelGGL(adamw_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param,
                       grad, exp_avg, exp_avg_sq, reinterpret_cast<unsigned short*>(shadow_bf16), n4, lr, beta1, beta2,
                       eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, (const morec_step_params*)nullptr);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

extern "C" int morec_adamw_sp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow, int shadow_dtype,
                              size_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                              const morec_step_params* sp, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !sp) return MOREC_E_ARG;
    if (n == 0) return MOREC_OK;
    if (shadow && !is_h16(shadow_dtype)) return MOREC_E_DTYPE;
    if (n % 4 || !aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq) ||
        (shadow && (reinterpret_cast<uintptr_t>(shadow) & 7u)))
        return MOREC_E_ALIGN;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (shadow && shadow_dtype == MOREC_F16)
        hipLaunchKernelGGL(adamw_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq,
                           reinterpret_cast<unsigned short*>(shadow), n4, lr, beta1, beta2, eps, weight_decay, 1.f, 1.f, 1.f, sp);
    else
        hipLaunchKernelGGL(adamw_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq,
                           reinterpret_cast<unsigned short*>(shadow), n4, lr, beta1, beta2, eps, weight_decay, 1.f, 1.f, 1.f, sp);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// ---- the device-resident step block ------------------------------------------------------------------------------------------------
__global__ void step_params_init_kernel(morec_step_params* sp, float init_scale, int step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        morec_step_params z = {};
        z.step = step;
        z.loss_scale = init_scale;
        z.inv_scale = 1.0f / init_scale;
        z.bc1 = 1.f; z.bc2 = 1.f;
        z.apply = 1;
        z.drop_seed = 0x9E3779B97F4A7C15ull * (uint64_t)(step + 1);
        uint64_t x = z.drop_seed;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        z.drop_seed_mixed = x ^ (x >> 31);
        *sp = z;
    }
}
extern "C" int morec_step_params_init(morec_step_params* sp, float init_scale, int step, void* stream) {
    if (!sp || !(init_scale > 0.f) || step < 0) return MOREC_E_ARG;
    hipLaunchKernelGGL(step_params_init_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sp, init_scale, step);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// found_inf |= any(!isfinite(grad)): one pass over the fp32 gradients (4 B / parameter), one atomic per wavefront that saw one
__global__ __launch_bounds__(256) void grad_check_kernel(const float* __restrict__ g, size_t n4, morec_step_params* __restrict__ sp) {
    bool bad = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        // !(|x| <= FLT_MAX) is true for +-inf and for NaN
        bad |= !(fabsf(v.x) <= 3.4028234663852886e38f) | !(fabsf(v.y) <= 3.4028234663852886e38f) | !(fabsf(v.z) <= 3.4028234663852886e38f) |
               !(fabsf(v.w) <= 3.4028234663852886e38f);
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&sp->found_inf, 1);
}
extern "C" int morec_grad_check_finite(const float* grad, size_t n, morec_step_params* sp, void* stream) {
    if (!grad || !sp) return MOREC_E_ARG;
    if (n == 0) return MOREC_OK;
    if (n % 4 || !aligned16(grad)) return MOREC_E_ALIGN;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(grad_check_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), grad, n4, sp);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}

// GradScaler.step + update (torch/amp/grad_scaler.py: step() skips optimizer.step() when an inf / NaN was found; update() multiplies the
// scale by backoff_factor then, and by growth_factor after growth_interval consecutive clean steps) as ONE single-thread kernel between
// the checks and the AdamW launches of a step.  The bias corrections are formed in double like torch's `1 - beta ** step`.
__global__ void step_decide_kernel(morec_step_params* sp, float beta1, float beta2, float growth, float backoff, int interval, int dynamic) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    morec_step_params z = *sp;
    z.inv_scale = 1.0f / z.loss_scale;        // the scale this step's gradients carry
    if (z.found_inf) {
        z.apply = 0;
        z.skipped += 1;
        if (dynamic) { z.loss_scale *= backoff; z.growth_tracker = 0; }
    } else {
        z.apply = 1;
        z.step += 1;
        z.bc1 = (float)(1.0 - pow((double)beta1, (double)z.step));
        z.bc2 = (float)(1.0 - pow((double)beta2, (double)z.step));
        if (dynamic) {
            z.growth_tracker += 1;
            if (z.growth_tracker >= interval) { z.loss_scale *= growth; z.growth_tracker = 0; }
        }
    }
    z.found_inf = 0;
    {   // the dropout seed word of the NEXT forward / backward pair (splitmix64 step): every decide call draws a new one
        uint64_t x = (z.drop_seed += 0x9E3779B97F4A7C15ull);
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        z.drop_seed_mixed = x ^ (x >> 31);
    }
    *sp = z;
}
extern "C" int morec_step_decide(morec_step_params* sp, float beta1, float beta2, float growth_factor, float backoff_factor,
                                 int growth_interval, int dynamic, void* stream) {
    if (!sp || growth_interval < 1 || !(growth_factor >= 1.f) || !(backoff_factor > 0.f && backoff_factor <= 1.f)) return MOREC_E_ARG;
    hipLaunchKernelGGL(step_decide_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sp, beta1, beta2, growth_factor,
                       backoff_factor, growth_interval, dynamic);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
