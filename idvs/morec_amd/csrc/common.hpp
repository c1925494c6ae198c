// common.hpp -- shared device helpers for the gfx950 (MI355X) kernels of libmorec_hip.so.
// wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>
#include "morec_hip.h"

#define MOREC_WAVE 64

struct bf16 {
    unsigned short v;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(bf16 x) { return __uint_as_float(((uint32_t)x.v) << 16); }
__device__ __forceinline__ float bfbits2f(uint32_t lo16) { return __uint_as_float(lo16 << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
// fp32 -> bf16, round-to-nearest-even: the native casts lower to v_cvt_pk_bf16_f32 (ONE instruction per pair; a
// hand-rolled integer RNE cost ~8 VALU per element and made every bf16 store path -- GEMM epilogue, LayerNorm,
// attention -- VALU-bound: the 256 x 256 GEMM epilogue alone took as long as its K = 768 main loop)
__device__ __forceinline__ unsigned short f2bf_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16 f2bf(float f) { return bf16{f2bf_bits(f)}; }

// ---- fp16 storage (MOREC_F16): IEEE half, 11-bit significand -- the reference's own GPU arithmetic (fp16 autocast + GradScaler,
// T/run.py:210,242-247).  Same data movement as bf16 (2-byte elements); only the conversions and the MFMA opcode differ.
struct f16 {
    unsigned short v;
};
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ float hbits2f(uint32_t lo16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)lo16); }
__device__ __forceinline__ float h2f(f16 x) { return hbits2f(x.v); }
// round-to-nearest-even (v_cvt_f16_f32); values past 65504 become +-inf, which is what the loss scaler's overflow check looks for
__device__ __forceinline__ unsigned short f2h_bits(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ f16 f2h(float f) { return f16{f2h_bits(f)}; }

// ---- typed 4-element vector IO (16 B for f32, 8 B for bf16) --------------------------------------
template <typename T>
struct io;
template <>
struct io<float> {
    static constexpr int dtype = MOREC_F32;
    __device__ __forceinline__ static void load4(const float* p, float (&o)[4]) {
        float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    __device__ __forceinline__ static void store4(float* p, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __device__ __forceinline__ static float load1(const float* p) { return *p; }
    __device__ __forceinline__ static void store1(float* p, float v) { *p = v; }
    // what a value becomes when written to memory as T and read back
    __device__ __forceinline__ static float round(float v) { return v; }
};
template <>
struct io<bf16> {
    static constexpr int dtype = MOREC_BF16;
    __device__ __forceinline__ static void load4(const bf16* p, float (&o)[4]) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = bfbits2f(v.x & 0xffffu); o[1] = bfbits2f(v.x >> 16);
        o[2] = bfbits2f(v.y & 0xffffu); o[3] = bfbits2f(v.y >> 16);
    }
    __device__ __forceinline__ static void store4(bf16* p, const float (&o)[4]) {
        const f32x4_t f = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, __builtin_convertvector(f, bf16x4_t));
    }
    __device__ __forceinline__ static float load1(const bf16* p) { return bf2f(*p); }
    __device__ __forceinline__ static void store1(bf16* p, float v) { *p = f2bf(v); }
    __device__ __forceinline__ static float round(float v) { return bf2f(f2bf(v)); }
};
template <>
struct io<f16> {
    static constexpr int dtype = MOREC_F16;
    __device__ __forceinline__ static void load4(const f16* p, float (&o)[4]) {
        const f16x4_t h = __builtin_bit_cast(f16x4_t, *reinterpret_cast<const uint2*>(p));
        const f32x4_t f = __builtin_convertvector(h, f32x4_t);
        o[0] = f[0]; o[1] = f[1]; o[2] = f[2]; o[3] = f[3];
    }
    __device__ __forceinline__ static void store4(f16* p, const float (&o)[4]) {
        const f32x4_t f = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, __builtin_convertvector(f, f16x4_t));
    }
    __device__ __forceinline__ static float load1(const f16* p) { return h2f(*p); }
    __device__ __forceinline__ static void store1(f16* p, float v) { *p = f2h(v); }
    __device__ __forceinline__ static float round(float v) { return h2f(f2h(v)); }
};

// ---- 16-byte vector IO: EV = 4 (f32) or 8 (bf16) elements per lane and access ------------------------------
template <typename T>
struct vio;
template <>
struct vio<float> {
    static constexpr int EV = 4;
    // load_raw / unpack: the two halves of load, for kernels that issue every load of a row before converting any
    __device__ __forceinline__ static uint4 load_raw(const float* p) { return *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ static void unpack(const uint4& v, float (&o)[4]) {
        o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
    }
    __device__ __forceinline__ static void load(const float* p, float (&o)[4]) { io<float>::load4(p, o); }
    __device__ __forceinline__ static void store(float* p, const float (&o)[4]) { io<float>::store4(p, o); }
};
template <>
struct vio<bf16> {
    static constexpr int EV = 8;
    __device__ __forceinline__ static uint4 load_raw(const bf16* p) { return *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ static void unpack(const uint4& v, float (&o)[8]) {
        o[0] = bfbits2f(v.x & 0xffffu); o[1] = bfbits2f(v.x >> 16); o[2] = bfbits2f(v.y & 0xffffu); o[3] = bfbits2f(v.y >> 16);
        o[4] = bfbits2f(v.z & 0xffffu); o[5] = bfbits2f(v.z >> 16); o[6] = bfbits2f(v.w & 0xffffu); o[7] = bfbits2f(v.w >> 16);
    }
    __device__ __forceinline__ static void load(const bf16* p, float (&o)[8]) { unpack(load_raw(p), o); }
    __device__ __forceinline__ static void store(bf16* p, const float (&o)[8]) {
        uint4 v;
        v.x = pack_bf16x2(o[0], o[1]); v.y = pack_bf16x2(o[2], o[3]); v.z = pack_bf16x2(o[4], o[5]); v.w = pack_bf16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(p) = v;
    }
};
template <>
struct vio<f16> {
    static constexpr int EV = 8;
    __device__ __forceinline__ static uint4 load_raw(const f16* p) { return *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ static void unpack(const uint4& v, float (&o)[8]) {
        const f16x8_t h = __builtin_bit_cast(f16x8_t, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (float)h[k];
    }
    __device__ __forceinline__ static void load(const f16* p, float (&o)[8]) { unpack(load_raw(p), o); }
    __device__ __forceinline__ static void store(f16* p, const float (&o)[8]) {
        uint4 v;
        v.x = pack_f16x2(o[0], o[1]); v.y = pack_f16x2(o[2], o[3]); v.z = pack_f16x2(o[4], o[5]); v.w = pack_f16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(p) = v;
    }
};
// ---- arithmetic selection for the kernels whose data movement only sees 2-byte elements (MFMA main loops, attention, scoring):
// T16 = bf16 | f16 picks the conversions and the MFMA opcode; fragments travel as raw 16-byte vectors (bf16x8_t = "8 x 16 bits").
typedef __attribute__((ext_vector_type(16))) float f32x16c_t;
template <typename T16>
struct h16;
template <>
struct h16<bf16> {
    static constexpr int dtype = MOREC_BF16;
    __device__ __forceinline__ static float bits2f(uint32_t lo16) { return bfbits2f(lo16); }
    __device__ __forceinline__ static uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    __device__ __forceinline__ static unsigned short bits(float f) { return f2bf_bits(f); }
    __device__ __forceinline__ static f32x4_t mma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ __forceinline__ static f32x16c_t mma32(bf16x8_t a, bf16x8_t b, f32x16c_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct h16<f16> {
    static constexpr int dtype = MOREC_F16;
    __device__ __forceinline__ static float bits2f(uint32_t lo16) { return hbits2f(lo16); }
    __device__ __forceinline__ static uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
    __device__ __forceinline__ static unsigned short bits(float f) { return f2h_bits(f); }
    __device__ __forceinline__ static f32x4_t mma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    __device__ __forceinline__ static f32x16c_t mma32(bf16x8_t a, bf16x8_t b, f32x16c_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
// EV consecutive fp32 values (parameters: gamma / beta / bias / position rows)
template <int EV>
__device__ __forceinline__ void load_f32v(const float* p, float (&o)[EV]) {
#pragma unroll
    for (int q = 0; q < EV / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
        o[4 * q] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
    }
}
template <int EV>
__device__ __forceinline__ void store_f32v(float* p, const float (&o)[EV]) {
#pragma unroll
    for (int q = 0; q < EV / 4; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

// ---- wave-level reductions (64 lanes) ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Butterfly steps over lane ^ 16 and lane ^ 32 in the VALU: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16-lane rows
// (the upper 32 lanes) of one register with the even rows (the lower 32 lanes) of another, so with both registers holding v the pair
// afterwards holds {v[lane & ~16], v[lane | 16]} ({v[lane & 31], v[lane | 32]}) and one add / max finishes the step -- instead of a
// ds_bpermute round trip through the LDS crossbar (~120 cycles, fully exposed in the one-wave-per-SIMD attention kernels, six per query
// block).  Inline asm: given the same value twice, the builtin form folds its two results into one (hipcc 7.2); the s_nop covers the
// VALU-write -> permlane-read wait states the compiler cannot see into.
__device__ __forceinline__ void permlane16_swap(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void permlane32_swap(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// sum / max over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} (every lane receives the result)
__device__ __forceinline__ float rows4_sum(float v) {
    float a = v, b = v;
    permlane16_swap(a, b);
    a += b;
    b = a;
    permlane32_swap(a, b);
    return a + b;
}
__device__ __forceinline__ float rows4_max(float v) {
    float a = v, b = v;
    permlane16_swap(a, b);
    a = fmaxf(a, b);
    b = a;
    permlane32_swap(a, b);
    return fmaxf(a, b);
}

// erf GELU (HF "gelu" / nn.GELU: 0.5 x (1 + erf(x / sqrt 2))) and its derivative.
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. at the fp32 rounding level of the result): one v_rcp, one
// v_exp and five FMAs instead of OCML erff's ~50-instruction expansion -- at 248 M activations per FFN layer the
// library erf alone cost as much VALU time as the GEMM main loop it is fused behind.  exp(-x^2/2) is shared between
// erf(x/sqrt2) and the Gaussian pdf of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf_unnorm) {
    const float ax = fabsf(x) * 0.70710678118654752440f;            // |x| / sqrt(2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float y = fmaf(1.061405429f, t, -1.453152027f);
    y = fmaf(y, t, 1.421413741f);
    y = fmaf(y, t, -0.284496736f);
    y = fmaf(y, t, 0.254829592f);
    y *= t;
    const float e = __expf(-ax * ax);                               // exp(-x^2 / 2)
    const float erf_abs = 1.0f - y * e;                             // erf(|x| / sqrt 2)
    cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    pdf_unnorm = e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float dgelu_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}
// Two elements at a time: the polynomial runs on v_pk_fma_f32 / v_pk_mul_f32 (half the VALU issue slots of the scalar form);
// v_rcp / v_exp stay per element.  Same arithmetic as gelu_parts, element for element.
__device__ __forceinline__ void gelu_parts2(f32x2_t x, f32x2_t& cdf, f32x2_t& pdf_unnorm) {
    const f32x2_t ax = {fabsf(x[0]) * 0.70710678118654752440f, fabsf(x[1]) * 0.70710678118654752440f};
    const f32x2_t den = ax * 0.3275911f + 1.0f;
    const f32x2_t t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2_t y = t * 1.061405429f + (-1.453152027f);
    y = y * t + 1.421413741f;
    y = y * t + (-0.284496736f);
    y = y * t + 0.254829592f;
    y = y * t;
    const f32x2_t na = -(ax * ax);
    const f32x2_t e = {__expf(na[0]), __expf(na[1])};
    const f32x2_t erf_abs = 1.0f - y * e;
    cdf = f32x2_t{copysignf(erf_abs[0], x[0]), copysignf(erf_abs[1], x[1])} * 0.5f + 0.5f;
    pdf_unnorm = e;
}
__device__ __forceinline__ void gelu4(float (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2_t x = {v[2 * q], v[2 * q + 1]};
        f32x2_t cdf, e;
        gelu_parts2(x, cdf, e);
        const f32x2_t r = x * cdf;
        v[2 * q] = r[0]; v[2 * q + 1] = r[1];
    }
}
// v[k] = gelu'(v[k])
__device__ __forceinline__ void dgelu4(float (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2_t x = {v[2 * q], v[2 * q + 1]};
        f32x2_t cdf, e;
        gelu_parts2(x, cdf, e);
        const f32x2_t d = (x * 0.39894228040143267794f) * e + cdf;
        v[2 * q] = d[0]; v[2 * q + 1] = d[1];
    }
}
// v[k] = gelu(v[k]), d[k] = gelu'(v[k]): one evaluation of the shared erf / exp for both
__device__ __forceinline__ void gelu4_with_deriv(float (&v)[4], float (&d)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2_t x = {v[2 * q], v[2 * q + 1]};
        f32x2_t cdf, e;
        gelu_parts2(x, cdf, e);
        const f32x2_t g = x * cdf;
        const f32x2_t dd = (x * 0.39894228040143267794f) * e + cdf;
        v[2 * q] = g[0]; v[2 * q + 1] = g[1];
        d[2 * q] = dd[0]; d[2 * q + 1] = dd[1];
    }
}
// v[k] *= gelu'(u[k])
__device__ __forceinline__ void dgelu4_mul(float (&v)[4], const float (&u)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2_t x = {u[2 * q], u[2 * q + 1]};
        f32x2_t cdf, e;
        gelu_parts2(x, cdf, e);
        const f32x2_t d = (x * 0.39894228040143267794f) * e + cdf;
        v[2 * q] *= d[0]; v[2 * q + 1] *= d[1];
    }
}

// ---- counter-based dropout RNG: a pure function of (seed, element index), so the backward pass regenerates
// the forward mask instead of storing it.  One 32-bit avalanche hash (lowbias32 round, seed folded in twice) serves a PAIR
// of consecutive elements: its two 16-bit halves are compared with p * 2^16.  (Round 1 hashed every element with six
// 32-bit multiplies -- quarter-rate v_mul_lo_u32 -- which was ~25 % of the LayerNorm-backward and a visible part of the
// attention kernels; this is 1.5 multiplies per element.)
struct DropRng {
    uint32_t s0, s1, thresh;   // keep iff 16-bit draw >= thresh, thresh = round-down(p * 2^16)
    float inv_keep;            // 1 / (1 - thresh / 2^16): exactly unbiased for the probability actually applied
    const uint64_t* src;       // device word XOR-ed into (s0, s1) at kernel entry (morec_dropout_seed_source), or null
};
// The process-wide seed source (capi.hip; include/morec_hip.h: morec_dropout_seed_source): with it, the masks of a launch are a function of
// (seed argument, *source) -- a captured graph whose seed ARGUMENTS are frozen still draws fresh masks at every replay.
const uint64_t* morec_drop_seed_src();
inline DropRng make_drop(float p, uint64_t seed) {
    DropRng d;
    d.src = p > 0.f ? morec_drop_seed_src() : nullptr;
    d.s0 = (uint32_t)seed;
    d.s1 = (uint32_t)(seed >> 32);
    const double t = (double)p * 65536.0;
    d.thresh = p <= 0.f ? 0u : (t >= 65535.0 ? 65535u : (t < 1.0 ? 1u : (uint32_t)t));
    d.inv_keep = 1.0f / (1.0f - (float)d.thresh / 65536.0f);
    return d;
}
// kernel entry: fold the device-resident seed word in (one scalar load; a no-op without a source)
__device__ __forceinline__ DropRng drop_resolve(DropRng d) {
    if (d.src) {
        const uint64_t v = *d.src;
        d.s0 ^= (uint32_t)v;
        d.s1 ^= (uint32_t)(v >> 32);
    }
    return d;
}
// hash of element pair `pair` (= element index >> 1)
__host__ __device__ inline uint32_t drop_hash(const DropRng& d, uint64_t pair) {
    const uint32_t hi = (uint32_t)(pair >> 32);
    uint32_t h = ((uint32_t)pair * 0x9E3779B1u) ^ d.s0 ^ (hi + ((hi << 16) | (hi >> 16)));
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h += d.s1; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__host__ __device__ inline bool drop_keep(const DropRng& d, uint64_t idx) {
    const uint32_t h = drop_hash(d, idx >> 1);
    return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= d.thresh;
}
// keep flags of EV consecutive elements starting at an EVEN index: one hash per pair
template <int EV>
__device__ __forceinline__ void drop_keep_vec(const DropRng& d, uint64_t idx0, bool (&keep)[EV]) {
#pragma unroll
    for (int k = 0; k < EV; k += 2) {
        const uint32_t h = drop_hash(d, (idx0 + k) >> 1);
        keep[k] = (h & 0xffffu) >= d.thresh;
        keep[k + 1] = (h >> 16) >= d.thresh;
    }
}

// ---- hazard guard behind a 16-byte global / buffer STORE in a kernel that keeps issuing MFMAs around it.  hipcc separates such a store from the
// next VALU write of its data registers by the two wait states the ISA asks for, and counts an interleaved v_mfma as one of them; measured on
// gfx950 (scripts/race_probe2.py, profiles/r04_store_hazard_probe.txt): with the memory pipe backed up by a second stream's kernel, dword 0 of
// the store data of 16 lanes was occasionally the NEXT value of that register -- 4 token rows of Swin attention gradients replaced by raw fp32
// bit patterns, a NaN in the weight gradient every few steps.  Three idle issue slots behind the store (nothing the compiler may move) close it.
__device__ __forceinline__ void store_b128_guard() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 2" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ---- packed ("varlen") token layouts with SPARE rows: the blocks a launcher appends behind its (sequence, head) grid zero the rows
// [cu[n_seq], total_rows) of a row-major output, 16 rows per block -- rows no sequence owns (a layout padded up to a bucket size so that
// one captured graph serves every batch of the bucket) then hold exact zeros instead of whatever the allocator left there, and the
// weight-gradient products that sum over ALL rows are unaffected by them.
__device__ __forceinline__ void zero_dead_rows(void* base, const int32_t* cu, int n_seq, int total_rows, size_t rowbytes, int eblock) {
    const int r0 = cu[n_seq] + eblock * 16;
    const int r1 = min(r0 + 16, total_rows);
    if (r0 >= r1) return;
    uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + (size_t)r0 * rowbytes);
    const size_t n16 = (size_t)(r1 - r0) * rowbytes / 16;
    for (size_t i = threadIdx.x; i < n16; i += blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}

// ---- host-side argument checks -------------------------------------------------------------------------
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int elt_size(int dtype) { return dtype == MOREC_F32 ? 4 : 2; }
static inline bool is_h16(int dtype) { return dtype == MOREC_BF16 || dtype == MOREC_F16; }
// host-side dtype dispatch: f(tag) with tag a null T* of the storage type; false = unknown dtype code
template <typename F>
static inline bool by_dtype(int dtype, F&& f) {
    switch (dtype) {
        case MOREC_F32: f((float*)nullptr); return true;
        case MOREC_BF16: f((bf16*)nullptr); return true;
        case MOREC_F16: f((f16*)nullptr); return true;
        default: return false;
    }
}
// the same over the two 16-bit types only
template <typename F>
static inline bool by_h16(int dtype, F&& f) {
    switch (dtype) {
        case MOREC_BF16: f((bf16*)nullptr); return true;
        case MOREC_F16: f((f16*)nullptr); return true;
        default: return false;
    }
}
#define MOREC_TAG_T(tag) typename std::remove_pointer<decltype(tag)>::type
#define MOREC_CHECK_LAUNCH()                        \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

// ---- deterministic mode (MOREC_DETERMINISTIC=1 / morec_tuning_set("deterministic", 1); capi.hip).  The reference sets torch's
// deterministic flags (T/run.py:313-314).  The kernels that end in fp32 atomics -- column sums of LayerNorm / bias gradients, the
// embedding-table scatters -- then leave per-block partials in a library-owned scratch instead and a second launch adds them to the
// destination in a FIXED order (or, for the scatters, one wave owns a destination row and sums its sources in index order): two runs of
// the same step on the same inputs give the same bits.  Covered: the 16-bit and fp32 text / ID towers.
bool morec_deterministic();
void morec_set_deterministic(int on);
// fp32 scratch of at least n floats for launches on stream s (one buffer per (device, stream): launches of one stream are ordered, two
// streams must not share); nullptr on allocation failure.  Grows by re-allocation behind a stream synchronisation.
float* morec_det_scratch(hipStream_t s, size_t n);
// dst[j] += part[0 * stride + j] + part[1 * stride + j] + ... (p ascending; j < n): plain read-modify-write, one thread per element
int morec_det_fold_add(const float* part, float* dst, int n_parts, size_t n, size_t stride, hipStream_t s);

// XCD-aware workgroup -> tile mapping: the dispatcher round-robins consecutive workgroup ids over
// the 8 XCDs (private L2 each); remap so that each XCD walks a CONTIGUOUS run of tiles, i.e. the
// tiles that share an A row-panel hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

