// image.hip -- device half of the vision input pipeline: decoded uint8 HWC item images of ARBITRARY size ->
// uint8 [n, R, R, 3], bit for bit what the reference's dataloader workers compute on the host with
//     tv.transforms.Resize((R, R))  on  Image.fromarray(LMDB_Image.get_image())          (V/data_utils/dataset.py:68-73,91-98)
// i.e. Pillow's BILINEAR resampler: a horizontal pass with per-output-column fixed-point taps (2^22 scale, rounded, clipped to
// uint8), then a vertical pass over that uint8 image (Pillow src/libImaging/Resample.c).  The taps depend only on
// (input size, R): the host builds one table per distinct size (idvs/morec_amd/data_utils/images.py::resize_table, double
// arithmetic exactly as Pillow's) and this kernel does the two integer passes fused -- each output pixel re-derives the
// horizontally resampled value of the <= ksize input rows it needs.  ToTensor + Normalize(0.5, 0.5) stay fused in
// morec_swin_patchify_u8.  The batch crosses PCIe as uint8 at its native size instead of 424 MB of fp32 per step.
#include "common.hpp"

namespace {
constexpr int PRECISION_BITS = 22;
__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}
// table layout (int32): [0] = ksize, then R rows of (first input index, tap count, taps[ksize])
__global__ __launch_bounds__(256) void image_resize_kernel(const uint8_t* __restrict__ src, const long long* __restrict__ meta,
                                                           const int32_t* __restrict__ tables, uint8_t* __restrict__ out, int R) {
    const int img = blockIdx.x, oy = blockIdx.y;
    const long long* m = meta + (size_t)img * 5;
    const uint8_t* im = src + m[0];
    const int W = (int)m[2];
    const int32_t* ht = tables + m[3];
    const int32_t* vt = tables + m[4];
    const int kh = ht[0], kv = vt[0];
    const int32_t* vrow = vt + 1 + oy * (2 + kv);
    const int y0 = vrow[0], ny = vrow[1];
    for (int ox = threadIdx.x; ox < R; ox += blockDim.x) {
        const int32_t* hrow = ht + 1 + ox * (2 + kh);
        const int x0 = hrow[0], nx = hrow[1];
        int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
        for (int ty = 0; ty < ny; ++ty) {
            const uint8_t* row = im + ((size_t)(y0 + ty) * W + x0) * 3;
            int h[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
            for (int tx = 0; tx < nx; ++tx) {
                const int k = hrow[2 + tx];
                h[0] += row[tx * 3 + 0] * k;
                h[1] += row[tx * 3 + 1] * k;
                h[2] += row[tx * 3 + 2] * k;
            }
            const int k = vrow[2 + ty];
            acc[0] += clip8(h[0]) * k;
            acc[1] += clip8(h[1]) * k;
            acc[2] += clip8(h[2]) * k;
        }
        uint8_t* o = out + (((size_t)img * R + oy) * R + ox) * 3;
        o[0] = (uint8_t)clip8(acc[0]);
        o[1] = (uint8_t)clip8(acc[1]);
        o[2] = (uint8_t)clip8(acc[2]);
    }
}
}  // namespace

extern "C" int morec_image_resize_u8(const uint8_t* src, const int64_t* meta, const int32_t* tables, uint8_t* out, int n, int R,
                                     void* stream) {
    if (!src || !meta || !tables || !out || n <= 0 || R <= 0) return MOREC_E_ARG;
    if (R > 4096) return MOREC_E_UNSUPPORTED;
    hipLaunchKernelGGL(image_resize_kernel, dim3(n, R), dim3(R <= 64 ? 64 : (R <= 128 ? 128 : 256)), 0,
                       reinterpret_cast<hipStream_t>(stream), src, reinterpret_cast<const long long*>(meta), tables, out, R);
    MOREC_CHECK_LAUNCH();
    return MOREC_OK;
}
