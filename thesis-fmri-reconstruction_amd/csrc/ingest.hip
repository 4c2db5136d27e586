// Device-side tail of the reference's image pipeline: a batch of already cropped/resized uint8 HWC
// images -> the engine's fp16 NHWC(8) input (and optionally the fp32 NCHW tensor the module API takes), fusing
//   RandomHorizontalFlip (train_vgan_stage1.py:166)  /  RandomShift = scipy.ndimage.shift(order=0, mode='nearest')
//   by whole pixels (data_preprocessing/data_loader.py:186-217)  ->  ToTensor (u8 / 255)  ->  GreyToColor
//   (data_loader.py:374-401: 1 -> 3 channels)  ->  Normalize(mean, std) (train_vgan_stage1.py:169).
// The random draws (flip flag, shift) are made by the caller and passed per image, so the kernel is deterministic.
// One 16-byte store per pixel; HBM bound and tiny (256 images of 64x64: 3 MB in, 8 MB out).
#include "kernels.h"

namespace fmri {

__global__ __launch_bounds__(256) void ingest_u8_kernel(const uint8_t* __restrict__ src, int N, int H, int W, int C,
                                                        const int* __restrict__ flip, const int* __restrict__ shift,
                                                        float m0, float m1, float m2, float r0, float r1, float r2,
                                                        half_t* __restrict__ dst16, float* __restrict__ dst32) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int n = (int)(i / ((int64_t)W * H));
        // output pixel (y, x) of the shifted image = input pixel (y - sy, x - sx), clamped to the edge ('nearest');
        // the flip is applied before the shift (transform order of the scripts)
        int sy = 0, sx = 0;
        if (shift) { sy = shift[2 * n]; sx = shift[2 * n + 1]; }
        int yy = y - sy, xx = x - sx;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        if (flip && flip[n]) xx = W - 1 - xx;
        const uint8_t* p = src + ((int64_t)(n * H + yy) * W + xx) * C;
        const float v0 = p[0] * (1.f / 255.f);
        const float v1 = C == 3 ? p[1] * (1.f / 255.f) : v0;
        const float v2 = C == 3 ? p[2] * (1.f / 255.f) : v0;
        const float o0 = (v0 - m0) * r0, o1 = (v1 - m1) * r1, o2 = (v2 - m2) * r2;
        if (dst16) {
            h8 o = {(half_t)o0, (half_t)o1, (half_t)o2, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f,
                    (half_t)0.f};
            *(h8*)(dst16 + i * 8) = o;
        }
        if (dst32) {
            const int64_t hw = (int64_t)H * W, b = (int64_t)n * 3 * hw + (int64_t)y * W + x;
            dst32[b] = o0; dst32[b + hw] = o1; dst32[b + 2 * hw] = o2;
        }
    }
}

int ingest_u8_launch(const uint8_t* src, int N, int H, int W, int C, const int* flip, const int* shift,
                     const float* mean3, const float* std3, half_t* dst16, float* dst32, hipStream_t st) {
    const int64_t total = (int64_t)N * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ingest_u8_kernel, dim3(blocks), dim3(256), 0, st, src, N, H, W, C, flip, shift, mean3[0],
                       mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2], dst16, dst32);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
