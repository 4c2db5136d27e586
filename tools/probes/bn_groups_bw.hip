// Probe: BatchNorm backward = a reduction pass over (x, g) followed by an apply pass over (x, g) -> dx.  On the large
// layers x and g together outgrow the 256 MB Infinity Cache, so the apply pass re-reads everything from HBM.  Channels are
// independent: if both passes run on ONE GROUP of channels at a time (row segments of Cg channels, row stride C), the
// group's working set fits the cache and the apply pass of a group finds its inputs there.  Time of the whole backward
// (all groups) per group count, for the shapes of the step's big layers.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bn_groups_bw tools/probes/bn_groups_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// rows x Cg channels at column offset c0 of a [M][C] tensor; NG inputs g (stacked tensors, M*C apart)
template <int APPLY, int NG>
__global__ __launch_bounds__(256) void pass(const half_t* __restrict__ x, const half_t* __restrict__ g, half_t* __restrict__ y,
                                            int M, int C, int c0, int cx_log2, float* __restrict__ part) {
    const int CX = 1 << cx_log2, RY = 256 >> cx_log2;
    const int cx = threadIdx.x & (CX - 1), ry = threadIdx.x >> cx_log2;
    const int64_t coff = c0 + cx * 8, sb = (int64_t)M * C;
    const int stride = gridDim.x * RY;
    float s0[8] = {}, s1[8] = {};
    for (int m = blockIdx.x * RY + ry; m < M; m += 2 * stride) {
        h8 xv[2], gv[2][NG];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int mm = m + u * stride;
            if (mm < M) {
                xv[u] = *(const h8*)(x + (int64_t)mm * C + coff);
#pragma unroll
                for (int q = 0; q < NG; ++q) gv[u][q] = *(const h8*)(g + q * sb + (int64_t)mm * C + coff);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int mm = m + u * stride;
            if (mm >= M) continue;
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                h8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (float)xv[u][j] * 0.5f;
                    float gg = (float)gv[u][q][j];
                    if (!(xh > -1.f)) gg = 0.f;
                    if (APPLY) o[j] = (half_t)(0.7f * (gg - 0.01f - xh * 0.02f));
                    else { s0[j] += gg; s1[j] += gg * xh; }
                }
                if (APPLY) *(h8*)(y + q * sb + (int64_t)mm * C + coff) = o;
            }
        }
    }
    if (!APPLY) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += s0[j] + s1[j];
        if (t == 1234.5f) part[0] = t;
    }
}

template <int NG>
static float backward(const half_t* x, const half_t* g, half_t* y, int M, int C, int groups, float* part) {
    const int Cg = C / groups;
    int cx = 0;
    while ((1 << cx) < Cg / 8) ++cx;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < groups; ++k) {
            hipLaunchKernelGGL((pass<0, NG>), dim3(768), dim3(256), 0, 0, x, g, y, M, C, k * Cg, cx, part);
            hipLaunchKernelGGL((pass<1, NG>), dim3(2048), dim3(256), 0, 0, x, g, y, M, C, k * Cg, cx, part);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    struct Shape { const char* name; int M, C, NG; } shapes[] = {
        {"disc.conv1 out: 3B x 32x32 x 128, two cotangent streams", 3 * 256 * 32 * 32, 128, 2},
        {"disc.conv2 out: 3B x 16x16 x 256, two cotangent streams", 3 * 256 * 16 * 16, 256, 2},
        {"dec.conv1 out : 2B x 32x32 x 128, one stream", 2 * 256 * 32 * 32, 128, 1},
        {"dec.conv0 out : 2B x 16x16 x 256, one stream", 2 * 256 * 16 * 16, 256, 1},
    };
    float* part;
    hipMalloc(&part, 64);
    for (auto& s : shapes) {
        const int64_t n = (int64_t)s.M * s.C;
        half_t *x, *g, *y;
        hipMalloc(&x, n * 2); hipMalloc(&g, n * 2 * s.NG); hipMalloc(&y, n * 2 * s.NG);
        hipMemset(x, 0x3c, n * 2); hipMemset(g, 0x38, n * 2 * s.NG);
        printf("%s  (x + g = %lld MB, dx = %lld MB)\n", s.name, (long long)((n * 2 * (1 + s.NG)) >> 20), (long long)((n * 2 * s.NG) >> 20));
        for (int groups : {1, 2, 4, 8}) {
            if (s.C / groups < 16) continue;
            const float ms = s.NG == 2 ? backward<2>(x, g, y, s.M, s.C, groups, part) : backward<1>(x, g, y, s.M, s.C, groups, part);
            const double bytes = (double)n * 2 * (2 * (1 + s.NG) + s.NG);
            printf("   %d channel group(s) of %3d (%3d-byte row segments): %7.1f us  (%.2f TB/s of the 2-pass byte count)\n", groups,
                   s.C / groups, s.C / groups * 2, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
        }
        hipFree(x); hipFree(g); hipFree(y);
    }
    return 0;
}
