// Probe: does a streaming read that walks a tensor in the REVERSE order of the pass that last touched it hit the Infinity
// Cache?  W writes S bytes front to back; R reads them front to back or back to front.  Build: hipcc --offload-arch=gfx950
// -O3 -o /tmp/mall_order tools/probes/mall_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wr(f4* p, int64_t n, float v) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = (f4){v, v, v, v};
}

__global__ __launch_bounds__(256) void rd(const f4* p, int64_t n, int rev, float* out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    float s = 0.f;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + u * stride;
            v[u] = p[rev ? n - 1 - j : j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n; i += stride) { const f4 v = p[rev ? n - 1 - i : i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const int sizes_mb[] = {64, 128, 192, 256, 400, 800};
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int smb : sizes_mb) {
        const int64_t bytes = (int64_t)smb << 20, n = bytes / 16;
        f4* p;
        hipMalloc(&p, bytes);
        for (int mode = 0; mode < 4; ++mode) {
            // mode 0: W then R fwd; 1: W then R rev; 2: R fwd then R fwd; 3: R fwd then R rev
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                if (mode < 2) hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, p, n, (float)rep);
                else hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, p, n, 0, out);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, p, n, mode & 1, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%4d MB  %s then read %s: %7.1f us  %6.2f TB/s\n", smb, mode < 2 ? "write" : "read ", (mode & 1) ? "rev" : "fwd",
                   best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
        hipFree(p);
    }
    return 0;
}
