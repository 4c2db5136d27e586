// Probe: what bounds the BatchNorm streaming passes (4.2-5.8 TB/s in the step while the flat RMSprop kernel reaches 6.8)?
// y = relu(x * a[c] + b[c]) on an fp16 [M][C] tensor (the forward apply pass: 2 B read + 2 B written per element) in the
// shipped row mapping and in variants: non-temporal stores / loads, contiguous row ranges per block, deeper unrolling,
// grid sizes.  Also the two-input backward form (x, g -> dx: 4 B read + 2 B written).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bn_stream_bw tools/probes/bn_stream_bw.hip && /tmp/bn_stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NT_ST, int NT_LD>
__device__ __forceinline__ void st8(h8* p, const h8& v) {
    if (NT_ST) __builtin_nontemporal_store(v, p); else *p = v;
}
template <int NT_LD>
__device__ __forceinline__ h8 ld8(const h8* p) {
    if (NT_LD) return __builtin_nontemporal_load(p);
    return *p;
}
__device__ __forceinline__ h8 fn(const h8& x, const float* a, const float* b) {
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { float f = (float)x[j] * a[j] + b[j]; o[j] = (half_t)(f > 0.f ? f : 0.f); }
    return o;
}
__device__ __forceinline__ h8 fn2(const h8& x, const h8& g, const float* a, const float* b) {
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xh = ((float)x[j] - b[j]) * a[j];
        float gg = (float)g[j];
        if (!(xh > 0.f)) gg = 0.f;
        o[j] = (half_t)(a[j] * (gg - 0.01f - xh * 0.02f));
    }
    return o;
}

// MAP 0: shipped interleaved sweep (row = by*RY + ry + k*gridDim.y*RY), U-fold unrolled; MAP 1: block owns a contiguous
// row range.  IN2: second input tensor.
template <int U, int MAP, int NT_ST, int NT_LD, int IN2>
__global__ __launch_bounds__(256) void k(const half_t* __restrict__ x, const half_t* __restrict__ g, half_t* __restrict__ y,
                                         int M, int C, int cx_log2, const float* __restrict__ pa, const float* __restrict__ pb) {
    const int CX = 1 << cx_log2, RY = 256 >> cx_log2;
    const int cx = threadIdx.x & (CX - 1), ry = threadIdx.x >> cx_log2;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = pa[cx * 8 + j]; b[j] = pb[cx * 8 + j]; }
    const int64_t coff = cx * 8;
    int m, stride, end;
    if (MAP == 0) { m = blockIdx.x * RY + ry; stride = gridDim.x * RY; end = M; }
    else { const int per = (M + gridDim.x - 1) / gridDim.x; m = blockIdx.x * per + ry; stride = RY; end = min(M, (int)(blockIdx.x + 1) * per); }
    for (; m + (U - 1) * stride < end; m += U * stride) {
        h8 xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = ld8<NT_LD>((const h8*)(x + (int64_t)(m + u * stride) * C + coff));
            if (IN2) gv[u] = ld8<NT_LD>((const h8*)(g + (int64_t)(m + u * stride) * C + coff));
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            st8<NT_ST, 0>((h8*)(y + (int64_t)(m + u * stride) * C + coff), IN2 ? fn2(xv[u], gv[u], a, b) : fn(xv[u], a, b));
    }
    for (; m < end; m += stride) {
        h8 xv = ld8<NT_LD>((const h8*)(x + (int64_t)m * C + coff)), gv = xv;
        if (IN2) gv = ld8<NT_LD>((const h8*)(g + (int64_t)m * C + coff));
        st8<NT_ST, 0>((h8*)(y + (int64_t)m * C + coff), IN2 ? fn2(xv, gv, a, b) : fn(xv, a, b));
    }
}

template <class K>
static float run(K kern, int blocks, const half_t* x, const half_t* g, half_t* y, int M, int C, const float* a, const float* b) {
    int cx = 0;
    while ((1 << cx) < C / 8) ++cx;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, g, y, M, C, cx, a, b);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    const int C = 128, M = 3 * 256 * 32 * 32;
    const int64_t n = (int64_t)M * C;
    half_t *x, *g, *y;
    float *a, *b;
    hipMalloc(&x, n * 2); hipMalloc(&g, n * 2); hipMalloc(&y, n * 2);
    hipMalloc(&a, 4 * C); hipMalloc(&b, 4 * C);
    hipMemset(x, 0x3c, n * 2); hipMemset(g, 0x38, n * 2); hipMemset(a, 0, 4 * C); hipMemset(b, 0, 4 * C);
    const int grids[] = {512, 1024, 2048, 4096, 8192, 16384};
#define ROW(name, K1, bytes_per)                                                                       \
    for (int gsz : grids) {                                                                            \
        const float ms = run(K1, gsz, x, g, y, M, C, a, b);                                            \
        printf("%-46s grid %5d: %7.1f us  %5.2f TB/s\n", name, gsz, ms * 1e3, bytes_per * n / (ms * 1e-3) / 1e12); \
    }
    ROW("apply  U4 interleaved (shipped)", (k<4, 0, 0, 0, 0>), 4.0)
    ROW("apply  U4 interleaved nt-store", (k<4, 0, 1, 0, 0>), 4.0)
    ROW("apply  U4 interleaved nt-store nt-load", (k<4, 0, 1, 1, 0>), 4.0)
    ROW("apply  U8 interleaved", (k<8, 0, 0, 0, 0>), 4.0)
    ROW("apply  U8 interleaved nt-store", (k<8, 0, 1, 0, 0>), 4.0)
    ROW("apply  U4 contiguous ranges", (k<4, 1, 0, 0, 0>), 4.0)
    ROW("apply  U4 contiguous ranges nt-store", (k<4, 1, 1, 0, 0>), 4.0)
    ROW("bwd    U4 interleaved (shipped form)", (k<4, 0, 0, 0, 1>), 6.0)
    ROW("bwd    U4 interleaved nt-store", (k<4, 0, 1, 0, 1>), 6.0)
    ROW("bwd    U4 interleaved nt-store nt-load", (k<4, 0, 1, 1, 1>), 6.0)
    ROW("bwd    U2 interleaved nt-store", (k<2, 0, 1, 0, 1>), 6.0)
    ROW("bwd    U4 contiguous ranges nt-store", (k<4, 1, 1, 0, 1>), 6.0)
    hipMemcpyAsync(y, x, n * 2, hipMemcpyDeviceToDevice, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0); hipMemcpyAsync(y, x, n * 2, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy D2D %lld MB: %7.1f us  %5.2f TB/s (read + write)\n", (long long)(n * 2 >> 20), ms * 1e3, 4.0 * n / (ms * 1e-3) / 1e12);
    return 0;
}
