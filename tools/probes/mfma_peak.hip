// Bare v_mfma_f32_16x16x32_f16 loop: what the matrix pipes of this device sustain on random operands with no memory
// traffic at all (calibration for the roofline fractions in DESIGN.md; the clock under MFMA load is below the 2.4 GHz
// the 2.5 PFLOP/s datasheet peak assumes).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/probes/mfma_peak.hip && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *(const h8*)(src + ((size_t)t * 8 + i) * 8 % (1 << 20));
        b[i] = *(const h8*)(src + ((size_t)t * 8 + 4 + i) * 8 % (1 << 20));
    }
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    f4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[t] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
static void run(int blocks_per_cu, const _Float16* src, float* out) {
    const int iters = 20000;
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * NACC * 2.0 * 16 * 16 * 32;
        if (rep == 2)
            printf("16 accumulators/wave: %d  waves/SIMD: %d  %.1f ms  %.0f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n",
                   NACC, blocks_per_cu, ms, flop / ms / 1e9,
                   2.4e9 * ms * 1e-3 / ((double)iters * NACC * blocks_per_cu));
    }
}

int main() {
    std::vector<_Float16> h(1 << 20);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    _Float16* src; float* out;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<16>(1, src, out);
    run<16>(2, src, out);
    run<8>(2, src, out);
    run<16>(4, src, out);
    return 0;
}
