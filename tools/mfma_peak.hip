// Measurement aid: sustained v_mfma_f32_16x16x4_f32 rate of the whole chip with no memory traffic
// (16 independent accumulator tiles per wave, like the GEMM main loop).  Gives the practical ceiling
// that roofline fractions in DESIGN.md section 6 are discussed against (the nominal 157.3 TFLOP/s
// assumes 2.4 GHz sustained).   build: hipcc --offload-arch=gfx950 -O2 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0)
{
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main(int argc, char **argv)
{
    const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 1;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = rep == 0 ? 200 : (rep == 1 ? 4000 : 40000);      // ~0.03 ms, ~0.7 ms, ~7 ms of MFMA
        hipLaunchKernelGGL(mfma_loop, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 1e-3f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 1e-3f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * wgs_per_cu * 4 * iters * 64 * 2048.0;
        printf("cus=%d wgs/cu=%d iters=%d  %.3f ms  %.1f TFLOP/s  (clock if 100%% dense: %.2f GHz)\n", cus, wgs_per_cu, iters, ms,
               flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 2.4);
    }
    return 0;
}
