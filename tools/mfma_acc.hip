// Measurement aid: v_mfma_f32_16x16x4_f32 issue rate of ONE wave per SIMD as a function of the number of accumulator tiles in
// rotation (the wave tile of a GEMM) and of the operand pattern (same A/B registers every time, or a fresh A / B register per
// k step as in a real k block).  Prints cycles per MFMA from s_memtime.   build: hipcc --offload-arch=gfx950 -O2 tools/mfma_acc.hip -o tools/mfma_acc
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC, int FRESH>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *t, int iters, float a0)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a = {a0 + threadIdx.x * 1e-6f, a0 * 2, a0 * 3, a0 * 4}, b = {1e-3f, 2e-3f, 3e-3f, 4e-3f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(FRESH ? a[j] : a[0], FRESH ? b[(j + q) & 3] : b[0], acc[q], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}

template <int NACC, int FRESH> void run(float *out, unsigned long long *t, int wgs)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, FRESH>), dim3(wgs), dim3(256), 0, 0, out, t, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("acc tiles %2d  fresh operands %d  workgroups %4d : %.1f cycles per MFMA\n", NACC, FRESH, wgs, (double)h / (iters * 4.0 * NACC));
}

int main()
{
    float *out; unsigned long long *t; hipMalloc(&out, 4096); hipMalloc(&t, 8);
    for (int wgs : {256, 512}) {
        run<1, 0>(out, t, wgs); run<2, 0>(out, t, wgs); run<4, 0>(out, t, wgs); run<8, 0>(out, t, wgs); run<16, 0>(out, t, wgs);
        run<2, 1>(out, t, wgs); run<4, 1>(out, t, wgs); run<8, 1>(out, t, wgs); run<16, 1>(out, t, wgs);
    }
    return 0;
}
