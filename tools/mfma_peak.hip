// Measurement aid: sustained MFMA rate of the whole chip with no memory traffic, AND where a shortfall against the data-sheet peak comes
// from: every wave counts its own shader cycles (s_memtime) and real time (s_memrealtime, 100 MHz) around the loop, so the output splits
//   achieved TFLOP/s  =  (flop per cycle per SIMD: the pipe's own rate; 64 for v_mfma_f32_16x16x4_f32, 1024 for v_mfma_f32_16x16x32_f16)
//                        x  (effective shader clock during the run)  x  1024 SIMDs.
// Round 5 quoted 136 .. 149 TF for fp32 from this tool's first version and called it "the sustained rate"; MI355X_MICROARCH.md measures
// 155 TF (99 % of 157.3).  With the split the two reconcile or not on the spot: pipe rate at ~100 % and a clock below 2.4 GHz = power
// management, not the probe.  16 independent accumulators per wave (the dependent-issue latency of 16x16x4 is 40
// cycles against 32 of issue: 2 chains would do), 1 / 2 / 4 waves per SIMD, no s_waitcnt in the loop, three run lengths.
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_peak.hip -o tools/mfma_peak      usage: tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

template <int F16>
__global__ __launch_bounds__(256) void mfma_loop(unsigned long long *stamps, int iters, float a0, float b0)
{
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i * 1e-3f); hb[i] = (_Float16)(b + i * 1e-3f); }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (F16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[w * 2] = c1 - c0; stamps[w * 2 + 1] = r1 - r0;
        if (s == 123.456f) stamps[0] = 0;
    }
}

template <int F16> static void run(int cus, const char *name, double flop_per_mfma, double ideal_per_clk_simd, double peak_tf)
{
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
        const size_t waves = (size_t)cus * wgs_per_cu * 4;
        unsigned long long *stamps; hipMalloc(&stamps, waves * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = (rep == 0 ? 200 : (rep == 1 ? 4000 : 40000)) * (F16 ? 2 : 1);      // ~0.03 ms, ~0.7 ms, ~7 ms per wave/SIMD
            hipLaunchKernelGGL(mfma_loop<F16>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, stamps, iters, 1.0f, 1e-3f);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_loop<F16>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, stamps, iters, 1.0f, 1e-3f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(waves * 2);
            hipMemcpy(h.data(), stamps, waves * 16, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (size_t w = 0; w < waves; ++w) { cyc += (double)h[w * 2]; real += (double)h[w * 2 + 1]; }
            cyc /= (double)waves; real /= (double)waves;
            const double mfmas = (double)iters * 64.0;                                  // per wave
            const double per_clk_simd = mfmas * flop_per_mfma * wgs_per_cu / cyc;        // waves of one SIMD share its pipe
            const double clock_ghz = cyc / (real * 10.0);                               // s_memrealtime ticks are 10 ns
            const double tf = (double)waves * mfmas * flop_per_mfma / ms / 1e9;
            printf("%-24s waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s = %.3f of %.1f | pipe %.1f flop/clk/SIMD = %.3f of %.0f | shader clock %.3f GHz (%.3f of 2.4)\n",
                   name, wgs_per_cu, ms, tf, tf / peak_tf, peak_tf, per_clk_simd, per_clk_simd / ideal_per_clk_simd, ideal_per_clk_simd, clock_ghz, clock_ghz / 2.4);
        }
        hipFree(stamps);
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs; peak = flop/clk/SIMD x 4 SIMDs x %d CUs x 2.4 GHz\n", p.name, cus, cus);
    run<0>(cus, "v_mfma_f32_16x16x4_f32", 2048.0, 64.0, 157.3);
    run<1>(cus, "v_mfma_f32_16x16x32_f16", 16384.0, 1024.0, 2516.6);
    return 0;
}
