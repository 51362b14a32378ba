// Measurement aid: what a launch of N workgroups costs when every workgroup only sleeps for a fixed time -- the dispatcher's
// share of a one-round launch.  Workgroups of 256 / 512 / 1024 threads, 240 VGPRs per lane (forced), LDS as given.
//   build: hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/dispatch_probe
//   usage: tools/dispatch_probe [sleep_cycles=0] [lds_kb=70]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NT>
__global__ __launch_bounds__(NT) void probe(int sleep64, float *out)
{
    extern __shared__ float lds[];
    if constexpr (NT <= 512) asm volatile("v_mov_b32 v230, 0" ::: "v230");          // forces a 231+ VGPR allocation: two 4-wave workgroups per CU at most
    else asm volatile("v_mov_b32 v120, 0" ::: "v120");
    for (int i = 0; i < sleep64; ++i) __builtin_amdgcn_s_sleep(64);
    if (out && threadIdx.x == 0 && blockIdx.x == 0xffffff) out[0] = lds[0];
}

template <int NT>
static float run(int wgs, int sleep64, size_t lds, hipStream_t s, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<NT>, dim3(wgs), dim3(NT), lds, s, sleep64, nullptr);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe<NT>, dim3(wgs), dim3(NT), lds, s, sleep64, nullptr);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main(int argc, char **argv)
{
    const int sleep64 = argc > 1 ? atoi(argv[1]) : 0;
    const size_t lds = (size_t)(argc > 2 ? atoi(argv[2]) : 70) * 1024;
    hipStream_t s; hipStreamCreate(&s);
    hipFuncSetAttribute((const void *)probe<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)probe<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)probe<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("sleep %d x s_sleep(64), LDS %zu KB per 256 threads; us per launch (back-to-back launches on one stream)\n", sleep64, lds >> 10);
    printf("%8s %10s %10s %10s\n", "waves", "256 thr", "512 thr", "1024 thr");
    for (int waves : {256, 512, 1024, 2048, 3072, 4096, 8192}) {
        const float a = run<256>(waves / 4, sleep64, lds, s, 300);
        const float b = run<512>(waves / 8, sleep64, lds * 2, s, 300);
        const float c = waves >= 16 ? run<1024>(waves / 16, sleep64, 0, s, 300) : 0;
        printf("%8d %10.2f %10.2f %10.2f\n", waves, a, b, c);
    }
    return 0;
}
