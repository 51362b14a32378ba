// Measurement aid (round 5): what does one vector-memory instruction cost a wave that streams v_mfma_f32_16x16x4_f32, as a function of
//   - how many of them go with a block of 16 MFMAs (the operand traffic of a 32 x 32 wave tile is 4 per block: 2 KB of A, 2 KB of B),
//   - where they are placed (all behind the block, or one after every 16 / X-th MFMA),
//   - what they are (global_load_dwordx4 to registers with a 64-bit address, with an SGPR base + 32-bit offset, LDS DMA),
//   - how many waves share a SIMD (256- or 512-thread workgroups, one per CU; 16 waves = two 512-thread workgroups).
// Every wave reads 1 KB per instruction from a 4 MB buffer (L2 / MALL resident); the loaded registers feed the next block's MFMAs, so
// nothing is dead.  Prints cycles per block (s_memtime, wave 0 of workgroup 0 and the chip-wide kernel time).
// build: hipcc --offload-arch=gfx950 -O2 tools/vmem_mfma_probe.hip -o tools/vmem_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <class T> __device__ __forceinline__ T gload(const void *p) { return *(const __attribute__((address_space(1))) T *)(p); }

// KIND 0: register loads, 64-bit per-lane address; 1: register loads, uniform base + 32-bit lane offset; 2: LDS DMA (uniform base + lane offset)
// X loads per block of 16 MFMAs; SPREAD 1: one load after every (16 / X)-th MFMA, 0: all loads behind the block
template <int KIND, int X, int SPREAD>
__global__ void probe(const float *buf, float *out, unsigned long long *t, int iters, unsigned mask)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4], a = {1.f + lane, 2.f, 3.f, 4.f}, b[8];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) b[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    unsigned pos = ((blockIdx.x * 16 + wave) * 8191u) & mask;          // this wave's position in the buffer, in KB
    const char *base = reinterpret_cast<const char *>(buf);
    const unsigned loff = lane * 16;
    char *mylds = lds + wave * 8192;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q >> 2], b[q % (X > 0 ? X : 1)][q >> 2], acc[q & 3], 0, 0, 0);
            if (X > 0 && SPREAD && (q + 1) % (16 / X) == 0) {
                const int x = q / (16 / X);
                __builtin_amdgcn_sched_barrier(0);
                const char *p = base + (size_t)pos * 1024;
                if (KIND == 0) b[x] = gload<f32x4>(p + loff);
                else if (KIND == 1) b[x] = gload<f32x4>(base + ((size_t)pos * 1024 + loff));
                else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + loff), (__attribute__((address_space(3))) void *)(mylds + x * 1024), 16, 0, 0);
                pos = (pos + 257u) & mask;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (X > 0 && !SPREAD) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < X; ++x) {
                const char *p = base + (size_t)pos * 1024;
                if (KIND == 0) b[x] = gload<f32x4>(p + loff);
                else if (KIND == 1) b[x] = gload<f32x4>(base + ((size_t)pos * 1024 + loff));
                else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + loff), (__attribute__((address_space(3))) void *)(mylds + x * 1024), 16, 0, 0);
                pos = (pos + 257u) & mask;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 2 && X > 0) {      // keep at most two blocks of DMA in flight; consume something from LDS so the buffer is live
            __builtin_amdgcn_s_waitcnt(0x0F70 | (X & 15));
            b[0] = *reinterpret_cast<const f32x4 *>(mylds + lane * 16);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}

template <int KIND, int X, int SPREAD> void run(const float *buf, float *out, unsigned long long *t, int threads, int wgs)
{
    const int iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<KIND, X, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = (size_t)(threads / 64) * 8192;
    hipLaunchKernelGGL((probe<KIND, X, SPREAD>), dim3(wgs), dim3(threads), lds, 0, buf, out, t, iters, 4095u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<KIND, X, SPREAD>), dim3(wgs), dim3(threads), lds, 0, buf, out, t, iters, 4095u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    const double waves_per_simd = (double)threads / 256.0 * wgs / 256.0;
    // MFMA-bound time of the kernel: iters blocks x 16 MFMAs x 32 cycles x waves per SIMD at 2.4 GHz
    const double ideal_us = iters * 16.0 * 32.0 * waves_per_simd / 2400.0;
    printf("kind %d  loads/block %d  %s  %4d threads x %3d wgs (%.0f waves/SIMD): wave 0: %7.1f cycles per block (%.1f per SIMD-block) | kernel %.1f us = %.2f x the MFMA time, %.1f B/clk/CU\n",
           KIND, X, SPREAD ? "spread" : "lumped", threads, wgs, waves_per_simd, (double)h / iters, (double)h / iters / waves_per_simd, ms * 1e3, ms * 1e3 / ideal_us,
           X * 1024.0 * iters * (threads / 64) * (wgs / 256.0) / (ms * 1e-3 * 2.4e9));
}

template <int KIND, int X> void both(const float *buf, float *out, unsigned long long *t)
{
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int threads = cfg == 0 ? 256 : 512, wgs = cfg == 2 ? 512 : 256;
        run<KIND, X, 0>(buf, out, t, threads, wgs);
        if (X > 0) run<KIND, X, 1>(buf, out, t, threads, wgs);
    }
}

int main()
{
    float *buf, *out; unsigned long long *t;
    hipMalloc(&buf, 4096 * 1024 + 4096); hipMalloc(&out, 4096); hipMalloc(&t, 8);
    hipMemset(buf, 0, 4096 * 1024 + 4096);
    both<0, 0>(buf, out, t);
    both<0, 2>(buf, out, t); both<0, 4>(buf, out, t); both<0, 8>(buf, out, t);
    both<1, 2>(buf, out, t); both<1, 4>(buf, out, t); both<1, 8>(buf, out, t);
    both<2, 2>(buf, out, t); both<2, 4>(buf, out, t); both<2, 8>(buf, out, t);
    return 0;
}
