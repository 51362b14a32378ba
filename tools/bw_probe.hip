// Read bandwidth of a buffer that is streamed over and over (working sets from L2-sized to HBM-sized): what a weight-streaming
// kernel can expect from the Infinity Cache vs HBM.  build: hipcc --offload-arch=gfx950 -O2 tools/bw_probe.hip -o tools/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f4 = __attribute__((ext_vector_type(4))) float;
template <int NT> __global__ __launch_bounds__(256) void rd(const f4 *__restrict__ p, size_t n16, float *out)
{
    // every thread keeps 8 x 16 B in flight; consecutive lanes read consecutive 16 B
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; i < n16; i += stride) {
        f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const f4 *q = p + i + k * 256; v[k] = (i + k * 256 < n16) ? (NT ? __builtin_nontemporal_load(q) : *q) : f4{0, 0, 0, 0}; }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
int main()
{
    const size_t sizes_mb[] = {16, 24, 48, 96, 120, 200, 320, 1024};
    float *out; hipMalloc(&out, 4);
    f4 *buf; hipMalloc(&buf, (size_t)1024 << 20); hipMemset(buf, 0, (size_t)1024 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nt = 0; nt < 2; ++nt)
        for (size_t mb : sizes_mb) for (int grid : {2048, 8192}) {
            const size_t n16 = (mb << 20) / 16;
            const int reps = 20;
            for (int w = 0; w < 3; ++w) { if (nt) rd<1><<<grid, 256>>>(buf, n16, out); else rd<0><<<grid, 256>>>(buf, n16, out); }
            hipEventRecord(a);
            for (int r = 0; r < reps; ++r) { if (nt) rd<1><<<grid, 256>>>(buf, n16, out); else rd<0><<<grid, 256>>>(buf, n16, out); }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("nt=%d %5zu MB grid %5d: %7.2f us per pass, %.2f TB/s\n", nt, mb, grid, ms * 1e3 / reps, (double)(mb << 20) / (ms * 1e-3 / reps) / 1e12);
        }
    return 0;
}
