// Go / no-go probe for recurrent weights ON CHIP in the offline wavefront (reference call site
// src/april_session.c:441-454: one session, a minute of audio in one call).  Today every time step of the layer-major wavefront
// re-streams the recurrent weights of the 12 active layers (gates h-half 8 MB + projection 2 MB each = 120 MB) and pays the launch
// boundaries: ~26 us per wavefront step.  The alternative: ONE persistent launch, every CU keeps its 1 / 256 of every layer's W_hh
// (16 gate columns x 512 k) and W_hr (2 columns x 1024 k) in registers -- 480 floats per thread at 256 threads per CU --, and the time
// steps are separated by hand-rolled grid barriers:
//     phase A  gates h-half (GEMV from registers) + cell for 4 hidden units per layer      -> publish 12 x 4 floats, barrier, gather u (12 x 1024)
//     phase B  projection (GEMV from registers) for 2 outputs per layer                     -> publish 12 x 2 floats, barrier, gather h (12 x 512)
// This probe runs exactly that data movement and synchronisation (the arithmetic is the right amount of FMAs on register-resident
// weights, not a checked LSTM) for 1500 steps and reports us per step, in four cumulative variants.  Go: <= 12 us per step.
//   exchange: 8-byte agent-scope atomics on both sides (MI355X_MICROARCH.md "valid forms": coherent across XCDs without fences);
//   barrier:  monotonic counters, relaxed agent-scope polls + s_sleep; flat (one counter) or two-level (per-XCC counter, the XCC's last
//             arriver bumps the top counter and releases its XCC through a generation word).  Every spin is bounded (a stuck barrier
//             sets a flag and the kernel runs out).
// build: hipcc --offload-arch=gfx950 -O3 tools/persistent_rnn_probe.hip -o tools/persistent_rnn_probe     usage: tools/persistent_rnn_probe [steps=1500]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int L = 12, D = 512, H = 1024, NCU = 256, NT = 256;
constexpr int UPC = H / NCU;            // hidden units per CU and layer (4): 16 gate columns
constexpr int HPC = D / NCU;            // projection outputs per CU and layer (2)
constexpr int WA = L * 16 * D / NT;     // 384 register-resident W_hh floats per thread
constexpr int WB = L * HPC * H / NT;    // 96 register-resident W_hr floats per thread

struct Sync { unsigned long long top; unsigned long long xcnt[8]; unsigned long long xgen[8]; unsigned long long xpop[8]; unsigned long long stuck; unsigned long long pad[7]; };

__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid barrier number `gen` (1, 2, ...); thread 0 of every workgroup; TWO = two-level form
template <int TWO> __device__ __forceinline__ void grid_barrier(Sync *s, unsigned long long gen, int xcc, unsigned nwg)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this workgroup's published granules have left
        long spins = 0;
        if (ld(&s->stuck)) { /* a barrier got stuck earlier: run out without waiting */ }
        else if (TWO) {
            const unsigned long long pop = ld(&s->xpop[xcc]);
            const unsigned long long mine = __hip_atomic_fetch_add(&s->xcnt[xcc], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (mine == pop * (gen - 1)) {                                // (one flat barrier came first) the XCC's last arriver: up to the top, then release the XCC
                __hip_atomic_fetch_add(&s->top, pop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (ld(&s->top) < (unsigned long long)nwg * gen) { __builtin_amdgcn_s_sleep(1); if (++spins > 400000) { st(&s->stuck, gen); break; } }
                st(&s->xgen[xcc], gen);
            } else {
                while (ld(&s->xgen[xcc]) < gen) { __builtin_amdgcn_s_sleep(1); if (++spins > 400000) { st(&s->stuck, gen); break; } }
            }
        } else {
            __hip_atomic_fetch_add(&s->top, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld(&s->top) < (unsigned long long)nwg * gen) { __builtin_amdgcn_s_sleep(1); if (++spins > 400000) { st(&s->stuck, gen); break; } }
        }
    }
    __syncthreads();
}

// VARIANT 0: barriers only; 1: + publish / gather through 8-byte agent atomics; 2: + the FMAs on register-resident weights
template <int VARIANT, int TWO>
__global__ __launch_bounds__(NT, 1) void rnn_probe(Sync *s, unsigned long long *ubuf, unsigned long long *hbuf, float *sink, int steps)
{
    __shared__ float u_l[L * H];       // 48 KB: the gathered u vectors of the 12 layers
    __shared__ float h_l[L * D];       // 24 KB
    const int t = threadIdx.x, cu = blockIdx.x;
    const unsigned nwg = gridDim.x;
    int xcc = 0;
    if (TWO) {                          // census: how many workgroups sit on my XCC (placement is not a contract: count, do not assume)
        xcc = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7);      // HW_REG_XCC_ID, bits 3:0
        if (t == 0) __hip_atomic_fetch_add(&s->xpop[xcc], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // register-resident weights (values irrelevant; kept live by the FMAs below)
    float wa[WA], wb[WB];
#pragma unroll
    for (int i = 0; i < WA; ++i) wa[i] = 1e-3f * (float)((t * 31 + i * 7) & 255);
#pragma unroll
    for (int i = 0; i < WB; ++i) wb[i] = 1e-3f * (float)((t * 17 + i * 5) & 255);
    for (int i = t; i < L * H; i += NT) u_l[i] = 0.01f;
    for (int i = t; i < L * D; i += NT) h_l[i] = 0.01f;
    // a flat barrier first, so that the census is complete before the two-level form reads it
    unsigned long long gen = 0;
    ++gen; grid_barrier<0>(s, gen, xcc, nwg);
    if (TWO && t == 0) { /* the flat barrier used `top`: the two-level form continues on it, counting workgroups in both forms */ }
    float acc = 0.f;
    for (int stp = 0; stp < steps; ++stp) {
        // ---- phase A: gates h-half for my 16 columns of every layer: thread = (column t / 16, k segment t % 16 of 32 k), 32 FMAs per layer
        float ga[L];
        if (VARIANT >= 2) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                float a = 0.f;
                const float *x = h_l + l * D + (t & 15) * 32;
#pragma unroll
                for (int k = 0; k < 32; ++k) a = __builtin_fmaf(wa[l * 32 + k], x[k], a);
                a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);      // the 16 k segments of a column
                ga[l] = a;
            }
        } else {
#pragma unroll
            for (int l = 0; l < L; ++l) ga[l] = (float)stp;
        }
        if (VARIANT >= 1) {
            // publish u: 4 units per layer = 2 granules per layer and CU (lanes 0 / 64 of the column groups stand in for the cell epilogue)
            if ((t & 127) == 0) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    unsigned long long g; float2 v = make_float2(ga[l], ga[l] + 1.f); __builtin_memcpy(&g, &v, 8);
                    st(&ubuf[(size_t)l * (H / 2) + cu * (UPC / 2) + (t >> 7)], g);
                }
            }
        }
        ++gen; grid_barrier<TWO>(s, gen, xcc, nwg);
        if (VARIANT >= 1) {
            // gather u of all layers: L * H / 2 = 6144 granules per CU, 24 per thread
#pragma unroll 8
            for (int i = t; i < L * H / 2; i += NT) { const unsigned long long g = ld(&ubuf[i]); float2 v; __builtin_memcpy(&v, &g, 8); u_l[2 * i] = v.x; u_l[2 * i + 1] = v.y; }
            __syncthreads();
        }
        // ---- phase B: projection for my 2 outputs of every layer: thread = (output t / 128, k segment t % 128 of 8 k)
        float pb[L];
        if (VARIANT >= 2) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                float a = 0.f;
                const float *x = u_l + l * H + (t & 127) * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) a = __builtin_fmaf(wb[l * 8 + k], x[k], a);
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o);
                pb[l] = a;
            }
        } else {
#pragma unroll
            for (int l = 0; l < L; ++l) pb[l] = (float)stp;
        }
        if (VARIANT >= 1) {
            if ((t & 63) == 0 && (t >> 6) < 2) {           // (two waves per output: wave 0 / 1 publish; the cross-wave add is omitted)
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    unsigned long long g; float2 v = make_float2(pb[l], pb[l] * 0.5f); __builtin_memcpy(&g, &v, 8);
                    if ((t >> 6) == 0) st(&hbuf[(size_t)l * (D / 2) + cu * (HPC / 2)], g);
                }
            }
        }
        ++gen; grid_barrier<TWO>(s, gen, xcc, nwg);
        if (VARIANT >= 1) {
#pragma unroll 4
            for (int i = t; i < L * D / 2; i += NT) { const unsigned long long g = ld(&hbuf[i]); float2 v; __builtin_memcpy(&v, &g, 8); h_l[2 * i] = v.x * 1e-3f + 0.01f; h_l[2 * i + 1] = v.y * 1e-3f + 0.01f; }
            __syncthreads();
        }
        acc += ga[stp % L] + pb[(stp + 1) % L];
    }
    if (acc == 123.456f) sink[cu * NT + t] = acc;
}

template <int VARIANT, int TWO> static void run(const char *name, int steps)
{
    Sync *s; unsigned long long *ub, *hb; float *sink;
    CK(hipMalloc((void **)&s, sizeof(Sync))); CK(hipMemset(s, 0, sizeof(Sync)));
    CK(hipMalloc((void **)&ub, (size_t)L * H / 2 * 8)); CK(hipMemset(ub, 0, (size_t)L * H / 2 * 8));
    CK(hipMalloc((void **)&hb, (size_t)L * D / 2 * 8)); CK(hipMemset(hb, 0, (size_t)L * D / 2 * 8));
    CK(hipMalloc((void **)&sink, (size_t)NCU * NT * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((rnn_probe<VARIANT, TWO>), dim3(NCU), dim3(NT), 0, 0, s, ub, hb, sink, steps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Sync h; CK(hipMemcpy(&h, s, sizeof h, hipMemcpyDeviceToHost));
    printf("%-72s %8.2f us per step (%d steps, %.2f ms)%s  [XCC populations:", name, ms * 1e3 / steps, steps, ms, h.stuck ? "  STUCK BARRIER" : "");
    for (int i = 0; i < 8; ++i) printf(" %llu", h.xpop[i]);
    printf("]\n");
    CK(hipFree(s)); CK(hipFree(ub)); CK(hipFree(hb)); CK(hipFree(sink));
}

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 1500;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs; %d workgroups x %d threads, %d + %d register-resident weights per thread (%d KB per CU), LDS %d KB\n", p.name, p.multiProcessorCount, NCU, NT, WA, WB,
           (WA + WB) * NT * 4 / 1024, (L * H + L * D) * 4 / 1024);
    run<0, 0>("two flat barriers per step, nothing else", steps);
    run<0, 1>("two two-level (per-XCC) barriers per step, nothing else", steps);
    run<1, 0>("flat barriers + publish / gather (u: 48 KB, h: 24 KB per CU and step)", steps);
    run<1, 1>("two-level barriers + publish / gather", steps);
    run<2, 1>("two-level barriers + publish / gather + GEMVs on register-resident weights", steps);
    run<2, 0>("flat barriers + publish / gather + GEMVs on register-resident weights", steps);
    printf("go / no-go: <= 12 us per wavefront step (the launch chain of the layer-major wavefront: ~26 us per step)\n");
    return 0;
}
