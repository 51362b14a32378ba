// Measurement aid: which workgroups of a launch share a CU, and what HW_ID says about them.  The first-round start skew of the
// two-workgroups-per-CU GEMM kernels (device_utils.h first_round_skew) delays ONE of the two workgroups that start together on a
// CU; it picks it by a bit of HW_ID.  This probe launches 4-wave workgroups that can only be co-resident in pairs (70 KB of LDS
// each), lets every one of them sleep ~8 us, and records (linear id, HW_ID, XCC_ID, start, end).  It prints, for the first
// 2 x CUs workgroups: how many CUs got exactly two of them, which HW_ID bit fields differ inside a pair, and the linear-id
// distance of the pair; then the same for the later rounds (does the parity of the field survive a replacement?).
// With lds_bytes given it also answers "how many of these workgroups does a CU hold at once": the histogram of workgroups per CU among
// those that started before the first one ended (the LDS allocation granule decides whether 3 x 53760 bytes fit the 160 KB).
//   build: hipcc --offload-arch=gfx950 -O2 tools/cu_pair_probe.hip -o tools/cu_pair_probe      usage: tools/cu_pair_probe [wgs=2048] [lds_bytes=71680]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned long long *rec, int sleep64)
{
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    for (int i = 0; i < sleep64; ++i) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 4 + 0] = ((unsigned long long)xcc << 32) | hw;
        rec[blockIdx.x * 4 + 1] = t0;
        rec[blockIdx.x * 4 + 2] = wall_clock64();
    }
    if (blockIdx.x == 0xffffff) rec[0] = (unsigned long long)lds[threadIdx.x];
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 2048;
    const size_t lds_bytes = argc > 2 ? (size_t)atol(argv[2]) : 70 * 1024;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned long long *d; hipMalloc((void **)&d, (size_t)wgs * 32);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) { hipMemset(d, 0, (size_t)wgs * 32); hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), lds_bytes, 0, d, 5); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h((size_t)wgs * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    auto cu_key = [&](int i) { const unsigned hw = (unsigned)h[(size_t)i * 4], xcc = (unsigned)(h[(size_t)i * 4] >> 32); return ((unsigned long long)(xcc & 0xf) << 16) | ((hw >> 8) & 0xff); };   // CU_ID[11:8] SH_ID[12] SE_ID[15:13]
    printf("%d CUs, %d workgroups; HW_ID fields: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] tg[19:16] vm[23:20] queue[26:24] state[29:27] me[31:30]\n", cus, wgs);
    for (int i = 0; i < 8; ++i) printf("  wg %4d: hw %08x xcc %x  start %llu\n", i, (unsigned)h[(size_t)i * 4], (unsigned)(h[(size_t)i * 4] >> 32), h[(size_t)i * 4 + 1] - h[1]);
    {   // co-residency: workgroups whose start precedes the earliest end of the launch, per CU
        unsigned long long first_end = ~0ull;
        for (int i = 0; i < wgs; ++i) if (h[(size_t)i * 4 + 2] && h[(size_t)i * 4 + 2] < first_end) first_end = h[(size_t)i * 4 + 2];
        std::map<unsigned long long, int> per_cu;
        for (int i = 0; i < wgs; ++i) if (h[(size_t)i * 4 + 1] < first_end) ++per_cu[cu_key(i)];
        std::map<int, int> hist;
        for (auto &kv : per_cu) ++hist[kv.second];
        printf("LDS %zu bytes per workgroup (256 threads): workgroups resident together per CU:", lds_bytes);
        for (auto &kv : hist) printf("  %d on %d CUs", kv.first, kv.second);
        printf("\n");
        if (argc > 2) return 0;
    }
    for (int round = 0; round * 2 * cus < wgs && round < 4; ++round) {
        std::map<unsigned long long, std::vector<int>> by_cu;
        for (int i = round * 2 * cus; i < (round + 1) * 2 * cus && i < wgs; ++i) by_cu[cu_key(i)].push_back(i);
        int pairs = 0, others = 0; unsigned diff_or = 0, diff_and = ~0u; long dist_sum = 0; int tg_par_differs = 0, wave_par_differs = 0;
        std::map<int, int> dist_hist;
        for (auto &kv : by_cu) {
            if (kv.second.size() != 2) { ++others; continue; }
            ++pairs;
            const unsigned a = (unsigned)h[(size_t)kv.second[0] * 4], b = (unsigned)h[(size_t)kv.second[1] * 4];
            diff_or |= a ^ b; diff_and &= a ^ b;
            dist_sum += kv.second[1] - kv.second[0]; ++dist_hist[kv.second[1] - kv.second[0]];
            tg_par_differs += (((a ^ b) >> 16) & 1); wave_par_differs += ((a ^ b) & 1);
        }
        printf("round %d (workgroups %d..%d): %zu CU keys, %d hold exactly two, %d another count; inside a pair HW_ID differs in bits %08x (always in %08x); tg parity differs in %d pairs, wave parity in %d; mean id distance %.1f\n",
               round, round * 2 * cus, (round + 1) * 2 * cus - 1, by_cu.size(), pairs, others, diff_or, diff_and, tg_par_differs, wave_par_differs, pairs ? (double)dist_sum / pairs : 0.0);
        int shown = 0;
        for (auto &kv : dist_hist) if (shown++ < 6) printf("    id distance %d: %d pairs\n", kv.first, kv.second);
    }
    return 0;
}
