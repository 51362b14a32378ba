// Device helpers shared by the GEMM and row kernels.  Anything that takes part in a floating-point reduction lives
// here exactly once: the fused (small-batch) and unfused paths must produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

#include <hip/hip_ext.h>

namespace aprilx {

// host side: every GEMM kernel goes out through APRIL_LAUNCH so that a pending pair of profiling events (gemm_profile_next_launch,
// kernels.h) rides in the kernel's own dispatch packet
struct ProfileEvents { hipEvent_t a = nullptr, b = nullptr; };
ProfileEvents &tl_profile_events();
#define APRIL_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                         \
    do {                                                                                                                           \
        aprilx::ProfileEvents &pe_ = aprilx::tl_profile_events();                                                                  \
        if (pe_.a) { hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, pe_.a, pe_.b, 0, __VA_ARGS__); pe_.a = pe_.b = nullptr; } \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                    \
    } while (0)

using f32x4 = __attribute__((ext_vector_type(4))) float;

// fixed-order block sum over 256 threads: xor-shuffle tree inside each wave, then the 4 wave sums in order
__device__ __forceinline__ float block_sum_256(float v, float *scratch4)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch4[wave] = v;
    __syncthreads();
    return ((scratch4[0] + scratch4[1]) + scratch4[2]) + scratch4[3];
}

// Finish the balanced slab tree over `parts` (1, 2, 4 or 8) partial planes ws[p][m_stride][N] at element (m, n).
// All loads are issued before the first add.
__device__ __forceinline__ float tree_sum(const float *ws, int parts, int m_stride, int N, int m, int n)
{
    float v[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) v[z] = z < parts ? ws[((size_t)z * m_stride + m) * N + n] : 0.0f;
    if (parts == 1) return v[0];
    if (parts == 2) return v[0] + v[1];
    if (parts == 4) return (v[0] + v[1]) + (v[2] + v[3]);
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

__device__ __forceinline__ f32x4 tree_sum4(const float *ws, int parts, int m_stride, int N, int m, int n)
{
    f32x4 v[8];
#pragma unroll
    for (int z = 0; z < 8; ++z)
        v[z] = z < parts ? *reinterpret_cast<const f32x4 *>(ws + ((size_t)z * m_stride + m) * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (parts == 1) return v[0];
    if (parts == 2) return v[0] + v[1];
    if (parts == 4) return (v[0] + v[1]) + (v[2] + v[3]);
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// BasicNorm scale of one row, (mean(y^2) + eps)^-1/2, from the row's sum-of-squares partials (one per SSQ_COLS columns,
// written by EPI_RESID_SSQ / ROW_RESID_SSQ): partials added in column order, v_rsq_f32 (1 ulp).  EVERY consumer of a
// normalised row calls this one function, so they all see the same scale.
__device__ __forceinline__ float row_scale(const RowScale &rs, int row)
{
    // partials are added in column order; the loads are issued eight at a time (a loop of dependent single loads costs a
    // memory round trip per partial: measured +16 us on the gate GEMM); padding terms are +0.0f, which leaves a
    // non-negative sum unchanged
    const float *p = rs.ssq + (size_t)row * rs.groups;
    float t = 0.0f;
    for (int j0 = 0; j0 < rs.groups; j0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = j0 + k < rs.groups ? p[j0 + k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
    }
    return __builtin_amdgcn_rsqf(t * rs.inv_n + rs.eps);
}

// sum of squares of a 4-column quad, then over the 8 consecutive lanes that own one SSQ_COLS-wide granule of a row
// (lane groups are 8-aligned; every lane of the wave must call this).  Order: ((q0+q1)+(q2+q3))+((q4+q5)+(q6+q7)) with
// q = (v0^2 + v1^2) + (v2^2 + v3^2) -- the same in the GEMM epilogue and in the row kernel.
__device__ __forceinline__ float granule_ssq(const f32x4 &y)
{
    float q = (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
    q += __shfl_xor(q, 1);
    q += __shfl_xor(q, 2);
    q += __shfl_xor(q, 4);
    return q;
}

// gates clock (GemmArgs::stamp): called by every thread of a workgroup at kernel entry / exit; total = workgroups of the launch.
// Slot layout (144 words): [0] start, [1] classes done, [2] sum of durations (10 ns ticks), [3] launches, [8 .. 71] 64 end stamps,
// [72 .. 135] 64 arrival counters (one per class = workgroup number mod 64).
// Contention is what a clock must not add: 512 co-resident workgroups hitting ONE L2 word with an atomic at the same moment are
// served one after the other (~12 ns each = 6 us), and a wave's first loads retire behind its own atomic -- the first version of this
// clock (every workgroup atomicMin on the start word, a returning atomicMax on one end word) read 43.7 us where rocprofv3 reports 36.3.
// So: the start is workgroup 0's alone (the grid starts first to last within 0.3 .. 0.7 us), the end stamps are spread over 64 words,
// and so are the arrivals: a workgroup counts itself in on its class's word, the last of a class counts the class in on ONE word (64
// arrivals per launch instead of one per workgroup: with a single ticket word the 1024 .. 1536 workgroups of a four- to six-problem
// launch read 118 .. 123 us where rocprofv3 reports 82.5).
__device__ __forceinline__ void stamp_begin(unsigned long long *slot, bool first_wg)
{
    if (slot && first_wg && threadIdx.x == 0) __hip_atomic_store(slot, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stamp_end(unsigned long long *slot, unsigned total, unsigned wg_linear)
{
    if (!slot) return;
    // (every wave has ISSUED its last store -- not "the stores have been acknowledged": the command processor's end-of-kernel stamp, which
    // rocprofv3 reports, does not wait for them either; with the wait, 512 co-resident workgroups ending together read 39.0 us where
    // rocprofv3 says 34.8, their 6 MB of cell state and u rows draining)
    __syncthreads();
    if (threadIdx.x == 0) {
        // (no fences: a release fence writes back the XCD's L2 -- microseconds per workgroup.  The returned old maximum makes the wave
        // wait until its own end stamp has landed before it counts itself in, so whoever closes a class finds the class's end stamps in
        // place, and whoever closes the launch every class's.)
        const unsigned cls = wg_linear & (STAMP_NENDS - 1);
        const unsigned long long old = atomicMax(slot + STAMP_ENDS + cls, (unsigned long long)wall_clock64());
        asm volatile("" :: "v"(old));
        const unsigned in_class = (total + STAMP_NENDS - 1 - cls) / STAMP_NENDS;             // workgroups 0 .. total - 1 with this residue
        if (atomicAdd(slot + STAMP_SUBS + cls, 1ull) == (unsigned long long)in_class - 1) {
            atomicExch(slot + STAMP_SUBS + cls, 0ull);
            const unsigned classes = total < (unsigned)STAMP_NENDS ? total : (unsigned)STAMP_NENDS;
            if (atomicAdd(slot + 1, 1ull) == (unsigned long long)classes - 1) {            // the last class of the launch closes the sample
                unsigned long long t1 = 0;
                for (int i = 0; i < STAMP_NENDS; ++i) { const unsigned long long e = atomicExch(slot + STAMP_ENDS + i, 0ull); t1 = e > t1 ? e : t1; }
                const unsigned long long t0 = atomicExch(slot, 0ull);
                atomicExch(slot + 1, 0ull);
                if (t0 && t1 > t0) { atomicAdd(slot + 2, t1 - t0); atomicAdd(slot + 3, 1ull); }
            }
        }
    }
}

// First-round start skew of the kernels that run two (or three) workgroups per CU over several rounds (GemmArgs::skew, x 4096 cycles; 0 = off,
// the default: a MEASUREMENT FORM).  The workgroups a CU gets at the start of a launch run in lock step -- all in their prologues, all in the
// K loop, all in their epilogues with the matrix pipe idle -- and their successors inherit the phase, because the slots free up together.  A
// phase offset, once established, is kept by the same mechanism: the odd threadgroup slot of every CU (HW_ID.TG_ID bit 0; tools/cu_pair_probe:
// the two workgroups of a first-round pair always differ in it, and they are 256 apart in dispatch order) sleeps once, and only workgroups of
// the first round (linear id < first_round) ever do.  Round 1's form of this slept in EVERY second block of 256 workgroup ids, every round, and
// lost.  This one is neutral (profiles/r06_first_round_skew.txt: gates at 4608 rows 323 us at any delay of 3 .. 12 x 4096 cycles): a
// workgroup alone on a CU drives the matrix pipe at ~0.55 of its rate (its own barriers and fragment waits), so what de-phasing wins in
// overlapped epilogues it loses in K loops that run alone.  Speed only, never correctness.
__device__ __forceinline__ void first_round_skew(int skew, unsigned wg_linear, unsigned first_round)
{
    if (skew <= 0 || wg_linear >= first_round) return;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((hw >> 16) & 1) for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(64);
}

// sigma and tanh on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each); absolute error ~1e-7, far inside
// the 1e-4 per-call parity tolerance, and ~10x fewer VALU instructions than libm's expf/tanhf
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * fast_sigmoid(2.0f * x) - 1.0f; }

}  // namespace aprilx
