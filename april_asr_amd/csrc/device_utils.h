// Device helpers shared by the GEMM and row kernels.  Anything that takes part in a floating-point reduction lives
// here exactly once: the fused (small-batch) and unfused paths must produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace aprilx {

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

// sigma and tanh on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each); absolute error ~1e-7, far inside
// the 1e-4 per-call parity tolerance, and ~10x fewer VALU instructions than libm's expf/tanhf
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * fast_sigmoid(2.0f * x) - 1.0f; }

}  // namespace aprilx
