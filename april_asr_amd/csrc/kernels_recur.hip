// The two GEMMs of a recurrent time step of a LONG FEED at a handful of rows (one session handing over a file: M = 1), as
// weight streams.  Reference path: the per-chunk encoder call of src/april_session.c:431-476 run on a whole recording; here the
// layer-major schedule (engine.cc, Engine::run_lm_wavefront) leaves two launches per time step whose cost is the layers'
// recurrent weights read once -- 96 MB of gate weights (h half) and 24 MB of projection weights at aprilv0 size, twelve layers
// per launch:
//   gates  EPI_LSTM, wave_mask 0b1100, p_add:  ((P + c2) + c3) + bias, LSTM cell           (P = the input half, EPI_XPART)
//   whr    EPI_HR, all of K in the workgroup:   h' = u x Whr, state row, x + h'
// The general kernels (kernels_gemm.hip) run them as 16 x 16 / 16 x 32 tiles with half of the waves idle and 6 KB of weights
// in flight per wave: 20.9 + 9.3 us per time step at twelve layers, against 14.9 + 3.6 us for the same bytes at the streaming
// rate of this GPU (tools/bw_probe: 6.8 TB/s from 48 MB up; the Infinity Cache adds nothing).  Here every wave owns whole
// chunks of one 16-column tile and has its whole k range in flight before its first MFMA.
//
// Arithmetic: the chains are those of gemm_body -- a chunk is ONE in-order chain of v_mfma_f32_16x16x4_f32 over its k blocks
// (k step j of a block = element j of the lane's operand quads), chunks meet as ((c0 + c1) + c2) + c3, slabs in the balanced
// pairwise tree -- so the results are bit-identical to the general kernels and to the streaming schedule
// (tests/test_gpu_layer_major.py compares every logit).
#include "kernels.h"
#include "device_utils.h"
#include <cstdio>
#include <cstdlib>

namespace aprilx {

namespace {

// The pointer members of GemmArgs are generic to the compiler (flat_load: counted on BOTH memory counters, so every LDS access would
// wait for the weight stream); everything here is global memory and is addressed as such.
template <class T> __device__ __forceinline__ T gload(const void *p) { return *(const __attribute__((address_space(1))) T *)(p); }
template <class T> __device__ __forceinline__ T gload_nt(const void *p) { return __builtin_nontemporal_load((const __attribute__((address_space(1))) T *)(p)); }
template <class T> __device__ __forceinline__ void gstore(void *p, const T &v) { *(__attribute__((address_space(1))) T *)(p) = v; }

constexpr int RLD = 20;                       // floats per row of a 16 x 16 LDS plane (16-byte aligned quads, conflict-free enough)
constexpr int RPLANE = 16 * RLD;

// TB consecutive k blocks (compile time: the whole stream is straight-line code, every load counted exactly) starting at the
// wave's operand bases ap / bp (uniform; 64 B resp. 1 KB per block) plus the lane's 32-bit byte offsets ao / bo: chains of c
// blocks, each finished chain handed to done(chunk number within the wave, acc).  At most 16 blocks (2 x 16 KB per wave) are in
// flight: all of them before the first MFMA when TB <= 16.
template <int TB, class F>
__device__ __forceinline__ void stream_chains(const char *ap, uint32_t ao, const char *bp, uint32_t bo, int c, F done)
{
    constexpr int R = TB < 16 ? TB : 16;
    f32x4 a[R], b[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        a[i] = gload<f32x4>(ap + (size_t)i * 64 + ao);
        b[i] = gload_nt<f32x4>(bp + (size_t)i * 1024 + bo);
    }
    __builtin_amdgcn_sched_barrier(0);              // (left alone, the scheduler sinks each load to its MFMAs: four blocks in flight instead of R)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int in_chunk = 0, chunk = 0;
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        const int s = i % R;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].x, b[s].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].y, b[s].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].z, b[s].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].w, b[s].w, acc, 0, 0, 0);
        if (i + R < TB) {
            a[s] = gload<f32x4>(ap + (size_t)(i + R) * 64 + ao);
            b[s] = gload_nt<f32x4>(bp + (size_t)(i + R) * 1024 + bo);
        }
        if (++in_chunk == c) { done(chunk, acc); ++chunk; in_chunk = 0; acc = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
}

__device__ __forceinline__ void plane_store(float *plane, int lane, const f32x4 &acc)
{
    const int col = lane & 15, r0 = (lane >> 4) * 4;        // accumulator register r = output row 4 (lane / 16) + r, column lane % 16
#pragma unroll
    for (int r = 0; r < 4; ++r) plane[(r0 + r) * RLD + col] = acc[r];
}

// ---- gates: a workgroup = two 16-column tiles (eight hidden units) x the two chunks of the recurrent half of K
template <int TB>
__device__ __forceinline__ void recur_gates_body(const GemmArgs &g)
{
    __shared__ __attribute__((aligned(16))) float red[4 * RPLANE];
    if (g.run_flag && gload<int>(g.run_flag) != g.run_gen) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int KB = 4 * TB, c = TB;                                     // (checked on the host: K = 64 TB)
    const int ct = blockIdx.x * 2 + (wave >> 1), half = wave & 1;         // this wave: chunk 2 + half of column tile ct
    // epilogue operands of this thread's (row, unit), fetched before the stream (row -> slot -> previous cell value is a dependent pair)
    const int et = threadIdx.x >> 6, erow = (threadIdx.x >> 2) & 15, eul = threadIdx.x & 3;
    const bool e_on = threadIdx.x < 128, e_ok = e_on && erow < g.M;
    const int en = (blockIdx.x * 2 + et) * 16 + eul * 4, eunit = en >> 2;
    const int em = erow < g.M ? erow : g.M - 1;
    float *cptr = nullptr;
    float cprev = 0.0f;
    f32x4 ebias = {0.f, 0.f, 0.f, 0.f}, xin = {0.f, 0.f, 0.f, 0.f};
    if (e_on) {
        cptr = g.c_state + (size_t)gload<int>(g.slot_idx + em) * g.hidden + eunit;
        ebias = gload<f32x4>(g.bias + en);
        xin = gload<f32x4>(g.p_add + (size_t)em * g.ldp + en);
        cprev = gload<float>(cptr);
    }
    int row = lane & 15;
    if (row >= g.M) row = g.M - 1;                                         // padding rows recompute the last row; never stored
    const int slot = g.aidx1 ? gload<int>(g.aidx1 + row) : row;
    // (all operands are far below 4 GiB per array: 32-bit lane offsets)
    const char *ap = reinterpret_cast<const char *>(g.a1) + (size_t)half * c * 64;
    const uint32_t ao = (uint32_t)(((size_t)slot * g.lda1 + (lane >> 4) * 4) * sizeof(float));
    const char *bp = reinterpret_cast<const char *>(g.wp) + ((size_t)ct * KB + (size_t)(2 + half) * c) * 1024;
    float *mine = red + wave * RPLANE;
    stream_chains<TB>(ap, ao, bp, (uint32_t)lane * 16u, c, [&](int, const f32x4 &acc) { plane_store(mine, lane, acc); });
    __syncthreads();
    if (e_on) {
        const int o = erow * RLD + eul * 4;
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(red + (2 * et) * RPLANE + o), p3 = *reinterpret_cast<const f32x4 *>(red + (2 * et + 1) * RPLANE + o);
        const f32x4 gt = ((xin + p2) + p3) + ebias;
        const float c_new = fast_sigmoid(gt.y) * cprev + fast_sigmoid(gt.x) * fast_tanh(gt.z);
        const float u = fast_sigmoid(gt.w) * fast_tanh(c_new);
        if (e_ok) { gstore<float>(cptr, c_new); gstore<float>(g.out + (size_t)erow * g.ldo + eunit, u); }
    }
}

// ---- projection: a workgroup = one 16-column tile, its 4 kz chunks dealt to the waves (consecutive chunks per wave)
constexpr int WHR_MAX_PLANES = 32, WHR_MAX_GROUPS = 32;

template <int TB>
__device__ __forceinline__ void recur_whr_body(const GemmArgs &g)
{
    __shared__ __attribute__((aligned(16))) float red[WHR_MAX_PLANES * RPLANE];
    __shared__ float part[16 * (WHR_MAX_GROUPS + 1)];
    if (g.run_flag && gload<int>(g.run_flag) != g.run_gen) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    const int KB = g.K >> 4, nch = 4 * g.kz, c = KB / nch, cpw = nch / nw;      // (checked on the host: cpw * c == TB)
    const int ct = blockIdx.x;
    // epilogue operands (threads 0..63: row = t / 4, quad = t % 4): slot, residual quad, the row's sum-of-squares partials
    const int erow = (int)threadIdx.x >> 2, eq = threadIdx.x & 3;
    const bool e_on = threadIdx.x < 64, e_ok = e_on && erow < g.M;
    const int em = erow < g.M ? erow : g.M - 1;
    const int en = ct * 16 + eq * 4;
    int eslot = 0;
    f32x4 eres = {0.f, 0.f, 0.f, 0.f};
    float sp[WHR_MAX_GROUPS / 4];                                        // partials eq * 8 .. eq * 8 + 7 of the row (its four threads share them through LDS)
    const int G = g.r_scale.groups;
    if (e_on) {
        eslot = g.slot_idx ? gload<int>(g.slot_idx + em) : em;
        eres = gload<f32x4>(g.resid + (size_t)em * g.ldr + en);
#pragma unroll
        for (int k = 0; k < WHR_MAX_GROUPS / 4; ++k) { const int j = eq * (WHR_MAX_GROUPS / 4) + k; sp[k] = gload<float>(g.r_scale.ssq + (size_t)em * G + (j < G ? j : G - 1)); }
    }
    int row = lane & 15;
    if (row >= g.M) row = g.M - 1;
    const int arow = g.aidx0 ? gload<int>(g.aidx0 + row) : row;
    const int kb0 = wave * cpw * c;
    const char *ap = reinterpret_cast<const char *>(g.a0) + (size_t)kb0 * 64;
    const uint32_t ao = (uint32_t)(((size_t)arow * g.lda0 + (lane >> 4) * 4) * sizeof(float));
    const char *bp = reinterpret_cast<const char *>(g.wp) + ((size_t)ct * KB + kb0) * 1024;
    float *mine = red + (size_t)wave * cpw * RPLANE;
    stream_chains<TB>(ap, ao, bp, (uint32_t)lane * 16u, c, [&](int q, const f32x4 &acc) { plane_store(mine + q * RPLANE, lane, acc); });
    if (e_on) {
#pragma unroll
        for (int k = 0; k < WHR_MAX_GROUPS / 4; ++k) part[erow * (WHR_MAX_GROUPS + 1) + eq * (WHR_MAX_GROUPS / 4) + k] = sp[k];
    }
    __syncthreads();
    if (e_on) {
        const int o = erow * RLD + eq * 4;
        // slab sums ((c0 + c1) + c2) + c3, then the balanced tree in slab order
        auto slab = [&](int z) {
            const float *p = red + (size_t)(4 * z) * RPLANE + o;
            return ((*reinterpret_cast<const f32x4 *>(p) + *reinterpret_cast<const f32x4 *>(p + RPLANE)) + *reinterpret_cast<const f32x4 *>(p + 2 * RPLANE)) + *reinterpret_cast<const f32x4 *>(p + 3 * RPLANE);
        };
        f32x4 v;
        if (g.kz == 8) v = ((slab(0) + slab(1)) + (slab(2) + slab(3))) + ((slab(4) + slab(5)) + (slab(6) + slab(7)));
        else if (g.kz == 4) v = (slab(0) + slab(1)) + (slab(2) + slab(3));
        else if (g.kz == 2) v = slab(0) + slab(1);
        else v = slab(0);
        float t = 0.0f;                                                   // the row's BasicNorm scale, partials added in column order (row_scale())
        for (int j = 0; j < G; ++j) t += part[erow * (WHR_MAX_GROUPS + 1) + j];
        const float rs = __builtin_amdgcn_rsqf(t * g.r_scale.inv_n + g.r_scale.eps);
        if (e_ok) {
            gstore<f32x4>(g.state + (size_t)eslot * g.ld_state + en, v);
            gstore<f32x4>(g.out + (size_t)erow * g.ldo + en, eres * rs + v);
        }
    }
}

template <int TB> __global__ __launch_bounds__(256, TB <= 16 ? 3 : 2) void recur_gates_kernel(GemmArgs g) { recur_gates_body<TB>(g); }
template <int TB> __global__ __launch_bounds__(256, TB <= 16 ? 3 : 2) void recur_gates_zkernel(const GemmArgs *__restrict__ zargs) { const GemmArgs g = zargs[blockIdx.y]; recur_gates_body<TB>(g); }
template <int TB> __global__ __launch_bounds__(512, 4) void recur_whr_kernel(GemmArgs g) { recur_whr_body<TB>(g); }
template <int TB> __global__ __launch_bounds__(512, 4) void recur_whr_zkernel(const GemmArgs *__restrict__ zargs) { const GemmArgs g = zargs[blockIdx.y]; recur_whr_body<TB>(g); }

// blocks per wave that have a kernel: gates = k blocks per chunk (K / 64), projection = K / 16 / waves
#define APRIL_RECUR_GATES_TB(X) X(1) X(2) X(3) X(4) X(6) X(8) X(12) X(16) X(24)
#define APRIL_RECUR_WHR_TB(X) X(1) X(2) X(3) X(4) X(5) X(6) X(8) X(10) X(12)
bool gates_tb_ok(int tb) {
#define X(n) if (tb == n) return true;
    APRIL_RECUR_GATES_TB(X)
#undef X
    return false;
}
bool whr_tb_ok(int tb) {
#define X(n) if (tb == n) return true;
    APRIL_RECUR_WHR_TB(X)
#undef X
    return false;
}
int whr_waves(const GemmArgs &g) { const int nch = 4 * g.kz; return nch < 8 ? nch : 8; }

int recur_enabled()
{
    static const int v = [] { const char *e = getenv("APRIL_RECUR_KERNELS"); return e && *e ? atoi(e) : 1; }();
    return v;
}

}  // namespace

// 1 = gates form, 2 = projection form, 0 = not one of the two (the general kernels take it)
int recur_form(const GemmArgs &g)
{
    if (!recur_enabled() || g.wt != 0 || g.a_op != AOP_NONE || g.M < 1 || g.M > 16 || g.K % 64 != 0) return 0;
    const int KB = g.K / 16;
    if (g.epi == EPI_LSTM && g.wave_mask == 0xC && g.p_add && g.kz == 1 && g.K0 * 2 == g.K && g.K1 == g.K0 && g.N % 32 == 0 && KB % 4 == 0 && gates_tb_ok(KB / 4) && !g.x_scale.ssq && g.out && !g.out16)
        return 1;
    if (g.epi == EPI_HR && g.wave_mask == 0xF && g.K1 == 0 && (g.kz == 1 || g.kz == 2 || g.kz == 4 || g.kz == 8) && KB % (4 * g.kz) == 0 && g.N % 16 == 0 &&
        whr_tb_ok(KB / whr_waves(g)) && g.r_scale.ssq && g.r_scale.groups <= WHR_MAX_GROUPS && !g.out16 && !g.state16)
        return 2;
    return 0;
}

// n problems of one shape (dev_args: their argument blocks in device memory) or one problem by value (dev_args == null)
void launch_recur(const GemmArgs &g, int form, const GemmArgs *dev_args, int n, hipStream_t s)
{
    const int KB = g.K / 16;
    if (form == 1) {
        const dim3 grid((unsigned)(g.N / 32), (unsigned)(dev_args ? n : 1), 1);
        switch (KB / 4) {
#define X(tb) case tb: if (dev_args) hipLaunchKernelGGL(recur_gates_zkernel<tb>, grid, dim3(256), 0, s, dev_args); else hipLaunchKernelGGL(recur_gates_kernel<tb>, grid, dim3(256), 0, s, g); return;
        APRIL_RECUR_GATES_TB(X)
#undef X
        }
    } else {
        const int nw = whr_waves(g);
        const dim3 grid((unsigned)(g.N / 16), (unsigned)(dev_args ? n : 1), 1);
        switch (KB / nw) {
#define X(tb) case tb: if (dev_args) hipLaunchKernelGGL(recur_whr_zkernel<tb>, grid, dim3(64 * nw), 0, s, dev_args); else hipLaunchKernelGGL(recur_whr_kernel<tb>, grid, dim3(64 * nw), 0, s, g); return;
        APRIL_RECUR_WHR_TB(X)
#undef X
        }
    }
    fprintf(stderr, "libapril(mi355x): launch_recur: no kernel for form %d, K = %d, kz = %d\n", form, g.K, g.kz);
    abort();
}

}  // namespace aprilx
