// The layer GEMMs at a handful of rows (M <= 16: one session handing over a file, or a few sessions streaming) as WEIGHT STREAMS.
// Reference path: the per-chunk encoder call of src/april_session.c:431-476.  At these sizes a GEMM is its weight matrix read once
// (the recurrent pair of a long feed's time step: 96 MB of gate weights + 24 MB of projection weights for twelve layers per launch),
// and the general kernels (kernels_gemm.hip) run them as a few dozen 16 x 16 / 16 x 32 tiles with 6 KB of weights in flight per
// wave -- 20.9 + 9.3 us per time step of a long feed against 14.9 + 3.6 us for the same bytes at the streaming rate of this GPU
// (tools/bw_probe: 6.8 TB/s from 48 MB up; the Infinity Cache adds nothing), 12.8 us for the 4 MB FFN-down GEMM of one session.
// Here every wave owns whole CHUNKS (the unit of the canonical summation) of one 16-column tile and has its whole k range in
// flight before its first MFMA.  Six forms:
//   cell forms (kz = 1, four waves)   gates recurrent half + LSTM cell (long feed, p_add), gates input half (long feed, EPI_XPART),
//                                     the one-launch gates GEMM + cell (streaming), FFN up + DoubleSwish
//   row forms (all of K, <= 16 waves) projection (EPI_HR), FFN down / bias + residual + sums of squares (EPI_RESID_SSQ)
//
// Arithmetic: the chains are those of gemm_body -- a chunk is ONE in-order chain of v_mfma_f32_16x16x4_f32 over its k blocks
// (k step j of a block = element j of the lane's operand quads), chunks meet as ((c0 + c1) + c2) + c3, slabs in the balanced
// pairwise tree, epilogues term by term -- so the results are bit-identical to the general kernels at any batch size and to the
// layer-major schedule (tests/test_gpu_recur_kernels.py runs the same sessions with these kernels off and on; the batch-invariance
// and layer-major == streaming tests of tests/test_gpu_parity.py cross the same boundary).
#include "kernels.h"
#include "device_utils.h"
#include <atomic>
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
// blocks, each finished chain handed to done(chunk number within the wave, acc).  RB weight blocks and RA activation blocks are
// in flight (rings; all of the stream before the first MFMA when TB <= both).  RA < RB where registers are short: the weights
// come from HBM, the activation rows from L2, so the late half of the activation ring costs an L2 round trip, not an HBM one.
template <int TB, int RB_, int RA_, class F>
__device__ __forceinline__ void stream_chains(const char *ap, uint32_t ao, const char *bp, uint32_t bo, int c, F done)
{
    constexpr int RB = TB < RB_ ? TB : RB_, RA = TB < RA_ ? TB : RA_;
    static_assert(RA <= RB, "the activation ring is the shorter one");
    f32x4 a[RA], b[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        if (i < RA) a[i] = gload<f32x4>(ap + (size_t)i * 64 + ao);
        b[i] = gload_nt<f32x4>(bp + (size_t)i * 1024 + bo);
    }
    __builtin_amdgcn_sched_barrier(0);              // (left alone, the scheduler sinks each load to its MFMAs: four blocks in flight instead of the ring)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int in_chunk = 0, chunk = 0;
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        const int sa = i % RA, sb = i % RB;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sa].x, b[sb].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sa].y, b[sb].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sa].z, b[sb].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sa].w, b[sb].w, acc, 0, 0, 0);
        if (i + RA < TB) a[sa] = gload<f32x4>(ap + (size_t)(i + RA) * 64 + ao);
        if (i + RB < TB) b[sb] = gload_nt<f32x4>(bp + (size_t)(i + RB) * 1024 + bo);
        if (++in_chunk == c) { done(chunk, acc); ++chunk; in_chunk = 0; acc = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
}

__device__ __forceinline__ void plane_store(float *plane, int lane, const f32x4 &acc)
{
    const int col = lane & 15, r0 = (lane >> 4) * 4;        // accumulator register r = output row 4 (lane / 16) + r, column lane % 16
#pragma unroll
    for (int r = 0; r < 4; ++r) plane[(r0 + r) * RLD + col] = acc[r];
}

// the rows' BasicNorm scales (RowScale): threads 0..63 (row = t / 4) fetch eight sum-of-squares partials each before the weight
// stream, park them in LDS after it; row_scale_sum adds a row's partials in column order (the order of row_scale())
constexpr int MAX_GROUPS = 32;
struct ScalePart { float v[MAX_GROUPS / 4]; };
__device__ __forceinline__ void scale_fetch(const RowScale &rs, int M, ScalePart &sp)
{
    const int row = (int)threadIdx.x >> 2, q = threadIdx.x & 3;
    const int m = row < M ? row : M - 1;
#pragma unroll
    for (int k = 0; k < MAX_GROUPS / 4; ++k) { const int j = q * (MAX_GROUPS / 4) + k; sp.v[k] = gload<float>(rs.ssq + (size_t)m * rs.groups + (j < rs.groups ? j : rs.groups - 1)); }
}
__device__ __forceinline__ void scale_park(float *part, const ScalePart &sp)
{
    const int row = (int)threadIdx.x >> 2, q = threadIdx.x & 3;
#pragma unroll
    for (int k = 0; k < MAX_GROUPS / 4; ++k) part[row * (MAX_GROUPS + 1) + q * (MAX_GROUPS / 4) + k] = sp.v[k];
}
__device__ __forceinline__ float row_scale_sum(const float *part, int row, const RowScale &rs)
{
    float t = 0.0f;
    for (int j = 0; j < rs.groups; ++j) t += part[row * (MAX_GROUPS + 1) + j];
    return __builtin_amdgcn_rsqf(t * rs.inv_n + rs.eps);
}

// ---- the fused-epilogue GEMMs whose K is one slab (kz = 1, four chunks), four waves per workgroup:
//   CF_GATES_H  gates, recurrent half (wave_mask 0b1100, p_add): 2 tiles x chunks 2, 3;  ((P + c2) + c3) + bias, LSTM cell
//   CF_XPART    gates, input half (wave_mask 0b0011, EPI_XPART): 2 tiles x chunks 0, 1;  P = (c0 + c1) (* row scale)
//   CF_GATES    the one-launch gates GEMM of a chunk step:        1 tile x chunks 0..3;  (((c0 + c1) * scale + c2) + c3) + bias, cell
//   CF_DSWISH   feed-forward up (EPI_BIAS_DSWISH):                1 tile x chunks 0..3;  y = (((c0 + c1) + c2) + c3) + bias, y sigma(y - 1)
enum { CF_GATES_H = 0, CF_XPART = 1, CF_GATES = 2, CF_DSWISH = 3 };

template <int TB, int FORM>
__device__ __forceinline__ void recur_cell_body(const GemmArgs &g)
{
    constexpr bool HALF = FORM == CF_GATES_H || FORM == CF_XPART;
    constexpr bool CELL = FORM == CF_GATES_H || FORM == CF_GATES;
    constexpr int TPW = HALF ? 2 : 1;                                      // column tiles per workgroup
    __shared__ __attribute__((aligned(16))) float red[4 * RPLANE];
    __shared__ float part[16 * (MAX_GROUPS + 1)];
    if (g.run_flag && gload<int>(g.run_flag) != g.run_gen) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int KB = 4 * TB, c = TB;                                     // (checked on the host: K = 64 TB)
    const int wtile = HALF ? (wave >> 1) : 0;
    const int chunk = FORM == CF_GATES_H ? 2 + (wave & 1) : (FORM == CF_XPART ? (wave & 1) : wave);
    const int ct = blockIdx.x * TPW + wtile;
    const bool scaled = (FORM == CF_XPART || FORM == CF_GATES) && g.x_scale.ssq != nullptr;
    // epilogue operands of this thread's (tile, row, quad), fetched before the stream (row -> slot -> previous cell value is a dependent pair)
    const int et = threadIdx.x >> 6, erow = (threadIdx.x >> 2) & 15, eul = threadIdx.x & 3;
    const bool e_on = (int)threadIdx.x < 64 * TPW, e_ok = e_on && erow < g.M;
    const int en = (blockIdx.x * TPW + et) * 16 + eul * 4, eunit = en >> 2;
    const int em = erow < g.M ? erow : g.M - 1;
    float *cptr = nullptr;
    float cprev = 0.0f;
    f32x4 ebias = {0.f, 0.f, 0.f, 0.f}, xin = {0.f, 0.f, 0.f, 0.f};
    ScalePart sp;
    if (scaled && threadIdx.x < 64) scale_fetch(g.x_scale, g.M, sp);
    if (e_on) {
        if (CELL) {
            cptr = g.c_state + (size_t)gload<int>(g.slot_idx + em) * g.hidden + eunit;
            cprev = gload<float>(cptr);
        }
        if (FORM != CF_XPART) ebias = gload<f32x4>(g.bias + en);
        if (FORM == CF_GATES_H) xin = gload<f32x4>(g.p_add + (size_t)em * g.ldp + en);
    }
    // this wave's chunk: k range [chunk c 16, (chunk + 1) c 16) of the A row, which lies in K segment 0 or 1 as a whole
    int row = lane & 15;
    if (row >= g.M) row = g.M - 1;                                         // padding rows recompute the last row; never stored
    const int koff = chunk * c * 16;
    const bool seg1 = g.K1 > 0 && koff >= g.K0;                            // uniform
    const int *aidx = seg1 ? g.aidx1 : g.aidx0;
    const int arow = aidx ? gload<int>(aidx + row) : row;
    // (all operands are far below 4 GiB per array: 32-bit lane offsets)
    const char *ap = reinterpret_cast<const char *>(seg1 ? g.a1 : g.a0) + (size_t)(seg1 ? koff - g.K0 : koff) * sizeof(float);
    const uint32_t ao = (uint32_t)(((size_t)arow * (seg1 ? g.lda1 : g.lda0) + (lane >> 4) * 4) * sizeof(float));
    const char *bp = reinterpret_cast<const char *>(g.wp) + ((size_t)ct * KB + (size_t)chunk * c) * 1024;
    float *mine = red + wave * RPLANE;
    stream_chains<TB, 16, 16>(ap, ao, bp, (uint32_t)lane * 16u, c, [&](int, const f32x4 &acc) { plane_store(mine, lane, acc); });
    if (scaled && threadIdx.x < 64) scale_park(part, sp);
    __syncthreads();
    if (e_on) {
        const int o = erow * RLD + eul * 4;
        const float *pl = red + (HALF ? 2 * et : 0) * RPLANE + o;           // the tile's planes, in chunk order
        const f32x4 pa = *reinterpret_cast<const f32x4 *>(pl), pb = *reinterpret_cast<const f32x4 *>(pl + RPLANE);
        if (FORM == CF_XPART) {
            const f32x4 x = scaled ? (pa + pb) * row_scale_sum(part, erow, g.x_scale) : (pa + pb);
            if (e_ok) gstore<f32x4>(g.out + (size_t)erow * g.ldo + en, x);
        } else if (FORM == CF_DSWISH) {
            const f32x4 y = (((pa + pb) + *reinterpret_cast<const f32x4 *>(pl + 2 * RPLANE)) + *reinterpret_cast<const f32x4 *>(pl + 3 * RPLANE)) + ebias;
            f32x4 v;
            v.x = y.x * fast_sigmoid(y.x - 1.0f); v.y = y.y * fast_sigmoid(y.y - 1.0f);
            v.z = y.z * fast_sigmoid(y.z - 1.0f); v.w = y.w * fast_sigmoid(y.w - 1.0f);
            if (e_ok) gstore<f32x4>(g.out + (size_t)erow * g.ldo + en, v);
        } else {
            f32x4 gt;
            if (FORM == CF_GATES_H) gt = ((xin + pa) + pb) + ebias;
            else {
                const f32x4 p2 = *reinterpret_cast<const f32x4 *>(pl + 2 * RPLANE), p3 = *reinterpret_cast<const f32x4 *>(pl + 3 * RPLANE);
                if (scaled) gt = (((pa + pb) * row_scale_sum(part, erow, g.x_scale) + p2) + p3) + ebias;
                else gt = (((pa + pb) + p2) + p3) + ebias;
            }
            const float c_new = fast_sigmoid(gt.y) * cprev + fast_sigmoid(gt.x) * fast_tanh(gt.z);
            const float u = fast_sigmoid(gt.w) * fast_tanh(c_new);
            if (e_ok) { gstore<float>(cptr, c_new); gstore<float>(g.out + (size_t)erow * g.ldo + eunit, u); }
        }
    }
}

// ---- the row-epilogue GEMMs (all of K in the workgroup, 4 kz chunks per column tile dealt to the waves, consecutive chunks per wave):
//   RF_HR         projection (EPI_HR): one 16-column tile, <= 8 waves;  h' = tree, state row, out = x * scale(x) + h'
//   RF_RESID_SSQ  feed-forward down / embed linear (EPI_RESID_SSQ): two tiles = one 32-column sum-of-squares granule, <= 16 waves;
//                 y = (resid +) tree + bias, out, ssq
enum { RF_HR = 0, RF_RESID_SSQ = 1 };

template <int TB, int FORM>
__device__ __forceinline__ void recur_row_body(const GemmArgs &g)
{
    constexpr int NTILE = FORM == RF_RESID_SSQ ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float dyn[];           // [chains][RPLANE] planes, then the scale partials
    if (g.run_flag && gload<int>(g.run_flag) != g.run_gen) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    // K cut across S workgroups (g.ksplit; 1 = the whole K here): a column granule's chunks are dealt to S consecutive workgroups in slab
    // order, kz / S whole slabs each -- see the hand-over below
    const int S = g.ksplit > 1 ? g.ksplit : 1;
    const int gran = (int)blockIdx.x / S, slice = (int)blockIdx.x - gran * S;
    const int KB = g.K >> 4, nchk = 4 * g.kz, nch = nchk / S, c = KB / nchk, cpw = NTILE * nch / nw;      // nch: chunks per column tile in THIS workgroup (checked on the host: cpw * c == TB)
    float *red = dyn, *part = dyn + (size_t)NTILE * nch * RPLANE;
    const int first = wave * cpw;                                         // this wave's chains: tile first / nch, chunks first % nch ..
    const int ct = gran * NTILE + first / nch, kb0 = (slice * nch + first % nch) * c;
    // epilogue operands.  RF_HR: threads 0..63 = (row, quad).  RF_RESID_SSQ: threads 0..127 = (row, 8 quads of the 32-column granule)
    const int erow = FORM == RF_HR ? (int)threadIdx.x >> 2 : (int)threadIdx.x >> 3, eq = FORM == RF_HR ? (threadIdx.x & 3) : (threadIdx.x & 7);
    const bool e_on = (int)threadIdx.x < 64 * NTILE, e_ok = e_on && erow < g.M;
    const int em = erow < g.M ? erow : g.M - 1;
    const int en = gran * NTILE * 16 + eq * 4;
    int eslot = 0;
    f32x4 eres = {0.f, 0.f, 0.f, 0.f}, ebias = {0.f, 0.f, 0.f, 0.f};
    ScalePart sp;
    if (FORM == RF_HR && threadIdx.x < 64) scale_fetch(g.r_scale, g.M, sp);
    if (e_on) {
        if (FORM == RF_HR) eslot = g.slot_idx ? gload<int>(g.slot_idx + em) : em;
        if (FORM == RF_HR || g.resid) eres = gload<f32x4>(g.resid + (size_t)em * g.ldr + en);
        if (FORM == RF_RESID_SSQ) ebias = gload<f32x4>(g.bias + en);
    }
    int row = lane & 15;
    if (row >= g.M) row = g.M - 1;
    const int arow = g.aidx0 ? gload<int>(g.aidx0 + row) : row;
    const char *ap = reinterpret_cast<const char *>(g.a0) + (size_t)kb0 * 64;
    const uint32_t ao = (uint32_t)(((size_t)arow * g.lda0 + (lane >> 4) * 4) * sizeof(float));
    const char *bp = reinterpret_cast<const char *>(g.wp) + ((size_t)ct * KB + kb0) * 1024;
    float *mine = red + (size_t)first * RPLANE;
    // (these workgroups run at four waves per SIMD, 128 registers per lane: sixteen weight blocks and eight activation blocks in flight, twelve and eight where the ring wraps)
    stream_chains<TB, ((TB > 16 || (FORM == RF_HR && TB > 12)) ? 12 : 16), (FORM == RF_RESID_SSQ || TB > 12) ? 8 : 16>(ap, ao, bp, (uint32_t)lane * 16u, c, [&](int q, const f32x4 &acc) { plane_store(mine + q * RPLANE, lane, acc); });
    if (FORM == RF_HR && threadIdx.x < 64) scale_park(part, sp);
    __syncthreads();
    const int tile = FORM == RF_HR ? 0 : eq >> 2;
    const int o = erow * RLD + (eq & 3) * 4;
    // slab sums ((c0 + c1) + c2) + c3 (z: slab number within this workgroup), then the balanced tree in slab order
    auto slab = [&](int z) {
        const float *p = red + (size_t)(tile * nch + 4 * z) * RPLANE + o;
        return ((*reinterpret_cast<const f32x4 *>(p) + *reinterpret_cast<const f32x4 *>(p + RPLANE)) + *reinterpret_cast<const f32x4 *>(p + 2 * RPLANE)) + *reinterpret_cast<const f32x4 *>(p + 3 * RPLANE);
    };
    f32x4 sl[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) sl[z] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (S > 1) {
        // Hand-over between the S workgroups of a granule, without a spin and without a fence: every workgroup leaves its slab sums (rows < M
        // only: 16 bytes per row, quad and slab) in the launch's workspace with agent-scope stores (they pass the XCD's L2, which is not
        // coherent with the other seven), waits until they have been acknowledged, and counts itself in on the granule's word; whoever finds
        // S - 1 there is the last one, re-arms the word, fetches all kz slab sums (agent-scope loads) and finishes tree + epilogue exactly as
        // the one-workgroup form does.  The others are done.  Same slab sums, same tree => the same bits as S = 1.
        const int zs = g.kz / S, ncol = 16 * NTILE;
        using gfloat = __attribute__((address_space(1))) float;
        using guint = __attribute__((address_space(1))) unsigned;
        gfloat *wsg = (gfloat *)(g.ks_ws) + (size_t)gran * g.kz * 16 * ncol;
        guint *cnt = (guint *)(g.ks_cnt) + gran;
        if (e_ok) {
            for (int z = 0; z < zs; ++z) {
                const f32x4 t = slab(z);
                gfloat *q = wsg + ((size_t)(slice * zs + z) * 16 + erow) * ncol + tile * 16 + (eq & 3) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) __hip_atomic_store(q + j, t[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): this wave's stores have been acknowledged
        asm volatile("" ::: "memory");
        __shared__ int ks_last;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(S - 1);
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ks_last = last;
        }
        __syncthreads();
        if (!ks_last) return;
        if (e_ok) {
            for (int z = 0; z < g.kz; ++z) {
                const gfloat *q = wsg + ((size_t)z * 16 + erow) * ncol + tile * 16 + (eq & 3) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) sl[z][j] = __hip_atomic_load(q + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (e_on) {
        f32x4 v;
        if (S > 1) {
            // the last of the granule's S workgroups (the hand-over above) holds every slab sum in sl[]
            if (g.kz == 8) v = ((sl[0] + sl[1]) + (sl[2] + sl[3])) + ((sl[4] + sl[5]) + (sl[6] + sl[7]));
            else if (g.kz == 4) v = (sl[0] + sl[1]) + (sl[2] + sl[3]);
            else v = sl[0] + sl[1];
        }
        else if (g.kz == 8) v = ((slab(0) + slab(1)) + (slab(2) + slab(3))) + ((slab(4) + slab(5)) + (slab(6) + slab(7)));
        else if (g.kz == 4) v = (slab(0) + slab(1)) + (slab(2) + slab(3));
        else if (g.kz == 2) v = slab(0) + slab(1);
        else v = slab(0);
        if (FORM == RF_HR) {
            const float rs = row_scale_sum(part, erow, g.r_scale);
            if (e_ok) {
                gstore<f32x4>(g.state + (size_t)eslot * g.ld_state + en, v);
                gstore<f32x4>(g.out + (size_t)erow * g.ldo + en, eres * rs + v);
            }
        } else {
            f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e_ok) {
                y = v + ebias;
                if (g.resid) y = eres + y;
                gstore<f32x4>(g.out + (size_t)erow * g.ldo + en, y);
            }
            const float ss = granule_ssq(y);                             // all lanes of waves 0 and 1 take part in the shuffles
            if (e_ok && eq == 0) gstore<float>(g.ssq_out + (size_t)erow * (g.N / SSQ_COLS) + en / SSQ_COLS, ss);
        }
    }
}

template <int TB, int FORM> __global__ __launch_bounds__(256, TB <= 16 ? 3 : 2) void recur_cell_kernel(GemmArgs g) { recur_cell_body<TB, FORM>(g); }
template <int TB, int FORM> __global__ __launch_bounds__(256, TB <= 16 ? 3 : 2) void recur_cell_zkernel(const GemmArgs *__restrict__ zargs) { const GemmArgs g = zargs[blockIdx.y]; recur_cell_body<TB, FORM>(g); }
template <int TB, int FORM> __global__ __launch_bounds__(FORM == RF_HR ? 512 : 1024, 4) void recur_row_kernel(GemmArgs g) { recur_row_body<TB, FORM>(g); }
template <int TB, int FORM> __global__ __launch_bounds__(FORM == RF_HR ? 512 : 1024, 4) void recur_row_zkernel(const GemmArgs *__restrict__ zargs) { const GemmArgs g = zargs[blockIdx.y]; recur_row_body<TB, FORM>(g); }

// blocks per wave that have a kernel: cell forms = k blocks per chunk (K / 64), row forms = chunks per wave x blocks per chunk
#define APRIL_RECUR_CELL_TB(X) X(1) X(2) X(3) X(4) X(6) X(8) X(12) X(16) X(24)
#define APRIL_RECUR_ROW_TB(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(10) X(12) X(16) X(24)
bool cell_tb_ok(int tb) {
#define X(n) if (tb == n) return true;
    APRIL_RECUR_CELL_TB(X)
#undef X
    return false;
}
bool row_tb_ok(int tb) {
#define X(n) if (tb == n) return true;
    APRIL_RECUR_ROW_TB(X)
#undef X
    return false;
}
int row_waves(const GemmArgs &g, int ntile) { const int chains = 4 * g.kz * ntile / (g.ksplit > 1 ? g.ksplit : 1), cap = ntile == 2 ? 16 : 8; return chains < cap ? chains : cap; }

int recur_enabled()
{
    static const int v = [] { const char *e = getenv("APRIL_RECUR_KERNELS"); return e && *e ? atoi(e) : 1; }();
    return v;
}

template <int FORM>
void launch_cell(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    constexpr int TPW = (FORM == CF_GATES_H || FORM == CF_XPART) ? 2 : 1;
    const dim3 grid((unsigned)(g.N / (16 * TPW)), (unsigned)(dev_args ? n : 1), 1);
    switch (g.K / 64) {
#define X(tb) case tb: if (dev_args) hipLaunchKernelGGL((recur_cell_zkernel<tb, FORM>), grid, dim3(256), 0, s, dev_args); else hipLaunchKernelGGL((recur_cell_kernel<tb, FORM>), grid, dim3(256), 0, s, g); return;
    APRIL_RECUR_CELL_TB(X)
#undef X
    }
    fprintf(stderr, "libapril(mi355x): launch_recur: no cell kernel for K = %d\n", g.K);
    abort();
}

template <int FORM>
void launch_row(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    constexpr int NTILE = FORM == RF_RESID_SSQ ? 2 : 1;
    const int S = g.ksplit > 1 ? g.ksplit : 1;
    const int nw = row_waves(g, NTILE), chains = 4 * g.kz * NTILE / S;      // (chains of ONE workgroup)
    const dim3 grid((unsigned)(g.N / (16 * NTILE) * S), (unsigned)(dev_args ? n : 1), 1);
    const size_t lds = ((size_t)chains * RPLANE + 16 * (MAX_GROUPS + 1)) * sizeof(float);
    switch (g.K / 16 / (4 * g.kz) * (chains / nw)) {
#define X(tb) case tb: { \
        if (lds > 64 * 1024) {      /* dynamic LDS beyond 64 KB has to be announced, per instantiation and device */ \
            static std::atomic<uint64_t> attr_devs{0}; \
            int dev = 0; (void)hipGetDevice(&dev); \
            const uint64_t bit = 1ull << (dev & 63); \
            if (!(attr_devs.load(std::memory_order_acquire) & bit)) { \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&recur_row_kernel<tb, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&recur_row_zkernel<tb, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                attr_devs.fetch_or(bit, std::memory_order_release); \
            } \
        } \
        if (dev_args) hipLaunchKernelGGL((recur_row_zkernel<tb, FORM>), grid, dim3(64 * nw), lds, s, dev_args); \
        else hipLaunchKernelGGL((recur_row_kernel<tb, FORM>), grid, dim3(64 * nw), lds, s, g); \
        return; }
    APRIL_RECUR_ROW_TB(X)
#undef X
    }
    fprintf(stderr, "libapril(mi355x): launch_recur: no row kernel for K = %d, kz = %d\n", g.K, g.kz);
    abort();
}

}  // namespace

// which stream kernel takes g (0: none, the general kernels do)
enum { RECUR_GATES_H = 1, RECUR_HR = 2, RECUR_GATES = 3, RECUR_XPART = 4, RECUR_DSWISH = 5, RECUR_RESID_SSQ = 6 };

int recur_form(const GemmArgs &g)
{
    if (!recur_enabled() || g.wt != 0 || g.a_op != AOP_NONE || g.M < 1 || g.M > 16 || g.K % 64 != 0 || g.out16 || g.state16) return 0;
    const int KB = g.K / 16;
    const bool kz_ok = g.kz == 1 || g.kz == 2 || g.kz == 4 || g.kz == 8;
    const bool halves = g.kz == 1 && g.K0 * 2 == g.K && g.K1 == g.K0 && cell_tb_ok(KB / 4);      // gates: [x | h], one slab
    const bool xs_ok = !g.x_scale.ssq || g.x_scale.groups <= MAX_GROUPS;
    if (g.epi == EPI_LSTM && halves && g.out && g.N % 32 == 0) {
        if (g.wave_mask == 0xC && g.p_add && !g.x_scale.ssq) return RECUR_GATES_H;
        if (g.wave_mask == 0xF && !g.p_add && xs_ok) return RECUR_GATES;
    }
    if (g.epi == EPI_XPART && halves && g.wave_mask == 0x3 && !g.p_add && xs_ok && g.N % 32 == 0) return RECUR_XPART;
    if (g.epi == EPI_BIAS_DSWISH && g.wave_mask == 0xF && g.kz == 1 && g.K1 == 0 && cell_tb_ok(KB / 4) && g.out && g.N % 16 == 0) return RECUR_DSWISH;
    if (g.wave_mask != 0xF || g.K1 != 0 || !kz_ok || KB % (4 * g.kz) != 0) return 0;
    const int S = g.ksplit > 1 ? g.ksplit : 1;
    if (S > 1 && (g.kz % S != 0 || g.kz / S < 1 || !g.ks_ws || !g.ks_cnt)) return 0;      // (recur_ksplit only hands out divisors of kz)
    if (g.epi == EPI_HR && g.N % 16 == 0 && g.r_scale.ssq && g.r_scale.groups <= MAX_GROUPS && row_tb_ok(KB / S / row_waves(g, 1))) return RECUR_HR;
    if (g.epi == EPI_RESID_SSQ && g.N % 32 == 0 && g.ssq_out && row_tb_ok(2 * KB / S / row_waves(g, 2))) return RECUR_RESID_SSQ;
    return 0;
}

// K cut of the row forms across workgroups (GemmArgs::ksplit) -- a MEASUREMENT FORM, off by default.  N = d_model gives 16 (FFN down: 32-column
// granules) or 32 (projection) workgroups per problem; the 4 MB of FFN-down weights of one session's step take 9.8 us in 16 workgroups
// beside 4.9 us for the same bytes of FFN up in 128.  Cutting K across S workgroups per granule with the in-launch hand-over of
// recur_row_body (agent-scope stores, one counter word, the last arriver finishes: no spin, no fence; bit-identical) was built and measured
// in round 6 (tools/kw_bench, profiles/r06_ksplit_bench.txt): the hand-over itself costs 2.7 .. 4 us -- store acknowledgement, the counter's
// round trip, the fetch of the slab sums, three dependent trips to the memory side -- against 1 .. 2 us saved on the stream: projection
// 4.4 -> 7.0 us, FFN down 8.8 -> 10.1 us at one row, 11.8 -> 11.9 at sixteen.  So the whole-K form stays.  S = the power of two that brings
// the launch to ~128 workgroups, at most kz (whole slabs per workgroup); needs the caller's workspace (ks_ws / ks_cnt).
// APRIL_RECUR_KSPLIT: 0 = off (default), 1 = planner, N > 1 = pin S.
static int g_ksplit_pin = -1;
void recur_ksplit_pin(int s) { g_ksplit_pin = s; }
int recur_ksplit(const GemmArgs &g, int n)
{
    static const int env_mode = [] { const char *e = getenv("APRIL_RECUR_KSPLIT"); return e && *e ? atoi(e) : 0; }();
    const int mode = g_ksplit_pin >= 0 ? g_ksplit_pin : env_mode;
    static const int target = [] { const char *e = getenv("APRIL_RECUR_KSPLIT_WGS"); return e && *e ? atoi(e) : 128; }();
    if (!mode || !recur_enabled() || !g.ks_ws || !g.ks_cnt || g.M < 1 || g.M > 16 || g.kz < 2 || (g.epi != EPI_HR && g.epi != EPI_RESID_SSQ)) return 1;
    const int wgs = g.N / (g.epi == EPI_RESID_SSQ ? 32 : 16) * (n > 0 ? n : 1);
    int S = 1;
    if (mode > 1) S = mode; else while (S * 2 <= g.kz && wgs * S < target) S *= 2;
    while (S > 1 && (S > g.kz || g.kz % S != 0)) S >>= 1;
    if (S > 1) {      // (a cut without a kernel for its blocks per wave: keep the whole-K form)
        GemmArgs t = g; t.ksplit = S;
        if (!recur_form(t)) return 1;
    }
    return S;
}

// n problems of one shape (dev_args: their argument blocks in device memory) or one problem by value (dev_args == null)
void launch_recur(const GemmArgs &g, int form, const GemmArgs *dev_args, int n, hipStream_t s)
{
    switch (form) {
    case RECUR_GATES_H: launch_cell<CF_GATES_H>(g, dev_args, n, s); return;
    case RECUR_GATES: launch_cell<CF_GATES>(g, dev_args, n, s); return;
    case RECUR_XPART: launch_cell<CF_XPART>(g, dev_args, n, s); return;
    case RECUR_DSWISH: launch_cell<CF_DSWISH>(g, dev_args, n, s); return;
    case RECUR_HR: launch_row<RF_HR>(g, dev_args, n, s); return;
    case RECUR_RESID_SSQ: launch_row<RF_RESID_SSQ>(g, dev_args, n, s); return;
    }
    fprintf(stderr, "libapril(mi355x): launch_recur: unknown form %d\n", form);
    abort();
}

}  // namespace aprilx
