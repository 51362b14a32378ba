// fp32 MFMA "skinny" GEMM for gfx950: out[M,N] = A[M,K] x W[K,N], M = sessions
// stepped together (1 .. thousands), W streamed once per workgroup column from a
// pre-packed HBM layout (see kernels.h).  v_mfma_f32_16x16x4_f32 is an exact
// in-order fp32 FMA chain, so results do not depend on M or on the tile shape.
//
// Workgroup = 256 threads = 4 waves; the waves split the K range of the
// workgroup's slab (split-K inside the CU keeps >= 4 independent 16-byte load
// streams per CU in flight at M <= 16, where the kernel is HBM-bound), partial
// tiles meet in LDS and the epilogue runs on the summed tile:
//   EPI_PARTIAL      slab sums -> workspace (row kernels finish bias/residual/norm)
//   EPI_LSTM         LSTM cell: sigma/tanh gates, c' written in place, u = sigma(o) tanh(c')
//   EPI_BIAS_DSWISH  y = acc + b ; y * sigmoid(y - 1)
// Replaces the ORT MatMul/Gemm nodes of the encoder/decoder/joiner graphs
// (reference call sites src/april_session.c:145,160,176).
#include "kernels.h"

namespace aprilx {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int MT, int NT>
struct TileCfg {
    static constexpr int BM = MT * 16, BN = NT * 16, LDR = BN + 4;
    static constexpr int LDS_FLOATS = 4 * BM * LDR;
};

template <int MT, int NT, int EPI, int AOP>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g)
{
    using Cfg = TileCfg<MT, NT>;
    extern __shared__ __attribute__((aligned(16))) float red[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware mapping: consecutive blockIdx.x land on different XCDs, so keep the
    // M-blocks that share one weight column on the same XCD (same x mod 8).
    const int nt0 = blockIdx.x * NT;
    const int m0 = blockIdx.y * Cfg::BM;
    const int z = blockIdx.z;
    const int KB = g.K >> 4;
    const int zb0 = (int)(((long)KB * z) / g.kz), zb1 = (int)(((long)KB * (z + 1)) / g.kz);
    const int nb = zb1 - zb0;
    const int wb0 = zb0 + (nb * wave) / 4, wb1 = zb0 + (nb * (wave + 1)) / 4;

    const int mrow = lane & 15, kq = lane >> 4;

    const float *arow0[MT], *arow0b[MT], *arow1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = m0 + mt * 16 + mrow;
        if (row >= g.M) row = g.M - 1;                        // padding rows recompute the last row; never stored
        const int r0 = g.aidx0 ? g.aidx0[row] : row;
        arow0[mt] = g.a0 + (size_t)r0 * g.lda0;
        arow0b[mt] = (AOP == AOP_TANH_ADD) ? g.a0b + (size_t)r0 * g.lda0 : nullptr;
        if (g.K1 > 0) { const int r1 = g.aidx1 ? g.aidx1[row] : row; arow1[mt] = g.a1 + (size_t)r1 * g.lda1; }
        else arow1[mt] = nullptr;
    }
    const f32x4 *wbase[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wbase[nt] = reinterpret_cast<const f32x4 *>(g.wp) + ((size_t)(nt0 + nt) * KB) * 64 + lane;
    const bool stream_once = gridDim.y == 1;   // weights read by exactly one workgroup: bypass-friendly loads

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_a = [&](int kb, f32x4 (&a)[MT]) {
        const int k = kb * 16 + kq * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (k < g.K0) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(arow0[mt] + k);
                if (AOP == AOP_TANH_ADD) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(arow0b[mt] + k);
                    v.x = tanhf(v.x + w.x); v.y = tanhf(v.y + w.y); v.z = tanhf(v.z + w.z); v.w = tanhf(v.w + w.w);
                }
                a[mt] = v;
            } else {
                a[mt] = *reinterpret_cast<const f32x4 *>(arow1[mt] + (k - g.K0));
            }
        }
    };
    auto load_b = [&](int kb, f32x4 (&b)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            b[nt] = stream_once ? __builtin_nontemporal_load(wbase[nt] + (size_t)kb * 64) : wbase[nt][(size_t)kb * 64];
    };

    // Register-staged software pipeline: DEPTH k-blocks of loads are in flight per wave while the
    // MFMAs of the oldest stage issue.  Small batches (MT == 1) are HBM-bound weight streams and want
    // many bytes in flight per CU (6 x 4 waves x (1+NT) KiB); large tiles are MFMA-bound and need
    // only enough depth to cover L2 latency.
    constexpr int DEPTH = (MT == 1) ? 6 : (MT == 2 ? 3 : 2);
    if (wb0 < wb1) {
        f32x4 a_st[DEPTH][MT], b_st[DEPTH][NT];
        const int last = wb1 - 1;
        // Loads are UNCONDITIONAL (indices clamped to the wave's last block) so the loop body is
        // straight-line code and the compiler can keep counted s_waitcnt vmcnt(N) instead of draining.
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) { const int kb = min(wb0 + s, last); load_a(kb, a_st[s]); load_b(kb, b_st[s]); }
        auto compute = [&](const f32x4 (&a)[MT], const f32x4 (&b)[NT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b[nt].w, acc[mt][nt], 0, 0, 0);
                }
        };
        int kb0 = wb0;
        for (; kb0 + DEPTH <= wb1; kb0 += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                compute(a_st[s], b_st[s]);
                // pin the refill right behind its stage's MFMAs: left alone, the scheduler sinks all
                // refills to the loop end and the next iteration waits out a full memory latency
                __builtin_amdgcn_sched_barrier(0);
                const int nk = min(kb0 + DEPTH + s, last);
                load_a(nk, a_st[s]); load_b(nk, b_st[s]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int rem = wb1 - kb0;       // < DEPTH; stage s already holds block kb0 + s
#pragma unroll
        for (int s = 0; s < DEPTH - 1; ++s)
            if (s < rem) compute(a_st[s], b_st[s]);
    }

    // ---- meet in LDS: red[wave][row][col]
    float *mine = red + (size_t)wave * Cfg::BM * Cfg::LDR;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                mine[(mt * 16 + kq * 4 + r) * Cfg::LDR + nt * 16 + mrow] = acc[mt][nt][r];
    __syncthreads();

    constexpr int PLANE = Cfg::BM * Cfg::LDR;
    auto summed = [&](int row, int col) {
        const int o = row * Cfg::LDR + col;
        return ((red[o] + red[PLANE + o]) + red[2 * PLANE + o]) + red[3 * PLANE + o];
    };

    if (EPI == EPI_PARTIAL) {
        for (int e = threadIdx.x; e < Cfg::BM * Cfg::BN; e += 256) {
            const int row = e / Cfg::BN, col = e % Cfg::BN;
            const int m = m0 + row;
            if (m < g.M) g.out[((size_t)z * g.m_stride + m) * g.N + nt0 * 16 + col] = summed(row, col);
        }
    } else if (EPI == EPI_BIAS_DSWISH) {
        for (int e = threadIdx.x; e < Cfg::BM * Cfg::BN; e += 256) {
            const int row = e / Cfg::BN, col = e % Cfg::BN;
            const int m = m0 + row, n = nt0 * 16 + col;
            if (m < g.M) {
                const float y = summed(row, col) + g.bias[n];
                g.out[(size_t)m * g.ldo + n] = y * sigmoidf_(y - 1.0f);
            }
        }
    } else {   // EPI_LSTM: 4 consecutive columns = gates i,f,g,o of one hidden unit
        constexpr int UN = Cfg::BN / 4;
        for (int e = threadIdx.x; e < Cfg::BM * UN; e += 256) {
            const int row = e / UN, ul = e % UN;
            const int m = m0 + row;
            if (m >= g.M) continue;
            const int n = nt0 * 16 + ul * 4;
            const int unit = n >> 2;
            const float gi = summed(row, ul * 4 + 0) + g.bias[n + 0];
            const float gf = summed(row, ul * 4 + 1) + g.bias[n + 1];
            const float gg = summed(row, ul * 4 + 2) + g.bias[n + 2];
            const float go = summed(row, ul * 4 + 3) + g.bias[n + 3];
            float *cp = g.c_state + (size_t)g.slot_idx[m] * g.hidden + unit;
            const float c_new = sigmoidf_(gf) * (*cp) + sigmoidf_(gi) * tanhf(gg);
            *cp = c_new;
            g.out[(size_t)m * g.ldo + unit] = sigmoidf_(go) * tanhf(c_new);
        }
    }
}

template <int MT, int NT>
static void dispatch(const GemmArgs &g, hipStream_t s)
{
    using Cfg = TileCfg<MT, NT>;
    dim3 grid((unsigned)(g.N / Cfg::BN), (unsigned)((g.M + Cfg::BM - 1) / Cfg::BM), (unsigned)g.kz);
    const size_t lds = (size_t)Cfg::LDS_FLOATS * sizeof(float);
#define LAUNCH(E, A) hipLaunchKernelGGL((gemm_f32_kernel<MT, NT, E, A>), grid, dim3(256), lds, s, g)
    if (g.epi == EPI_PARTIAL) { if (g.a_op == AOP_TANH_ADD) LAUNCH(EPI_PARTIAL, AOP_TANH_ADD); else LAUNCH(EPI_PARTIAL, AOP_NONE); }
    else if (g.epi == EPI_LSTM) LAUNCH(EPI_LSTM, AOP_NONE);
    else LAUNCH(EPI_BIAS_DSWISH, AOP_NONE);
#undef LAUNCH
}

// Tile choice depends on M only through occupancy; numerics are tile-independent.
void launch_gemm(const GemmArgs &g, hipStream_t s)
{
    const int ntiles = g.N / 16;
    int mt = g.M <= 16 ? 1 : (g.M <= 32 ? 2 : 4);
    const int mblocks = (g.M + mt * 16 - 1) / (mt * 16);
    int nt = 4;
    while (nt > 1 && ((ntiles % nt) != 0 || (long)(ntiles / nt) * mblocks * g.kz < 256)) nt >>= 1;
    if (mt == 1) { if (nt == 4) dispatch<1, 4>(g, s); else if (nt == 2) dispatch<1, 2>(g, s); else dispatch<1, 1>(g, s); }
    else if (mt == 2) { if (nt == 4) dispatch<2, 4>(g, s); else if (nt == 2) dispatch<2, 2>(g, s); else dispatch<2, 1>(g, s); }
    else { if (nt == 4) dispatch<4, 4>(g, s); else if (nt == 2) dispatch<4, 2>(g, s); else dispatch<4, 1>(g, s); }
}

}  // namespace aprilx
