// GM_KW schedule of the MFMA GEMM (kernels.h) for gfx950: the waves of a workgroup split K (as GM_FULLK does), but the
// ACTIVATION operand of every wave arrives in full 128-byte lines through a wave-private LDS ring (global_load_lds_dwordx4),
// the weights stay on the direct register stream.
//
// Why it exists (round 5): the N = d_model GEMMs of a layer (LSTM projection K = hidden, FFN down K = ffn) at 256..768 rows per
// launch ran as 16 x 32 full-K tiles whose A fragment load -- lane (i, kq) reads 16 bytes of row i -- touches sixteen cache lines
// per quarter wave, 64 bytes used of each: the CU's vector cache delivered ~24 B/clk and the kernels sat at 0.26 / 0.35 of the
// fp32 MFMA peak (profiles/r04_b256_kernel_stats.csv), twice the MFMA time in operand waits.  GM_TILE (waves split the OUTPUT tile,
// operands shared through LDS) needs 64 x 64 tiles to pay and a 256-row, N = 512 problem has only 32 of those.  Here the tile stays
// small (32 x 32 outputs: 128 workgroups per 256-row problem, one eight-wave workgroup per CU at two problems per launch) and K is
// split across the EIGHT waves; since every wave owns its own k range nothing is shared, so there is no barrier in the K loop --
// each wave runs its own DMA ring: a stage is two k blocks = 128 bytes of each of the tile's rows (8 rows per 1 KB DMA
// instruction, eight full lines), stored with the 16-byte segments of row R xor-swizzled by (R >> 1) & 7 so that the MFMA A
// fragment read (ds_read_b128) is conflict-free (the layout of kernels_gemm_tile.hip).
//
// Canonical summation (kernels.h) is kept exactly: a chunk is one in-order MFMA chain over its k blocks, a slab is
// ((c0 + c1) + c2) + c3, slabs meet pairwise in slab order.  Wave w owns chunks [w cpw, (w + 1) cpw), cpw = 4 kz / NW:
//   cpw >= 4  whole slabs: folded in registers (S, then R = S0 + S1 for two slabs), the waves meet once as the balanced tree
//             ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7))   (NW = 8)   or   (p0 + p1) + (p2 + p3)   (NW = 4)
//   cpw == 1  one chunk per wave: the planes meet as ((p0 + p1) + p2) + p3 per slab, two slabs (NW = 8, kz = 2) as s0 + s1
// => bit-identical to GM_SLAB / GM_FULLK / GM_TILE at any batch size (tools/kw_bench compares every output bitwise; the engine's
// batch-invariance tests cross the schedule boundary).
// Epilogues: EPI_HR (projection), EPI_RESID_SSQ (FFN down), EPI_BIAS_DSWISH (FFN up, four waves), term by term those of gemm_body.
// Replaces the MatMul nodes of the encoder graph (reference call site src/april_session.c:131-148).
#include "kernels.h"
#include "device_utils.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace aprilx {

namespace {

template <class T> __device__ __forceinline__ T gload(const void *p) { return *(const __attribute__((address_space(1))) T *)(p); }
template <class T> __device__ __forceinline__ void gstore(void *p, const T &v) { *(__attribute__((address_space(1))) T *)(p) = v; }

template <int N> __device__ __forceinline__ void wait_vm()
{
    // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14; expcnt / lgkmcnt at their maxima)
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }      // lgkmcnt(0) only

// Tile 16 MT x 16 NT per workgroup, NW waves splitting K, D ring stages per wave
template <int MT, int NT, int NW, int D> struct KwGeom {
    static constexpr int BM = 16 * MT, BN = 16 * NT, LDR = BN + 4, NTH = 64 * NW;
    static constexpr int STAGE_BYTES = BM * 128;                          // two k blocks of the tile's rows
    static constexpr int RING_BYTES = D * STAGE_BYTES, PLANE_BYTES = BM * LDR * 4;
    static constexpr int WAVE_BYTES = ((RING_BYTES > PLANE_BYTES ? RING_BYTES : PLANE_BYTES) + 1023) / 1024 * 1024;
    static constexpr int LDS_MAIN = NW * WAVE_BYTES;
    static constexpr int PLANE = WAVE_BYTES / 4;                          // floats between the waves' planes
    static constexpr int NPIECE = 2 * MT;                                 // 1 KB DMA pieces per stage
    static constexpr int LX = NPIECE, LY = 2 * NT, GSZ = LX + LY;         // memory operations per stage: {DMA pieces}, {B of k blocks 0, 1}
};

template <int MT, int NT, int NW, int EPI, int D, int CPW>
__device__ __forceinline__ void gemm_kw_body(const GemmArgs &g, const int bx, const int by, const unsigned wg_linear)
{
    using G = KwGeom<MT, NT, NW, D>;
    constexpr int BM = G::BM, BN = G::BN, LDR = G::LDR, NTH = G::NTH, PLANE = G::PLANE;
    constexpr bool ROW_EPI = EPI == EPI_HR || EPI == EPI_RESID_SSQ;
    constexpr bool LSTM = EPI == EPI_LSTM;                // gates: A = [y | h(slot)] in two K segments, four waves = the four chunks, cell epilogue
    static_assert(ROW_EPI || EPI == EPI_BIAS_DSWISH || LSTM, "no GM_KW form of this epilogue");
    static_assert(!LSTM || (NW == 4 && CPW == 1), "the gates GEMM has one slab: a chunk per wave");
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    static_assert(CPW == 1 || CPW == 4, "a wave owns one chunk or one slab");
    constexpr bool DB1 = MT * NT >= 16;                    // 64 x 64 wave tiles: the register-lean form of the K loop (below)
    extern __shared__ __attribute__((aligned(1024))) float red[];
    char *lds = reinterpret_cast<char *>(red);

    if (g.run_flag && gload<int>(g.run_flag) != g.run_gen) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // measurement only (tools/kw_bench built with -DAPRIL_GEMM_TRACE): s_memtime stamps of wave 0 at the phase boundaries
#ifdef APRIL_GEMM_TRACE
    auto stamp = [&](int i) { if (g.trace && wave == 0) g.trace[(size_t)wg_linear * 16 + i] = __builtin_amdgcn_s_memtime(); };
    unsigned long long ph_acc[5] = {0, 0, 0, 0, 0}, ph_t = 0;
    auto lap = [&](int i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph_acc[i] += now - ph_t; ph_t = now; };
#else
    auto stamp = [](int) {};
    auto lap = [](int) {};
    (void)wg_linear;
#endif
    stamp(0);
    const int nt0 = bx * NT, n0 = bx * BN, m0 = by * BM;
    const int KB = g.K >> 4;
    const int c = KB / (4 * g.kz);                        // k blocks per chunk (even: checked on the host)
    constexpr int cpw = CPW;                              // chunks per wave = 4 kz / NW (checked on the host): one chunk, or one slab
    const int T = cpw * c;                                // k blocks of this wave
    const int first_kb = wave * T;
    const int nstage = T >> 1;                            // a multiple of D, >= D (checked on the host)
    const int cs = c >> 1;                                // stages per chunk

    // ---- BasicNorm scale of the residual rows (EPI_HR): partials fetched first thing, reduced through LDS after the K loop
    // (EPI_LSTM: the scale of x = y * scale(y), folded into the sum of the y half: ((p0 + p1) * scale + p2) + p3)
    const RowScale &rsc = LSTM ? g.x_scale : g.r_scale;
    const bool NEED_SCL = (EPI == EPI_HR || LSTM) && rsc.ssq != nullptr;
    float *scl = red + G::LDS_MAIN / 4;
    float stg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int TPR = NTH / BM;
    const int ppt = (rsc.groups + TPR - 1) / TPR;
    const bool staged = NEED_SCL && ppt <= 4 && !(MT == 1 || MT * NT >= 16);      // (the register-lean forms fetch the partials behind the loop; NEED_SCL is EPI_HR only)
    const int srow = threadIdx.x / TPR, sj0 = (threadIdx.x % TPR) * ppt;
    if (NEED_SCL && staged) {
        int r = m0 + srow;
        if (r >= g.M) r = g.M - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ppt && sj0 + k < rsc.groups) stg[k] = gload<float>(rsc.ssq + (size_t)r * rsc.groups + sj0 + k);
    }

    // ---- DMA pieces of this wave: piece i = rows 8 i .. 8 i + 7 of the tile (lane -> row 8 i + (lane >> 3), 16-byte segment
    // lane & 7 of the stage's 128 bytes, swizzled on the SOURCE side: the DMA writes LDS linearly)
    // (uniform 64-bit base that advances with the stage + one 32-bit byte offset per lane and piece: the saddr form of the load;
    // all operands are far below 4 GiB per array)
    uint32_t aoff[G::NPIECE];
    // (EPI_LSTM: waves 0, 1 walk the two chunks of K segment 0 = the y rows, waves 2, 3 those of segment 1 = the h rows of the rows' slots)
    const bool seg1 = LSTM && wave >= 2;                    // uniform
    const char *abase = seg1 ? reinterpret_cast<const char *>(g.a1) + (size_t)(first_kb * 16 - g.K0) * 4 : reinterpret_cast<const char *>(g.a0) + (size_t)first_kb * 64;
    {
        const int *aidx = seg1 ? g.aidx1 : g.aidx0;
        const int lda = seg1 ? g.lda1 : g.lda0;
        int arows[G::NPIECE];
#pragma unroll
        for (int i = 0; i < G::NPIECE; ++i) {
            const int row = m0 + i * 8 + (lane >> 3);
            arows[i] = row >= g.M ? g.M - 1 : row;         // padding rows recompute the last row; never stored
        }
        if (aidx) {
#pragma unroll
            for (int i = 0; i < G::NPIECE; ++i) arows[i] = gload<int>(aidx + arows[i]);
        }
#pragma unroll
        for (int i = 0; i < G::NPIECE; ++i) {
            const int R = i * 8 + (lane >> 3);
            const int gseg = (lane & 7) ^ ((R >> 1) & 7);
            aoff[i] = (uint32_t)((size_t)arows[i] * lda * 4 + gseg * 16);
        }
    }
    char *my = lds + wave * G::WAVE_BYTES;
    auto issue_dma = [&](int slot) {
#ifdef APRIL_KW_ABLATE
        if (g.debug == 4 || g.debug == 6 || g.debug == 7 || g.debug == 9) return;      // measurement: no DMA (stale LDS contents)
#endif
#pragma unroll
        for (int i = 0; i < G::NPIECE; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(abase + aoff[i]),
                                             (__attribute__((address_space(3))) void *)(my + slot * G::STAGE_BYTES + i * 1024), 16, 0, 0);
        }
#ifdef APRIL_KW_ABLATE
        if (g.debug == 3) return;                          // measurement: every stage re-reads the first one (cache-resident operands)
#endif
        abase += 128;
    };
    // weights: uniform base (advances with the k block) + lane offset; tile nt of k block kb = 1 KB at ((nt0 + nt) KB + kb) 1024
    const char *bp = reinterpret_cast<const char *>(g.wp) + ((size_t)nt0 * KB + first_kb) * 1024;
    const size_t bstride = (size_t)KB * 1024;
    const uint32_t boff = (uint32_t)lane * 16u;
    // 64 x 64 wave tiles keep ONE stage of weights in registers (a k block is 64 MFMAs = 2 k cycles: the next block's weights, issued behind
    // the block that frees their registers, have landed by then); smaller tiles a ring of D stages
    constexpr int DB = DB1 ? 1 : D;
    f32x4 be[DB][NT], bo[DB][NT];
#ifdef APRIL_KW_ABLATE
#pragma unroll
    for (int t = 0; t < DB; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { be[t][nt] = f32x4{1.f, 1.f, 1.f, 1.f}; bo[t][nt] = be[t][nt]; }
#endif
    auto load_b = [&](f32x4 (&b)[NT]) {
#ifdef APRIL_KW_ABLATE
        if (g.debug == 6 || g.debug == 7 || g.debug == 8) {      // measurement: no weight loads (the registers stay defined for the compiler)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(b[nt]));
            return;
        }
#endif
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = gload<f32x4>(bp + nt * bstride + boff);
#ifdef APRIL_KW_ABLATE
        if (g.debug == 3) return;
#endif
        bp += 1024;
    };

    // ---- what the epilogue reads besides the sums: fetched before the K loop (as in gemm_body); the 64 x 64 wave tiles have no registers
    // to park it in during the loop (accumulators + slab sums + a stage of weights = 224 of 256) and fetch it behind the loop instead
    constexpr int QROW = BN / 4, NQ = BM * QROW, QPT = (NQ + NTH - 1) / NTH;
    f32x4 e_bias[QPT], e_res[ROW_EPI ? QPT : 1];
    int e_slot[ROW_EPI ? QPT : 1];
    bool e_ok[QPT];
    float *l_cptr[LSTM ? QPT : 1];                        // EPI_LSTM: this thread's (row, hidden unit) cells: slot -> previous cell value, fetched up front
    float l_cprev[LSTM ? QPT : 1];
    auto fetch_epilogue_operands = [&]() {
    if constexpr (LSTM) {
        int qslot[QPT];
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            int r = m0 + q / QROW;
            e_ok[i] = q < NQ && r < g.M;
            if (r >= g.M) r = g.M - 1;
            qslot[i] = gload<int>(g.slot_idx + r);
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int n = q < NQ ? n0 + (q % QROW) * 4 : n0;
            l_cptr[i] = g.c_state + (size_t)qslot[i] * g.hidden + (n >> 2);      // (padding rows point at the last row's cell: read, never stored)
            e_bias[i] = gload<f32x4>(g.bias + n);
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) l_cprev[i] = gload<float>(l_cptr[i]);
        return;
    }
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int q = threadIdx.x + i * NTH;
        int m = m0 + q / QROW;
        e_ok[i] = q < NQ && m < g.M;
        if (m >= g.M) m = g.M - 1;
        const int qn = q < NQ ? n0 + (q % QROW) * 4 : n0;
        e_bias[i] = (EPI != EPI_HR) ? gload<f32x4>(g.bias + qn) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (ROW_EPI) {
            e_slot[i] = (EPI == EPI_HR && g.slot_idx) ? gload<int>(g.slot_idx + m) : m;
            e_res[i] = (EPI == EPI_HR || (EPI == EPI_RESID_SSQ && g.resid)) ? gload<f32x4>(g.resid + (size_t)m * g.ldr + qn) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    };
    // (16-row projection tiles: 80 registers = six waves per SIMD = three workgroups per CU, so that the 768 workgroups of a three-problem
    // launch are ONE round: 17.7 -> 14.7 us; FFN down, twice the K, is operand-issue bound and LOSES with the third workgroup: 22.5 -> 24.5 us)
    constexpr bool LATE_EPI = DB1 || (MT == 1 && EPI == EPI_HR);
    if constexpr (!LATE_EPI) fetch_epilogue_operands();

    // ---- fragment addresses inside a stage buffer: k block p, m tile mt: row 16 mt + mrow, segment (4 p + kq) ^ ((mrow >> 1) & 7)
    const int mrow = lane & 15, kq = lane >> 4;
    int a_rd[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) a_rd[p] = mrow * 128 + (((p * 4 + kq) ^ ((mrow >> 1) & 7)) << 4);
    auto read_frags = [&](int slot, int p, f32x4 (&a)[MT]) {
#ifdef APRIL_KW_ABLATE
        if (g.debug == 6) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(a[mt]));
            return;
        }
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4 *>(my + slot * G::STAGE_BYTES + a_rd[p] + mt * 2048);
    };

    f32x4 acc[MT][NT], S[CPW == 4 ? MT : 1][CPW == 4 ? NT : 1];
#define APRIL_KW_EACH(expr) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { expr; }
    APRIL_KW_EACH(acc[mt][nt] = (f32x4{0.f, 0.f, 0.f, 0.f}))
    int chunk_i = 0, in_chunk = 0;
    auto fold = [&]() {                                   // a chunk chain is complete (CPW == 4: S = ((c0 + c1) + c2) + c3)
        if constexpr (CPW == 4) {
            // one accumulator tile at a time (pinned): left alone, the scheduler computes all sums into fresh registers first, which at 64 x 64
            // per wave (64 + 64 registers) overflows the file
            const bool first = chunk_i == 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    S[mt][nt] = first ? acc[mt][nt] : S[mt][nt] + acc[mt][nt];
                    acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (MT * NT >= 16) __builtin_amdgcn_sched_barrier(0);
                }
            ++chunk_i;
        }
    };
    // k step outermost: consecutive MFMAs go to different accumulators; per accumulator the k steps of a block in order j = 0..3
    // (the canonical chain)
    auto mfma_block = [&](const f32x4 (&a)[MT], const f32x4 (&b)[NT]) {
#ifdef APRIL_KW_ABLATE
        if (g.debug == 5 || g.debug == 8 || g.debug == 9) {      // measurement: no MFMAs (DMA + loads + LDS reads alone); the operands stay live
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" :: "v"(a[mt]));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("" :: "v"(b[nt]));
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
    };

    // ---- K loop.  Stage t = k blocks (2 t, 2 t + 1) of the wave's range: A in ring buffer t % D, B in be / bo [t % D].  Memory
    // operations are issued in the order LX(0) LY(0) .. LX(D-1) LY(D-1) | LX(D) .. LY(D) | ..., LX(t) = {DMA(t)}, LY(t) = {B even(t),
    // B odd(t)}; iteration s issues LX(s + D) behind its first k block (the buffer of stage s has been read: lgkmcnt(0)) and
    // LY(s + D) behind its second, and waits for LX(s + 1) in between -- (D - 1) GSZ younger operations stay in flight (the tail
    // counts down; the weights are register loads, which the compiler's own counts cover).  Nothing here is shared between waves:
    // no barrier.  The stage is [MFMA block][issue][MFMA block][issue] with the two issue halves about equal (FFN down, two problems:
    // 17.0 -> 16.1 us against all issue behind the first block).  g.skew (APRIL_KW_SKEW, default 0) delays the second half of the waves
    // by that many x 64 cycles at the start: measured 0 .. 14, no effect (profiles/r05_kw_phase_trace.txt) -- the two waves of a SIMD
    // do not settle into alternating phases, and what a memory instruction costs beside MFMAs does not depend on which wave issues it
    // (tools/vmem_mfma_probe); kept as a measurement knob.
    f32x4 a0[MT], a1[MT];
#ifdef APRIL_KW_ABLATE
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { a0[mt] = f32x4{1.f, 1.f, 1.f, 1.f}; a1[mt] = a0[mt]; }
#endif
    if constexpr (DB1) {
        // one stage of weights in registers, two stages of activation rows in the ring (D == 2).  Issue order: DMA(0) DMA(1) Be(0) Bo(0) |
        // iteration s: {DMA(s + 2), Be(s + 1)} behind the first k block, {Bo(s + 1)} behind the second; the explicit wait in between is for
        // DMA(s + 1), younger than it: Be(s), Bo(s), DMA(s + 2), Be(s + 1) (the weights are register loads: the compiler's own counts)
        static_assert(D == 2, "two ring stages");
        if (g.debug != 1) {
            issue_dma(0); issue_dma(1); load_b(be[0]); load_b(bo[0]);
            wait_vm<G::NPIECE + 2 * NT>();                 // DMA(0) has landed
            __builtin_amdgcn_sched_barrier(0);
            read_frags(0, 0, a0);
            stamp(1);
            // m tile outermost: a block's fragment of m tile mt is dead after that tile's 16 MFMAs (k steps in order per accumulator, four
            // accumulators in rotation), so the NEXT block's fragment is read into the same registers right behind them: one fragment set
            auto mfma_mt = [&](int mt, const f32x4 (&b)[NT]) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
            };
            auto read_frag1 = [&](int slot, int p, int mt) { a0[mt] = *reinterpret_cast<const f32x4 *>(my + slot * G::STAGE_BYTES + a_rd[p] + mt * 2048); };
            for (int s = 0; s < nstage; ++s) {
                const int slot = s & 1;
                const bool more_a = s + 2 < nstage, more_b = s + 1 < nstage;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    mfma_mt(mt, be[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frag1(slot, 1, mt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                wait_lgkm0();                              // both k blocks of the buffer are in registers
                if (more_a) issue_dma(slot);
                if (more_b) load_b(be[0]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    mfma_mt(mt, bo[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (mt == 0) { if (more_a) wait_vm<G::NPIECE + 3 * NT>(); else if (more_b) wait_vm<3 * NT>(); }      // DMA(s + 1) has landed
                    read_frag1(slot ^ 1, 0, mt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more_b) load_b(bo[0]);
                if constexpr (CPW == 4) { if (++in_chunk == cs) { in_chunk = 0; fold(); } }
            }
        }
    } else
    if (g.debug != 1) {
#pragma unroll
        for (int t = 0; t < D; ++t) { issue_dma(t); load_b(be[t]); load_b(bo[t]); }
        wait_vm<(D - 1) * G::GSZ + G::LY>();               // LX(0) has landed
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, 0, a0);
        if (wave >= NW / 2) for (int i = 0; i < g.skew; ++i) __builtin_amdgcn_s_sleep(1);
        stamp(1);
        auto stage = [&](const int slot, const bool more, auto wait_next) {
            read_frags(slot, 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(a0, be[slot]);
            __builtin_amdgcn_sched_barrier(0);
            lap(0);
            wait_lgkm0();                                  // both k blocks of the buffer are in registers
            lap(1);
            if (more) issue_dma(slot);
            lap(2);
            wait_next();                                   // LX(s + 1) has landed
            lap(3);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(slot + 1 == D ? 0 : slot + 1, 0, a0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(a1, bo[slot]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) { load_b(be[slot]); load_b(bo[slot]); }
            if constexpr (CPW == 4) { if (++in_chunk == cs) { in_chunk = 0; fold(); } }
            lap(4);
        };
#ifdef APRIL_GEMM_TRACE
        ph_t = __builtin_amdgcn_s_memtime();
#endif
        int s = 0;
        for (; s + D < nstage; s += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) stage(j, true, [] { wait_vm<(D - 1) * G::GSZ>(); });
        }
        // tail: stages nstage - D .. nstage - 1 (slots 0 .. D - 1), nothing left to issue; younger than LX(s + 1): LY(s + 1) and the
        // whole groups of stages s + 2 .. nstage - 1
        static_assert(D == 2 || D == 4, "ring depth 2 or 4");
        if constexpr (D == 4) {
            stage(0, false, [] { wait_vm<G::LY + 2 * G::GSZ>(); });
            stage(1, false, [] { wait_vm<G::LY + G::GSZ>(); });
            stage(2, false, [] { wait_vm<G::LY>(); });
            stage(3, false, [] {});
        } else {
            stage(0, false, [] { wait_vm<G::LY>(); });
            stage(1, false, [] {});
        }
    }
    if constexpr (LATE_EPI) fetch_epilogue_operands();
    stamp(2);
#ifdef APRIL_GEMM_TRACE
    if (g.trace && lane == 0 && (wave == 0 || wave == 4)) {      // phase sums of wave 0 and of wave 4 (the younger wave of the same SIMD, as a rule)
        for (int i = 0; i < 5; ++i) g.trace[(size_t)wg_linear * 16 + (wave == 0 ? 6 : 11) + i] = ph_acc[i];
    }
    if (g.trace && lane == 0) {                            // when the LAST wave of the workgroup left its K loop
        atomicMax(&g.trace[(size_t)wg_linear * 16 + 5], (unsigned long long)__builtin_amdgcn_s_memtime());
    }
#endif

    // ---- the waves meet: every wave parks its result in its own region (its ring is drained: no DMA in flight, all reads done)
    {
        float *mine = red + (size_t)wave * PLANE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (CPW == 4) mine[(mt * 16 + kq * 4 + r) * LDR + nt * 16 + mrow] = S[mt][nt][r];
                    else mine[(mt * 16 + kq * 4 + r) * LDR + nt * 16 + mrow] = acc[mt][nt][r];
                }
    }
    if (NEED_SCL) {
        // the rows' scales: partials (in registers since the first instruction of the kernel) -> LDS, BM threads add them in column
        // order (the order of row_scale()); rows are padded to G + 1 floats
        const int Gn = rsc.groups;
        float *part = scl + BM;
        if (staged) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < ppt && sj0 + k < Gn) part[srow * (Gn + 1) + sj0 + k] = stg[k];
        } else {
            for (int i = threadIdx.x; i < BM * Gn; i += NTH) {
                int r = m0 + i / Gn;
                if (r >= g.M) r = g.M - 1;
                part[(i / Gn) * (Gn + 1) + i % Gn] = gload<float>(rsc.ssq + (size_t)r * Gn + i % Gn);
            }
        }
    }
    __syncthreads();
    if (NEED_SCL) {
        if (threadIdx.x < BM) {
            const int Gn = rsc.groups;
            const float *part = scl + BM;
            float t = 0.0f;
            for (int j = 0; j < Gn; ++j) t += part[threadIdx.x * (Gn + 1) + j];
            scl[threadIdx.x] = __builtin_amdgcn_rsqf(t * rsc.inv_n + rsc.eps);
        }
        __syncthreads();
    }
    stamp(3);
    auto summed4 = [&](int o) {
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + o), p1 = *reinterpret_cast<const f32x4 *>(red + PLANE + o);
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(red + 2 * PLANE + o), p3 = *reinterpret_cast<const f32x4 *>(red + 3 * PLANE + o);
        if constexpr (NW == 8) {
            const f32x4 p4 = *reinterpret_cast<const f32x4 *>(red + 4 * PLANE + o), p5 = *reinterpret_cast<const f32x4 *>(red + 5 * PLANE + o);
            const f32x4 p6 = *reinterpret_cast<const f32x4 *>(red + 6 * PLANE + o), p7 = *reinterpret_cast<const f32x4 *>(red + 7 * PLANE + o);
            if (cpw >= 4) return ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));        // eight tree nodes
            return (((p0 + p1) + p2) + p3) + (((p4 + p5) + p6) + p7);                      // two slabs of four chunks
        } else {
            if (cpw >= 4) return (p0 + p1) + (p2 + p3);                                     // four tree nodes
            return ((p0 + p1) + p2) + p3;                                                   // one slab of four chunks
        }
    };
    if constexpr (LSTM) {
        // every load this epilogue depends on was issued before the K loop: settle them once, so that nothing in the loop below waits on
        // the memory counter while the previous quad's stores are in flight (gemm_body)
        wait_vm<0>();
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int row = q / QROW, col = (q % QROW) * 4;
            const int o = row * LDR + col;
            f32x4 gt = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < NQ) {
                const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + o), p1 = *reinterpret_cast<const f32x4 *>(red + PLANE + o);
                const f32x4 p2 = *reinterpret_cast<const f32x4 *>(red + 2 * PLANE + o), p3 = *reinterpret_cast<const f32x4 *>(red + 3 * PLANE + o);
                // x = y * scale(y) entered the GEMM as y: the y half of the sum takes the row's scale; without a scale the plain slab
                if (NEED_SCL) gt = (((p0 + p1) * scl[row] + p2) + p3) + e_bias[i];
                else gt = (((p0 + p1) + p2) + p3) + e_bias[i];
            }
            const float c_new = fast_sigmoid(gt.y) * l_cprev[i] + fast_sigmoid(gt.x) * fast_tanh(gt.z);
            const float u = fast_sigmoid(gt.w) * fast_tanh(c_new);
            if (e_ok[i]) { gstore<float>(l_cptr[i], c_new); gstore<float>(g.out + (size_t)(m0 + row) * g.ldo + ((n0 + col) >> 2), u); }
        }
    } else
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int q = threadIdx.x + i * NTH;
        const int row = q / QROW, col = (q % QROW) * 4;
        const int m = m0 + row, n = n0 + col;
        const f32x4 v = q < NQ ? summed4(row * LDR + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_HR) {
            if (e_ok[i]) {
                const float rs = scl[row];
                gstore<f32x4>(g.state + (size_t)e_slot[i] * g.ld_state + n, v);
                gstore<f32x4>(g.out + (size_t)m * g.ldo + n, e_res[i] * rs + v);
            }
        } else if (EPI == EPI_RESID_SSQ) {
            const bool ok = e_ok[i];
            f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                y = v + e_bias[i];
                if (g.resid) y = e_res[i] + y;
                gstore<f32x4>(g.out + (size_t)m * g.ldo + n, y);
            }
            const float ss = granule_ssq(y);               // all lanes take part in the shuffles
            if (ok && (q & 7) == 0) gstore<float>(g.ssq_out + (size_t)m * (g.N / SSQ_COLS) + n / SSQ_COLS, ss);
        } else {   // EPI_BIAS_DSWISH
            if (e_ok[i]) {
                const f32x4 y = v + e_bias[i];
                f32x4 o;
                o.x = y.x * fast_sigmoid(y.x - 1.0f); o.y = y.y * fast_sigmoid(y.y - 1.0f);
                o.z = y.z * fast_sigmoid(y.z - 1.0f); o.w = y.w * fast_sigmoid(y.w - 1.0f);
                gstore<f32x4>(g.out + (size_t)m * g.ldo + n, o);
            }
        }
    }
    stamp(4);
#undef APRIL_KW_EACH
}

// waves per SIMD the register budget must allow: one eight-wave workgroup per CU (the 128 registers of two spill inside the K loop) -- three for the
// 16-row projection tiles --, three four-wave ones
// Which tile a workgroup takes.  Workgroups go to the eight XCDs round robin in dispatch order, so with the plain mapping (column tile =
// blockIdx.x) an XCD owns two of the sixteen column tiles of an N = 512 problem and ALL of its rows: the weights cross the fabric once,
// the activation rows eight times.  xcd_rc = 2 deals the tiles as 2 row halves x 4 column quarters instead: XCD x = 4 r + c
// takes rows of half r and columns of quarter c, the j-th workgroup it receives walks its sub-block column-fastest -- weights twice, rows
// four times across the fabric.  Needs grid.x % 4 == 0 and grid.y % 2 == 0 (then every problem of a z-batched launch starts at XCD 0).
__device__ __forceinline__ void kw_tile_of(const int xcd_rc, int &bx, int &by)
{
    const int gx = (int)gridDim.x, gy = (int)gridDim.y;
    bx = (int)blockIdx.x; by = (int)blockIdx.y;
    if (xcd_rc == 2 && (gx & 3) == 0 && (gy & 1) == 0) {
        const int L = bx + gx * by;
        const int x = L & 7, j = L >> 3, qx = gx >> 2, hy = gy >> 1;
        by = (x >> 2) * hy + j / qx;
        bx = (x & 3) * qx + j % qx;
    }
}

template <int MT, int NT, int NW, int EPI, int D, int CPW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? ((MT == 1 && EPI == EPI_HR) ? 6 : 2) : (MT == 4 ? 2 : 3)) void gemm_kw_kernel(GemmArgs g)
{
    int bx, by;
    kw_tile_of(g.xcd_rc, bx, by);
    gemm_kw_body<MT, NT, NW, EPI, D, CPW>(g, bx, by, blockIdx.x + gridDim.x * blockIdx.y);
}

// n independent same-shape problems in one launch (see gemm_f32_zkernel): blockIdx.z picks the argument block
template <int MT, int NT, int NW, int EPI, int D, int CPW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? ((MT == 1 && EPI == EPI_HR) ? 6 : 2) : (MT == 4 ? 2 : 3)) void gemm_kw_zkernel(const GemmArgs *__restrict__ zargs)
{
    const GemmArgs g = zargs[blockIdx.z];
    int bx, by;
    kw_tile_of(g.xcd_rc, bx, by);
    gemm_kw_body<MT, NT, NW, EPI, D, CPW>(g, bx, by, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// Mixed tiles for a z-batched launch whose 32-row tiles are not a whole number of rounds (three 256-row problems at N = 512: 384 tiles of
// 32 x 32 = one and a half per CU; as 16-row tiles 768 = three per CU, the planner's choice so far): the first n - 1 problems on 32-row tiles,
// the last one on 16-row tiles -- 256 + 256 workgroups, every CU one of each (a 16-row round costs ~0.55 of a 32-row one: 1.55 against
// 1.65).  One-dimensional grid, the 32-row tiles first.  Same bodies, same arguments per tile: the sums do not depend on the tile shape.
// MEASUREMENT FORM, off (APRIL_KW_MIXED=1): bit-identical, but 1.334-1.345 against 1.320 ms per 256-session step -- the 16-row workgroups of the
// plain launch run up to six per CU and hide each other's latencies, two per CU beside a 32-row one do not (the FFN-up twin of this form pays: kernels_gemm.hip).
template <int NT, int NW, int EPI, int D, int CPW>
__global__ __launch_bounds__(64 * NW, 2) void gemm_kw_zkernel_mixed(const GemmArgs *__restrict__ zargs, int gx, int gy32, int nbig_problems)
{
    const int nbig = gx * gy32 * nbig_problems;
    int tile = (int)blockIdx.x;
    if (tile < nbig) {
        const int bx = tile % gx, r = tile / gx, by = r % gy32, z = r / gy32;
        const GemmArgs g = zargs[z];
        gemm_kw_body<2, NT, NW, EPI, D, CPW>(g, bx, by, (unsigned)tile);
    } else {
        tile -= nbig;
        const GemmArgs g = zargs[nbig_problems];
        gemm_kw_body<1, NT, NW, EPI, D, CPW>(g, tile % gx, tile / gx, (unsigned)(nbig + tile));
    }
}

// the mixed form applies (launch_kw_one<1, ...> asks): n problems whose 32-row tiles leave a partial round while n - 1 of them and the
// last one's 16-row tiles are whole rounds
template <int NT, int NW, int EPI, int D, int CPW>
bool launch_kw_mixed(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    static const int mixed = [] { const char *v = getenv("APRIL_KW_MIXED"); return v && *v ? atoi(v) : 0; }();
    if (!mixed || !dev_args || n < 2 || g.xcd_rc != 0 || g.M % 32 != 0) return false;
    using G2 = KwGeom<2, NT, NW, D>;
    using G1 = KwGeom<1, NT, NW, D>;
    const long gx = g.N / G2::BN, gy32 = g.M / 32, t32 = gx * gy32, t16 = 2 * t32;
    if ((t32 * n) % 256 == 0 || (t32 * (n - 1)) % 256 != 0 || t16 % 256 != 0) return false;
    const int sg = (EPI == EPI_HR && g.r_scale.ssq) ? g.r_scale.groups : 0;
    const size_t lds2 = (size_t)G2::LDS_MAIN + (size_t)(G2::BM + (sg ? G2::BM * (sg + 1) : 0)) * sizeof(float);
    const size_t lds1 = (size_t)G1::LDS_MAIN + (size_t)(G1::BM + (sg ? G1::BM * (sg + 1) : 0)) * sizeof(float);
    const size_t lds = lds2 > lds1 ? lds2 : lds1;
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kw_zkernel_mixed<NT, NW, EPI, D, CPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((gemm_kw_zkernel_mixed<NT, NW, EPI, D, CPW>), dim3((unsigned)(t32 * (n - 1) + t16)), dim3(G2::NTH), lds, s, dev_args, (int)gx, (int)gy32, n - 1);
    return true;
}

template <int MT, int NT, int NW, int EPI, int D, int CPW>
void launch_kw_one(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    if constexpr (MT == 1 && NW == 8 && NT == 2 && (EPI == EPI_HR || EPI == EPI_RESID_SSQ)) {
        if (launch_kw_mixed<NT, NW, EPI, D, CPW>(g, dev_args, n, s)) return;
    }
    using G = KwGeom<MT, NT, NW, D>;
    dim3 grid((unsigned)(g.N / G::BN), (unsigned)((g.M + G::BM - 1) / G::BM), (unsigned)std::max(1, n));
    const int sg = (EPI == EPI_HR && g.r_scale.ssq) ? g.r_scale.groups : ((EPI == EPI_LSTM && g.x_scale.ssq) ? g.x_scale.groups : 0);
    const size_t lds = (size_t)G::LDS_MAIN + (size_t)(G::BM + (sg ? G::BM * (sg + 1) : 0)) * sizeof(float);
    // dynamic LDS beyond 64 KB has to be announced, per instantiation AND per device (see kernels_gemm_tile.hip)
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kw_kernel<MT, NT, NW, EPI, D, CPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kw_zkernel<MT, NT, NW, EPI, D, CPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    if (dev_args) hipLaunchKernelGGL((gemm_kw_zkernel<MT, NT, NW, EPI, D, CPW>), grid, dim3(G::NTH), lds, s, dev_args);
    else hipLaunchKernelGGL((gemm_kw_kernel<MT, NT, NW, EPI, D, CPW>), grid, dim3(G::NTH), lds, s, g);
}

template <int MT, int NT, int NW, int D>
bool dispatch_kw(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s, bool dry)
{
    const int cpw = 4 * g.kz / NW;
    if constexpr (NW == 8) {
        if (g.epi == EPI_HR && cpw == 4) { if (!dry) launch_kw_one<MT, NT, NW, EPI_HR, D, 4>(g, dev_args, n, s); return true; }
        if (g.epi == EPI_HR && cpw == 1) { if (!dry) launch_kw_one<MT, NT, NW, EPI_HR, D, 1>(g, dev_args, n, s); return true; }
        if (g.epi == EPI_RESID_SSQ && cpw == 4) { if (!dry) launch_kw_one<MT, NT, NW, EPI_RESID_SSQ, D, 4>(g, dev_args, n, s); return true; }
        if (g.epi == EPI_RESID_SSQ && cpw == 1) { if (!dry) launch_kw_one<MT, NT, NW, EPI_RESID_SSQ, D, 1>(g, dev_args, n, s); return true; }
    } else {
        if (g.epi == EPI_BIAS_DSWISH && cpw == 1) { if (!dry) launch_kw_one<MT, NT, NW, EPI_BIAS_DSWISH, D, 1>(g, dev_args, n, s); return true; }
        if constexpr (NT == 2) { if (g.epi == EPI_LSTM && cpw == 1) { if (!dry) launch_kw_one<MT, NT, NW, EPI_LSTM, D, 1>(g, dev_args, n, s); return true; } }
    }
    return false;
}

}  // namespace

// GM_KW eligibility of a GEMM (shape + operands): the host-side rule launch_gemm's planner and the engine share.  Returns the
// number of waves (8 or 4), or 0.
int gemm_kw_waves(const GemmArgs &g)
{
    if (g.wt != 0 || g.a_op != AOP_NONE || g.wave_mask != 0xF || g.p_add || g.N % 32 != 0 || g.K % 64 != 0) return 0;
    if (g.epi == EPI_LSTM) {
        // the one-launch gates GEMM of a chunk step: [y | h(slot)] in two equal K segments, one slab, the cell epilogue
        if (g.kz != 1 || g.K1 != g.K0 || g.K0 * 2 != g.K || !g.a1 || !g.slot_idx || !g.c_state || !g.out || g.out16) return 0;
        const int c = g.K / 16 / 4;
        if (c % 2 != 0 || (c / 2) % 2 != 0) return 0;       // chunks of whole stages, whole rounds of the two-stage ring
        return 4;
    }
    if (g.K1 != 0) return 0;
    if (g.epi != EPI_HR && g.epi != EPI_RESID_SSQ && g.epi != EPI_BIAS_DSWISH) return 0;
    if (g.epi == EPI_BIAS_DSWISH && g.x_scale.ssq) return 0;
    const int KB = g.K / 16;
    if (KB % (4 * g.kz) != 0) return 0;
    const int c = KB / (4 * g.kz);
    if (c % 2 != 0) return 0;                              // chunks of whole stages
    int nw = 0;
    if (g.epi == EPI_BIAS_DSWISH) nw = g.kz == 1 ? 4 : 0;  // one chunk per wave
    else if (g.kz == 8 || g.kz == 2) nw = 8;               // one slab / one chunk per wave
    if (!nw) return 0;
    const int nstage = (4 * g.kz / nw) * c / 2;
    if (nstage < 2 || nstage % 2 != 0) return 0;           // ring depth 2 (4 where nstage allows): whole rounds of the ring (the 64 x 64 form takes any nstage >= 2)
    return nw;
}

// launch of a GEMM whose plan chose GM_KW: tile 16 mt x 16 nt, g.zs == g.kz
// the kernel of (g, tile): launched, or (dry) only looked up -- ONE table for plan_kw's question and for the launch
static bool kw_find(const GemmArgs &g, int mt, int nt, const GemmArgs *dev_args, int n, hipStream_t s, bool dry)
{
    const int nw = gemm_kw_waves(g);
    bool ok = false;
    static const int ring = [] { const char *v = getenv("APRIL_KW_RING"); return v && *v ? atoi(v) : 2; }();      // ring stages per wave: 2; 4 measured the same (tools/kw_bench: the loads are never waited for) at twice the LDS
    const int nstage = nw ? (4 * g.kz / nw) * (g.K / 16 / (4 * g.kz)) / 2 : 0;
    if (nw == 8 && ring == 4 && nstage % 4 == 0 && mt == 2 && nt == 2) ok = dispatch_kw<2, 2, 8, 4>(g, dev_args, n, s, dry);
    else if (nw == 8) {
        if (mt == 4 && nt == 4) ok = dispatch_kw<4, 4, 8, 2>(g, dev_args, n, s, dry);
        else if (mt == 2 && nt == 2) ok = dispatch_kw<2, 2, 8, 2>(g, dev_args, n, s, dry);
        else if (mt == 1 && nt == 2) ok = dispatch_kw<1, 2, 8, 2>(g, dev_args, n, s, dry);
    } else if (nw == 4) {
        if (g.epi == EPI_LSTM) {
            if (mt == 4 && nt == 2) ok = dispatch_kw<4, 2, 4, 2>(g, dev_args, n, s, dry);
            else if (mt == 2 && nt == 2) ok = dispatch_kw<2, 2, 4, 2>(g, dev_args, n, s, dry);
        }
        else if (mt == 2 && nt == 4) ok = dispatch_kw<2, 4, 4, 2>(g, dev_args, n, s, dry);
    }
    return ok;
}

// is there a GM_KW kernel for this GEMM on 16 mt x 16 nt tiles?  (plan_kw asks before it commits to the schedule: pinned tile shapes and
// environment knobs must never pick a form that was not instantiated)
bool gemm_kw_has_kernel(const GemmArgs &g, int mt, int nt) { return kw_find(g, mt, nt, nullptr, 0, nullptr, true); }

void launch_gemm_kw(const GemmArgs &g, int mt, int nt, const GemmArgs *dev_args, int n, hipStream_t s)
{
    if (!kw_find(g, mt, nt, dev_args, n, s, false)) {
        fprintf(stderr, "libapril(mi355x): launch_gemm_kw: no kernel for epi %d tile %d x %d, %d waves\n", g.epi, 16 * mt, 16 * nt, gemm_kw_waves(g));
        abort();
    }
}

}  // namespace aprilx
