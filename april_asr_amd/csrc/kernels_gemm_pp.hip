// GM_PP schedule of the binary16-operand MFMA GEMM (kernels.h) for gfx950 -- round 6, BASELINE configs[4] (fp16 MFMA path).
//
// What it replaces: the eight-wave 128 x 128 form of GM_TILE (kernels_gemm_tile.hip) ran the fp16 gates GEMM at 0.17 of the
// dense fp16 MFMA peak: every wave issued its DMA pieces, its fragment reads and its MFMAs in one in-order stream, in lock step
// with the seven others (one barrier per stage), so whatever sat between two MFMA blocks ran while the matrix pipe drained (a
// 64 x 32 wave tile: 6 KB of fragment reads and 2 DMA instructions per 8 MFMAs = 128 cycles of matrix work).
//
// Schedule: one workgroup = eight waves = TWO GROUPS of four (waves 0-3 / 4-7: one wave of each group per SIMD).  The tile is
// 16 MT rows x 128 columns; group g owns rows [g BM/2, (g+1) BM/2), its four waves split that half 2 x 2 -- at MT = 16 a wave
// tile is 64 x 64 = 16 v_mfma_f32_16x16x32_f16 per k block against 8 fragment reads (the densest ratio the 16 x 16 shape
// allows).  The groups run ONE PHASE APART (ping-pong): while group 0 is in the compute phase of k block j -- its 16 MFMAs with
// the fragment reads of k block j + 1 spread between them, into a second register set -- group 1 is in its load phase (the DMA
// pieces of a later stage: scalar instructions and buffer_load ... lds only), then they swap:
//
//     barrier index   0      1      2      3      4      5     ...   2n+1
//     group 0         | L0   | C0   | L1   | C1   | L2   | ...  C(n-1) |  -   |
//     group 1         | -    | L0   | C0   | L1   | C1   | ...  L(n-1) | C(n-1)
//
// so a SIMD's two waves are never in the same kind of phase: one wave's memory instructions issue beside the other's MFMAs
// instead of in front of its own.  Both groups share the weight pieces of a stage (one B image per 256 rows: 85 flop per operand
// byte at MT = 16, 64 at MT = 8).  What made room for the second fragment set is the fp16 one-chain rule (kernels.h): the
// accumulator is the only persistent register set.
//
// Measured (tools/pp_bench, s_memtime phase trace of the 256-row tile, cycles per k block and wave): load phase 280 + 210..260 at its
// barrier, compute phase 490..530 + 25..40 -- i.e. the phases are now set by the compute phase, whose 16 MFMAs take ~24 cycles each
// (~19 is the instruction's floor) plus ~12 per interleaved ds_read_b128.  Variants measured and not kept: all eight reads in the load
// phase (load 416 > compute 382: 29.2 vs 28.3 us per 256-row tile round), s_setprio around either phase (no effect), k-block-major
// weight addressing (an L2 channel experiment: no effect), the fragment reads + MFMAs of a k block as one phase per group without
// the second register set (first form: 1376 cycles per k block).  The prologue (~4 us: the first two stages come from HBM) and the
// LSTM epilogue (~5 us at MT = 16: 16 cells per thread) are a third of a tile's time and run on an otherwise empty CU (144 KB of
// stage buffers: one workgroup per CU).
//
// LDS image of one stage (two k blocks = 64 k = 128 bytes per activation row), as GM_TILE's:
//   A: [BM rows][128 B], 16-byte segment g of row R stored at segment g ^ ((R >> 1) & 7) (conflict-free ds_read_b128 of the A
//      fragment; the DMA writes LDS linearly, so the swizzle is applied to the per-lane SOURCE address)
//   B: [2 k blocks][8 n tiles][1 KB] in the packed weight order (launch_repack_x32) = the B fragment of every lane.
// Three stage buffers.  The pieces of stage s + 2 are issued during the two load phases of stage s (a wave's share: PPW / 2 per
// load phase), into the buffer stage s - 1 was read from -- its last reads (the compute phase of k block 2 s - 2, group 1 last)
// were retired by lgkmcnt(0) before barrier 4 s - 1, and the first of those DMA instructions issues behind barrier 4 s.  Stage
// s + 1 is waited for (counted vmcnt: younger pieces stay in flight) by group 0 at the end of the load phase of k block 2 s + 1 and
// by group 1 at the end of the compute phase of k block 2 s -- both in front of barrier 4 s + 3, behind which group 0 reads it.
// DMA pieces cost no vector ALU instruction: buffer descriptors in SGPRs, 32-bit lane offsets fixed per tile, scalar stage offsets.
//
// Summation: the fp16 one-chain rule (kernels.h) -- one MFMA chain over all k blocks in k order, multiplied by the row's BasicNorm
// scale where the y half of K ends (EPI_LSTM / EPI_XPART); the layer-major h half continues the chain from P.  GM_TILE's fp16 forms
// of the same GEMMs compute exactly this; tools/pp_bench compares every output of the two bitwise (tests/test_gpu_gemm_pp.py).
// Epilogues: EPI_LSTM (two A segments [y16 | h16(slot)], cell update), EPI_BIAS_DSWISH, and the layer-major halves of the gate
// GEMM (EPI_XPART + wave_mask 0x3, EPI_LSTM + wave_mask 0xC + p_add).  Every other GEMM keeps GM_TILE.
// Replaces the ORT MatMul nodes of the encoder's LSTM / FFN blocks (reference call site src/april_session.c:131-148).
#include "kernels.h"
#include "device_utils.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace aprilx {

namespace {

using h4 = __attribute__((ext_vector_type(4))) _Float16;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ h4 to_h4(const f32x4 &v) { return h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; }

template <int MT> struct PPGeom {
    static constexpr int NT = 8, NS = 3, NW = 8, NTH = 64 * NW;
    static constexpr int BM = 16 * MT, BN = 16 * NT, LDR = BN + 4;
    static constexpr int A_BYTES = BM * 128, B_BYTES = 2 * NT * 1024, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int APW = 2 * MT / NW;                       // activation pieces (8 rows x 128 B) per wave and stage
    static constexpr int PPW = APW + 2;                           // + this wave's weight piece of either k block
    static constexpr int MTW = MT / 4, NTW = NT / 2;              // MFMA tiles per wave: (BM / 2 / 2 / 16) x (BN / 2 / 16)
    static constexpr int PLANE_BYTES = BM * LDR * 4;
    static constexpr int LDS_MAIN = NS * STAGE_BYTES > PLANE_BYTES ? NS * STAGE_BYTES : PLANE_BYTES;
    static constexpr int LDS_BYTES = LDS_MAIN + BM * 4;           // + the rows' BasicNorm scales
    static_assert(MT == 16 || MT == 8, "tile rows 256 or 128");
    static_assert(PPW % 2 == 0, "a wave issues half of its pieces in either load phase of a stage");
};

template <int N> __device__ __forceinline__ void wait_vm()
{
    // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14; expcnt / lgkmcnt at their maxima)
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }      // s_waitcnt lgkmcnt(0) only

// Every pointer of a GemmArgs block is global memory.  The z-batched entry point reads the block from memory, so the compiler only
// knows generic pointers and emits FLAT loads / stores -- and a pending flat operation forces s_waitcnt vmcnt(0) lgkmcnt(0) at the next
// wait (it may be served by either path), which would drain the DMA stages the prologue has just put in flight.  gp() says "global".
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T *gp(T *p) { return (__attribute__((address_space(1))) T *)p; }

template <int MT, int EPI>
__device__ __forceinline__ void gemm_pp_body(const GemmArgs &g)
{
    using G = PPGeom<MT>;
    constexpr int BM = G::BM, BN = G::BN, NT = G::NT, MTW = G::MTW, NTW = G::NTW, NTH = G::NTH, LDR = G::LDR, PPW = G::PPW, APW = G::APW, HP = PPW / 2;
    static_assert(EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH || EPI == EPI_XPART, "GM_PP: the kz = 1 GEMMs of a layer (gates, FFN up)");
    extern __shared__ __attribute__((aligned(1024))) float red[];
    char *lds = reinterpret_cast<char *>(red);

    if (g.run_flag && *gp(g.run_flag) != g.run_gen) return;
#ifdef APRIL_GEMM_TRACE
    const unsigned long long tr_start = __builtin_amdgcn_s_memtime();
    unsigned long long tr_loop0 = tr_start, tr_loop1 = tr_start;
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int wrow = ((grp * 2 + wm) * MTW) * 16;         // first tile row of this wave
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int KB = g.K / 32;
    const int c = KB / 4;                                  // k blocks per chunk (kz = 1)
    const bool half_x = g.wave_mask == 0x3, half_h = EPI == EPI_LSTM && g.wave_mask == 0xC;
    const int T = (half_x || half_h) ? 2 * c : 4 * c;      // k blocks of this workgroup (even: checked on the host)
    const int first_kb = half_h ? 2 * c : 0;
    const int nstage = T >> 1;

    // ---- BasicNorm scales of the tile's rows: the sum-of-squares partials make one trip from global memory at kernel start
    const RowScale &rsc = g.x_scale;
    const bool fold_scale = (EPI == EPI_LSTM || EPI == EPI_XPART) && rsc.ssq != nullptr;
    float *scl = red + G::LDS_MAIN / 4;
    constexpr int TPR = NTH / BM, STG = 16;
    float stg[STG];
    const int ppt = (rsc.groups + TPR - 1) / TPR;
    const int srow = threadIdx.x / TPR, sj0 = (threadIdx.x % TPR) * ppt;
    if (fold_scale) {
        int r = m0 + srow;
        if (r >= g.M) r = g.M - 1;
#pragma unroll
        for (int k = 0; k < STG; ++k) stg[k] = (k < ppt && sj0 + k < rsc.groups) ? gp(rsc.ssq)[(size_t)r * rsc.groups + sj0 + k] : 0.0f;
    }

    // ---- the epilogue's thread -> cell map (thread t owns the 4-column quad t % 32 of rows t / 32 + 16 i) and, for EPI_LSTM, the rows'
    // slots: loaded FIRST, in front of every DMA piece
    constexpr int QROW = BN / 4, NQ = BM * QROW, QPT = NQ / NTH, QRS = NTH / QROW;
    static_assert(QROW == 32 && NQ % NTH == 0, "quad layout of the epilogue");
    const int qcol = threadIdx.x % QROW, qrow0 = threadIdx.x / QROW;
    const int ncol = n0 + qcol * 4, l_unit = ncol >> 2;
    int l_slot[EPI == EPI_LSTM ? QPT : 1];
    if (EPI == EPI_LSTM) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            int r = m0 + qrow0 + QRS * i;
            if (r >= g.M) r = g.M - 1;                    // (padding rows read the last row's cell: never stored)
            l_slot[i] = gp(g.slot_idx)[r];
        }
    }

    // ---- DMA pieces of this wave.  i < APW: activation rows 8 P .. 8 P + 7 of the tile, P = wave + 8 i (lane -> row P 8 + (lane >> 3),
    // 16-byte segment lane & 7 of the stage's 128 bytes, swizzled); i = APW, APW + 1: the weight piece (k block 0 / 1, n tile `wave`).
    // Sources are wave-uniform bases + 32-bit lane offsets; the activations may come in two K segments (gates: [y | h(slot)], K0 a
    // multiple of the stage's 64 k): stage `seg1_stage` of this workgroup is the first of segment 1.
    const int k_begin = first_kb * 32;
    const bool two_seg = g.K1 > 0;
    const bool start_in_1 = two_seg && k_begin >= g.K0;
    const int seg1_stage = (two_seg && !start_in_1) ? (g.K0 - k_begin) / 64 : 0x7fffffff;
    // (segment 1's rows are slots: the indirection is LOADED here and turned into offsets where it is first needed -- the switch of
    // the pieces to segment 1, deep inside the K loop -- so that the first DMA stages do not wait for it)
    unsigned aoff0[APW];
    int arows1[APW];
    unsigned gsegv[APW];
    {
        int arows[APW];
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            int row = m0 + (wave + 8 * i) * 8 + (lane >> 3);
            arows[i] = row >= g.M ? g.M - 1 : row;       // padding rows recompute the last row; never stored
            arows1[i] = arows[i];
        }
        if (g.aidx0) {
#pragma unroll
            for (int i = 0; i < APW; ++i) arows[i] = gp(g.aidx0)[arows[i]];
        }
        if (two_seg && g.aidx1) {
#pragma unroll
            for (int i = 0; i < APW; ++i) arows1[i] = gp(g.aidx1)[arows1[i]];
        }
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int R = (wave + 8 * i) * 8 + (lane >> 3);
            gsegv[i] = (unsigned)(((lane & 7) ^ ((R >> 1) & 7)) * 16);
            aoff0[i] = (unsigned)arows[i] * (unsigned)g.lda0 * 2u + gsegv[i];
        }
    }
    // buffer descriptors (wave-uniform, SGPRs) + 32-bit lane offsets + scalar stage offsets: a DMA piece costs no vector ALU
    // instruction (buffer_load_dwordx4 v, s[rsrc], s_off offen lds; m0 = the LDS row of the piece)
    const char *abase0 = reinterpret_cast<const char *>(g.a0) + (size_t)(start_in_1 ? 0 : k_begin) * 2;
    const char *abase1 = reinterpret_cast<const char *>(g.a1) + (size_t)(start_in_1 ? k_begin - g.K0 : 0) * 2;
    const char *wbase = reinterpret_cast<const char *>(g.wp) + ((size_t)(blockIdx.x * NT + wave) * KB + first_kb) * 1024;
    constexpr int RSRC_FLAGS = 0x00020000;                 // raw buffer, 32-bit data format (gfx9 dword 3)
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(abase0), 0, -1, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(two_seg ? abase1 : abase0), 0, -1, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wbase), 0, -1, RSRC_FLAGS);
    const int woff = lane * 16;
    int issued = 0;                                        // stages issued so far
    int seg_base = 0;                                      // first stage of the segment being issued
    __amdgpu_buffer_rsrc_t rs_a = start_in_1 ? rs_a1 : rs_a0;
    unsigned aoff[APW];
    if (start_in_1) {                                      // (a branch, not a select: segment 0's offsets do not wait for the slot indirection of segment 1)
#pragma unroll
        for (int i = 0; i < APW; ++i) aoff[i] = (unsigned)arows1[i] * (unsigned)g.lda1 * 2u + gsegv[i];
    } else {
#pragma unroll
        for (int i = 0; i < APW; ++i) aoff[i] = aoff0[i];
    }
#ifdef APRIL_GEMM_TRACE
    const int dbg = g.debug;                               // measurement builds (tools/pp_bench_trace, APRIL_GEMM_DEBUG): 8 = no DMA (stale LDS), 10 = no MFMAs
#else
    constexpr int dbg = 0;
#endif
    auto issue_half = [&](int p, int buf) {                // half p (0 / 1) of this wave's pieces of stage `issued`, into buffer buf
        if (dbg == 8) { if (p == 1) ++issued; return; }
        if (p == 0 && issued == seg1_stage) {              // (once per tile: the activation pieces move on to segment 1)
            rs_a = rs_a1; seg_base = seg1_stage;
#pragma unroll
            for (int i = 0; i < APW; ++i) aoff[i] = (unsigned)arows1[i] * (unsigned)g.lda1 * 2u + gsegv[i];
        }
        char *db = lds + buf * G::STAGE_BYTES;
#pragma unroll
        for (int i = p * HP; i < (p + 1) * HP; ++i) {
            if (i < APW) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void *)(db + (wave + 8 * i) * 1024), 16, (int)aoff[i < APW ? i : 0],
                                                         (issued - seg_base) * 128, 0, 0);
            } else {
                const int kb = i - APW;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(db + G::A_BYTES + (kb * NT + wave) * 1024), 16, woff,
                                                         (issued * 2 + kb) * 1024, 0, 0);
            }
        }
        if (p == 1) ++issued;
    };
    // ---- what the epilogue reads besides the sums, fetched before the K loop
    f32x4 l_bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH) l_bias = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(gp(g.bias) + ncol);
    // ---- fragment addresses inside a stage buffer
    const int mrow = lane & 15, kq = lane >> 4;
    int a_rd[2];                                          // k block p of the stage: row mrow, segment (4 p + kq) ^ ((mrow >> 1) & 7)
#pragma unroll
    for (int p = 0; p < 2; ++p) a_rd[p] = (wrow + mrow) * 128 + (((p * 4 + kq) ^ ((mrow >> 1) & 7)) << 4);
    const int b_rd = G::A_BYTES + wn * NTW * 1024 + lane * 16;

    // fp16 one-chain rule (kernels.h): the accumulator is the only register set -- one MFMA chain over all k blocks, multiplied by
    // the row's BasicNorm scale where the y half of K ends; the layer-major h half continues the chain from P
    f32x4 acc[MTW][NTW];
#define APRIL_PP_EACH(expr) _Pragma("unroll") for (int mt = 0; mt < MTW; ++mt) _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) { expr; }
    APRIL_PP_EACH(acc[mt][nt] = (f32x4{0.f, 0.f, 0.f, 0.f}))
    if (EPI == EPI_LSTM && half_h && g.p_add) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int row = m0 + wrow + mt * 16 + kq * 4 + r;
                    if (row >= g.M) row = g.M - 1;
                    acc[mt][nt][r] = gp(g.p_add)[(size_t)row * g.ldp + n0 + (wn * NTW + nt) * 16 + mrow];
                }
    }
    float xrs[MTW][4];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) xrs[mt][r] = 1.0f;
    const int scale_at = (fold_scale && !half_h) ? 2 * c : -1;      // the k block (of this workgroup's range) in front of which the y half is complete
    auto apply_scale = [&]() {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = acc[mt][nt][r] * xrs[mt][r];
    };

    // ---- stages 0 and 1 are on their way before anything else of the prologue waits for memory (every independent load of the prologue -- scale partials, slots, bias, P -- was issued in front of them)
    // (unconditional, always two stages -- gemm_pp_ok guarantees them: the compiler can then COUNT the DMA pieces behind the slot loads and
    // waits for the slots with vmcnt(2 PPW) instead of draining the DMA before the cell prefetch goes out)
    issue_half(0, 0); issue_half(1, 0);
    issue_half(0, 1); issue_half(1, 1);

    // EPI_LSTM: the previous cell values of this thread's cells, behind the first DMA stages (their slots were fetched in front of them,
    // so waiting for the slots does not wait for the DMA: memory operations retire in order); nothing in the epilogue waits for memory
    float l_cprev[EPI == EPI_LSTM ? QPT : 1];
    if (EPI == EPI_LSTM) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) l_cprev[i] = gp(g.c_state)[(size_t)l_slot[i] * g.hidden + l_unit];
    }

    {
        // ---- prologue (stages 0 and 1 are in flight): the rows' scales through the third buffer
        if (fold_scale) {
            // partials (in registers since the first instruction of the kernel) -> LDS, BM threads add them in column order (the
            // order of row_scale()); rows are padded to Gn + 1 floats (conflict-free column walks).  Staged in stage buffer 2,
            // which takes its first DMA piece behind the barriers below.
            const int Gn = rsc.groups;
            float *part = reinterpret_cast<float *>(lds + 2 * G::STAGE_BYTES);
            if (ppt <= STG) {
#pragma unroll
                for (int k = 0; k < STG; ++k) if (k < ppt && sj0 + k < Gn) part[srow * (Gn + 1) + sj0 + k] = stg[k];
            } else {
                for (int i = threadIdx.x; i < BM * Gn; i += NTH) {
                    int r = m0 + i / Gn;
                    if (r >= g.M) r = g.M - 1;
                    part[(i / Gn) * (Gn + 1) + i % Gn] = gp(rsc.ssq)[(size_t)r * Gn + i % Gn];
                }
            }
            // (raw barriers: __syncthreads() would drain every DMA stage and the cell prefetch in flight)
            wait_lgkm0(); __builtin_amdgcn_s_barrier();
            if (threadIdx.x < BM) {
                float t = 0.0f;
                for (int j = 0; j < Gn; ++j) t += part[threadIdx.x * (Gn + 1) + j];
                scl[threadIdx.x] = __builtin_amdgcn_rsqf(t * rsc.inv_n + rsc.eps);
            }
            wait_lgkm0(); __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xrs[mt][r] = scl[wrow + mt * 16 + kq * 4 + r];
        }
        // stage 0 has landed (this wave's pieces); younger than it: stage 1 and the cell prefetch
        constexpr int CPV = EPI == EPI_LSTM ? QPT : 0;
        wait_vm<PPW + CPV>();
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();                      // barrier 0

        // two fragment sets: the MFMAs of k block j read set j & 1 while the reads of k block j + 1 fill the other one
        f32x4 fa0[MTW], fb0[NTW], fa1[MTW], fb1[NTW];
        auto read_frags = [&](const char *sb, int pos, f32x4 (&fa)[MTW], f32x4 (&fb)[NTW]) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) fa[mt] = *reinterpret_cast<const f32x4 *>(sb + a_rd[pos] + mt * 2048);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) fb[nt] = *reinterpret_cast<const f32x4 *>(sb + b_rd + (pos * NT + nt) * 1024);
        };
        read_frags(lds, 0, fa0, fb0);                      // k block 0
        wait_lgkm0();                                      // (every path into the loop arrives with no LDS read pending: no compiler-made waits in front of the MFMAs)
        if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one phase behind group 0

        int buf = 0, jkb = 0;
        // measurement only (tools/pp_bench_trace, PPB_TRACE=1): waves 0 and 4 accumulate s_memtime intervals of a k block's parts:
        // [0] load phase: DMA issue + fragment read issue, [1] wait at the barrier that ends the load phase, [2] MFMA issue + the
        // fragments' arrival, [3] wait at the barrier that ends the compute phase, [4] the scale (per tile)
#ifdef APRIL_GEMM_TRACE
        unsigned long long tr_acc[5] = {0, 0, 0, 0, 0}, tr_t = __builtin_amdgcn_s_memtime();
        tr_loop0 = tr_t;
        auto lapt = [&](int i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tr_acc[i] += now - tr_t; tr_t = now; };
#else
        auto lapt = [](int) {};
#endif
        // One k block of one group: load phase | barrier | compute phase | barrier.
        //   load phase     the DMA pieces of stage u + 2 (half p of this wave's share): no vector ALU, no LDS instruction;
        //   compute phase  the MFMAs of k block j with the fragment reads of k block j + 1 spread between them (one ds_read_b128 behind
        //                  every second MFMA: they issue in the shadow of the matrix pipe and land in the other register set).
        // Stage u + 1 must have landed before its first fragment read (k block 2 u + 2, read in the compute phase of k block 2 u + 1,
        // group 0 first: behind barrier 4 u + 3): group 0 waits for its pieces at the end of the load phase of k block 2 u + 1 (all of
        // stage u + 2 issued: PPW pieces stay in flight), group 1 at the end of the compute phase of k block 2 u (half 0 of stage u + 2
        // issued) -- both in front of that barrier.
        constexpr int NM = MTW * NTW, NR = MTW + NTW;
        auto kblock = [&](const int p, const bool do_issue, f32x4 (&fa)[MTW], f32x4 (&fb)[NTW], f32x4 (&fan)[MTW], f32x4 (&fbn)[NTW]) {
            const char *sb = lds + buf * G::STAGE_BYTES;
            const char *nsb = p == 0 ? sb : lds + (buf == 2 ? 0 : buf + 1) * G::STAGE_BYTES;      // stage of k block j + 1
            if (do_issue) issue_half(p, buf == 0 ? 2 : buf - 1);
            if (p == 1 && grp == 0) { if (do_issue) wait_vm<PPW>(); else wait_vm<0>(); }
            __builtin_amdgcn_sched_barrier(0);
            lapt(0);
            __builtin_amdgcn_s_barrier();
            lapt(1);
            __builtin_amdgcn_sched_barrier(0);
            if (jkb == scale_at) { apply_scale(); lapt(4); __builtin_amdgcn_sched_barrier(0); }
            // (the reads are unconditional -- behind the last k block they fetch a stage nobody needs -- so that the phase is ONE scheduling
            // region and the interleave below applies)
            read_frags(nsb, 1 - p, fan, fbn);
            if (dbg == 10) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) asm volatile("" :: "v"(fa[mt]));
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) asm volatile("" :: "v"(fb[nt]));
            } else {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[mt]), __builtin_bit_cast(h8, fb[nt]), acc[mt][nt], 0, 0, 0);
            }
            // issue order: MFMA, MFMA, read, MFMA, MFMA, read ... (every accumulator takes one MFMA per k block: their order is free)
            if constexpr (NM >= 2 * NR) {
#pragma unroll
                for (int i = 0; i < NR; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * NR, 0);
            } else {
                constexpr int NI = NM < NR ? NM : NR;
#pragma unroll
                for (int i = 0; i < NI; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                if constexpr (NM > NI) __builtin_amdgcn_sched_group_barrier(0x008, NM - NI, 0);
                if constexpr (NR > NI) __builtin_amdgcn_sched_group_barrier(0x100, NR - NI, 0);
            }
            __builtin_amdgcn_sched_barrier(0);             // (the MFMAs stay IN FRONT of the wait: they do not depend on the reads in flight)
            wait_lgkm0();                                  // the fragments of k block j + 1 are in (and this wave's reads of their stage are done)
            if (p == 0 && grp == 1) { if (do_issue) wait_vm<HP>(); else wait_vm<0>(); }
            __builtin_amdgcn_sched_barrier(0);
            lapt(2);
            __builtin_amdgcn_s_barrier();
            lapt(3);
            __builtin_amdgcn_sched_barrier(0);
            ++jkb;
        };
        int s = 0;
        for (; s + 2 < nstage; ++s) {                      // stages whose load phases issue stage s + 2
            kblock(0, true, fa0, fb0, fa1, fb1); kblock(1, true, fa1, fb1, fa0, fb0);
            buf = buf == 2 ? 0 : buf + 1;
        }
        for (; s < nstage; ++s) {
            kblock(0, false, fa0, fb0, fa1, fb1); kblock(1, false, fa1, fb1, fa0, fb0);
            buf = buf == 2 ? 0 : buf + 1;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        if (jkb == scale_at) apply_scale();                // (EPI_XPART: the y half is all there is)
#ifdef APRIL_GEMM_TRACE
        tr_loop1 = __builtin_amdgcn_s_memtime();
        if (g.trace && (wave == 0 || wave == 4) && lane == 0) {
            const size_t wg = blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
            for (int i = 0; i < 5; ++i) g.trace[wg * 16 + (wave >> 2) * 8 + i] = tr_acc[i];
            g.trace[wg * 16 + (wave >> 2) * 8 + 5] = tr_loop0 - tr_start;
            g.trace[wg * 16 + (wave >> 2) * 8 + 7] = (unsigned long long)T;
        }
#endif
    }

    // ---- the workgroup's sums -> LDS plane (each wave owns its part; no cross-wave addition) -> 4-column quads per thread
    __syncthreads();                                       // the last stage has been read by every wave
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wrow + mt * 16 + kq * 4 + r) * LDR + (wn * NTW + nt) * 16 + mrow] = acc[mt][nt][r];
    __syncthreads();
    f32x4 v[QPT];
#pragma unroll
    for (int i = 0; i < QPT; ++i) v[i] = *reinterpret_cast<const f32x4 *>(red + (qrow0 + QRS * i) * LDR + qcol * 4);

    if (EPI == EPI_XPART) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int m = m0 + qrow0 + QRS * i;
            if (m < g.M) *reinterpret_cast<__attribute__((address_space(1))) f32x4 *>(gp(g.out) + (size_t)m * g.ldo + ncol) = v[i];
        }
    } else if (EPI == EPI_BIAS_DSWISH) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int m = m0 + qrow0 + QRS * i;
            if (m < g.M) {
                const f32x4 y = v[i] + l_bias;
                f32x4 o;
                o.x = y.x * fast_sigmoid(y.x - 1.0f); o.y = y.y * fast_sigmoid(y.y - 1.0f);
                o.z = y.z * fast_sigmoid(y.z - 1.0f); o.w = y.w * fast_sigmoid(y.w - 1.0f);
                if (g.out) *reinterpret_cast<__attribute__((address_space(1))) f32x4 *>(gp(g.out) + (size_t)m * g.ldo + ncol) = o;
                if (g.out16) *reinterpret_cast<__attribute__((address_space(1))) h4 *>(gp(reinterpret_cast<_Float16 *>(g.out16)) + (size_t)m * g.ldo + ncol) = to_h4(o);
            }
        }
    } else {   // EPI_LSTM: the quad = gates i, f, g, o of one hidden unit; the BasicNorm scale of the y half was folded in after chunk 1
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int m = m0 + qrow0 + QRS * i;
            const f32x4 gt = v[i] + l_bias;
            const float c_new = fast_sigmoid(gt.y) * l_cprev[i] + fast_sigmoid(gt.x) * fast_tanh(gt.z);
            const float u = fast_sigmoid(gt.w) * fast_tanh(c_new);
            if (m < g.M) {
                gp(g.c_state)[(size_t)l_slot[i] * g.hidden + l_unit] = c_new;
                if (g.out) gp(g.out)[(size_t)m * g.ldo + l_unit] = u;
                if (g.out16) gp(reinterpret_cast<_Float16 *>(g.out16))[(size_t)m * g.ldo + l_unit] = (_Float16)u;
            }
        }
    }
#ifdef APRIL_GEMM_TRACE
    if (g.trace && (wave == 0 || wave == 4) && lane == 0) {
        const size_t wg = blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        g.trace[wg * 16 + (wave >> 2) * 8 + 6] = __builtin_amdgcn_s_memtime() - tr_loop1;
    }
#endif
#undef APRIL_PP_EACH
}

template <int MT, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(GemmArgs g)
{
    if constexpr (EPI == EPI_LSTM) stamp_begin(g.stamp, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
    gemm_pp_body<MT, EPI>(g);
    if constexpr (EPI == EPI_LSTM) stamp_end(g.stamp, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// n independent same-shape problems in one launch (see gemm_f32_zkernel): blockIdx.z picks the argument block
template <int MT, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_pp_zkernel(const GemmArgs *__restrict__ zargs)
{
    const GemmArgs g = zargs[blockIdx.z];
    if constexpr (EPI == EPI_LSTM) stamp_begin(g.stamp, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
    gemm_pp_body<MT, EPI>(g);
    if constexpr (EPI == EPI_LSTM) stamp_end(g.stamp, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

template <int MT, int EPI>
void launch_pp_one(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    using G = PPGeom<MT>;
    dim3 grid((unsigned)(g.N / G::BN), (unsigned)((g.M + G::BM - 1) / G::BM), (unsigned)(dev_args ? n : 1));
    // dynamic LDS beyond 64 KB has to be announced, per instantiation AND per device (one engine per GPU)
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_pp_kernel<MT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_pp_zkernel<MT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    if (dev_args) APRIL_LAUNCH((gemm_pp_zkernel<MT, EPI>), grid, dim3(G::NTH), (size_t)G::LDS_BYTES, s, dev_args);
    else APRIL_LAUNCH((gemm_pp_kernel<MT, EPI>), grid, dim3(G::NTH), (size_t)G::LDS_BYTES, s, g);
}

}  // namespace

// can this GEMM run on GM_PP with 16 * mt tile rows?  (operand shapes only; the planner decides whether it should)
bool gemm_pp_ok(const GemmArgs &g, int mt)
{
    if (g.wt != 1 || g.kz != 1 || g.a_op != AOP_NONE || (mt != 16 && mt != 8)) return false;
    if (g.epi != EPI_LSTM && g.epi != EPI_BIAS_DSWISH && g.epi != EPI_XPART) return false;
    if (g.N % 128 != 0 || g.K % 128 != 0) return false;                     // chunks of whole 32-k blocks
    const bool half_x = g.wave_mask == 0x3, half_h = g.epi == EPI_LSTM && g.wave_mask == 0xC;
    if (!half_x && !half_h && g.wave_mask != 0xF) return false;
    if (half_x && g.epi == EPI_BIAS_DSWISH) return false;
    const int c = g.K / 128;
    if ((half_x || half_h) && ((c & 1) || g.K < 256)) return false;         // the half walks 2 c k blocks = whole stages, at least two of them
    if (g.K1 > 0 && (g.K0 % 64 != 0 || g.K0 + g.K1 != g.K)) return false;
    if (g.K1 == 0 && g.K0 != g.K) return false;
    if ((g.epi == EPI_LSTM || g.epi == EPI_XPART) && g.x_scale.ssq) {
        // the partials of a tile's rows are staged in one stage buffer
        if ((size_t)16 * mt * (g.x_scale.groups + 1) * 4 > (size_t)(16 * mt * 128 + 16384)) return false;
    }
    // 32-bit lane offsets into the activation rows (rows are slots when indexed: bounded by what the caller allocated; the engine's
    // arrays are far below 2 GB per layer, and the launch wrapper cannot see their extent: the direct rows are checked here)
    if ((uint64_t)g.M * (uint64_t)g.lda0 * 2 >= (1ull << 31)) return false;
    return true;
}

void launch_gemm_pp(const GemmArgs &g, int mt, const GemmArgs *dev_args, int n, hipStream_t s)
{
    bool ok = gemm_pp_ok(g, mt);
    if (ok) {
        if (mt == 16) {
            if (g.epi == EPI_LSTM) launch_pp_one<16, EPI_LSTM>(g, dev_args, n, s);
            else if (g.epi == EPI_BIAS_DSWISH) launch_pp_one<16, EPI_BIAS_DSWISH>(g, dev_args, n, s);
            else launch_pp_one<16, EPI_XPART>(g, dev_args, n, s);
        } else {
            if (g.epi == EPI_LSTM) launch_pp_one<8, EPI_LSTM>(g, dev_args, n, s);
            else if (g.epi == EPI_BIAS_DSWISH) launch_pp_one<8, EPI_BIAS_DSWISH>(g, dev_args, n, s);
            else launch_pp_one<8, EPI_XPART>(g, dev_args, n, s);
        }
    }
    if (!ok) { fprintf(stderr, "libapril(mi355x): launch_gemm_pp: no kernel for epi %d tile rows %d (wt %d kz %d)\n", g.epi, 16 * mt, g.wt, g.kz); abort(); }
}

}  // namespace aprilx
