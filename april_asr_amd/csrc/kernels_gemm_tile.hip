// GM_TILE schedule of the MFMA GEMM (kernels.h) for gfx950: the workgroup's four waves split the OUTPUT tile (2 x 2) and
// every wave walks all of the workgroup's K range, so both operands are shared between waves through LDS.
//
// Why it exists: the N = d_model GEMMs (LSTM projection K = hidden, FFN down K = ffn, embed linear, encoder_proj) have few
// outputs and a long K.  The K-split schedules (GM_SLAB / GM_FULLK) give every wave its own k range, so nothing is shared
// and the workgroup's operand traffic is that of its tile shape: 16 x 32 tiles at 256 rows = 5.3 flop per byte through L1,
// which held those kernels at 0.27 .. 0.36 of the fp32 MFMA peak (round-2 profile), 0.44 .. 0.50 with 64 x 32 tiles at
// thousands of rows.  Here a 64 x 64 tile moves (64 + 64) * 4 bytes per k for 2 * 64 * 64 flops = 16 flop per byte, the
// activations arrive in full 128-byte lines (LDS DMA, global_load_lds_dwordx4) instead of 16 rows x 64 bytes per
// instruction, and the chip is filled by cutting K across workgroups at slab boundaries (grid.z = kz / zs) when the
// output tiles alone are too few.
//
// Canonical summation (kernels.h) is kept exactly: a chunk is one in-order MFMA chain over its k blocks, a slab is
// ((c0 + c1) + c2) + c3, slabs meet pairwise in slab order.  A wave folds chunk -> slab -> tree levels in registers (as
// GM_FULLK does); a workgroup that owns all kz slabs finishes the row epilogue itself, otherwise it writes the tree sum of
// its zs slabs as one partial plane and the row kernel (or decide_kernel) finishes the same tree.  Same chains, same tree
// => bit-identical to GM_SLAB / GM_FULLK, whatever the batch size picks.
//
// LDS image of one stage (two k blocks = 32 k = 128 bytes per activation row):
//   A: [16 MT rows][128 B], 16-byte segment g of row R stored at segment g ^ ((R >> 1) & 7): the MFMA A fragment read
//      (lane (i, kq) reads row i, segment 4 p + kq) is then conflict-free for ds_read_b128's lane groups.  The DMA writes
//      LDS linearly (wave-uniform base + lane * 16), so the swizzle is applied to the per-lane SOURCE address.
//   B: [2 k blocks][4 n tiles][1 KB] in the packed weight order, i.e. already the B fragment of every lane.
// Four stage buffers, ONE s_barrier per stage, placed between the stage's two k blocks (see the K loop): counted vmcnt keeps
// two to three stages of DMA in flight, the fragments of the next k block are read from LDS while the current MFMAs issue.
// WT = 1 (fp16-operand mode, BASELINE configs[4]): the same kernel on binary16 operands -- a k block is 32 k
// (v_mfma_f32_16x16x32_f16, the gfx950 shape; fp32 accumulate), weights packed per (n tile, 32-k block) as the B fragment of
// that instruction, activations READ as binary16 (written by the producing epilogue: out16 / state16), so a stage is again
// 128 bytes per activation row and 1 KB per weight piece and everything above holds unchanged.  The summation structure
// (chunks, slabs, tree) is the same; the chains are MFMA-internal sums of 32 products instead of in-order fp32 FMAs.
// Epilogues: the row epilogues and partial planes, plus (for the gates and FFN-up GEMMs of fp16 engines, and for
// measurements in fp32) EPI_LSTM over two A segments [y | h(slot)] with the BasicNorm scale folded in after the second
// chunk, and EPI_BIAS_DSWISH.
// Replaces the ORT MatMul nodes inside the encoder / joiner graphs (reference call sites src/april_session.c:145,176).
#include "kernels.h"
#include "device_utils.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace aprilx {

namespace {

#ifndef APRIL_TILE_STAGES
#define APRIL_TILE_STAGES 3
#endif
constexpr int TILE_STAGES = APRIL_TILE_STAGES;

// Tile shape: 16 MT rows x 16 NT columns per workgroup, NWM x NWN waves, each owning (MT / NWM) x (NT / NWN) MFMA tiles.
//   <2, 4, 2, 2>, <4, 4, 2, 2>   32 / 64 rows x 64 columns, four waves (all epilogues, fp32 and fp16)
//   <4, 8, 2, 4>                 64 x 128, eight waves (wave tile 32 x 32): the fp16 projection / FFN-down GEMMs (N = d_model)
//   <8, 8, 2, 4>                 128 x 128, eight waves: twice the flops per operand byte -- the fp16 gates and FFN-up GEMMs, whose
//                                k block is 64 SIMD cycles of MFMA against 8 KB of operands at 64 x 64 (bound by the CU's L2 -> LDS rate)
template <int MT, int NT = 4, int NWM = 2, int NWN = 2, int NS_ = TILE_STAGES> struct TileGeom {
    static constexpr int NS = NS_;                                        // stage buffers
    static constexpr int BM = 16 * MT, BN = 16 * NT, LDR = BN + 4, NW = NWM * NWN, NTH = 64 * NW;
    static constexpr int A_BYTES = BM * 128, B_BYTES = 2 * NT * 1024, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int PIECES = 2 * MT + 2 * NT, PPW = PIECES / NW;      // 1 KB DMA pieces per stage / per wave
    static constexpr int MTW = MT / NWM, NTW = NT / NWN;                  // MFMA tiles per wave
    static constexpr int PLANE_FLOATS = BM * LDR;
    static constexpr int LDS_MAIN = (NS * STAGE_BYTES > PLANE_FLOATS * 4) ? NS * STAGE_BYTES : PLANE_FLOATS * 4;
    static_assert(NS >= 3 && NS <= 8, "three to eight stage buffers (the tail of the K loop settles every outstanding stage once fewer than NS - 3 remain)");
    static_assert(PIECES % NW == 0 && MT % NWM == 0 && NT % NWN == 0, "pieces and tiles are dealt evenly to the waves");
};

template <int N> __device__ __forceinline__ void wait_vm()
{
    // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14; expcnt / lgkmcnt at their maxima)
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

using h4 = __attribute__((ext_vector_type(4))) _Float16;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ h4 to_h4(const f32x4 &v) { return h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; }

template <int MT, int EPI, int WT, int NT = 4, int NWM = 2, int NWN = 2, int NSB = TILE_STAGES>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs &g, const int zg)
{
    using G = TileGeom<MT, NT, NWM, NWN, NSB>;
    constexpr int BM = G::BM, MTW = G::MTW, NTW = G::NTW, NTH = G::NTH, LDR = G::LDR, TILE_BN = G::BN;
    constexpr int KBLK = WT ? 32 : 16, AE = WT ? 2 : 4;          // k per k block, bytes per activation element
    constexpr bool ROW_EPI = EPI == EPI_HR || EPI == EPI_RESID_SSQ || EPI == EPI_SLOT_STORE;
    static_assert(ROW_EPI || EPI == EPI_PARTIAL || EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH || EPI == EPI_XPART, "no GM_TILE form of this epilogue");
    extern __shared__ __attribute__((aligned(1024))) float red[];
    char *lds = reinterpret_cast<char *>(red);

    if (g.run_flag && *g.run_flag != g.run_gen) return;
    if constexpr (NWM * NWN == 4) first_round_skew(g.skew, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), (unsigned)g.skew_wgs);      // (measurement form, off by default: device_utils.h)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    // measurement only (tools/tile_bench built with -DAPRIL_GEMM_TRACE): wave 0 accumulates s_memtime intervals of the K loop's
    // phases: [0] reads + first MFMA block, [1] chunk ends, [2] DMA wait + barrier, [3] DMA issue + reads + second MFMA block,
    // [4] whole loop, [5] prologue up to the loop, [6] epilogue; compiled out of the product
#ifdef APRIL_GEMM_TRACE
    unsigned long long tr_acc[7] = {0, 0, 0, 0, 0, 0, 0}, tr_t = __builtin_amdgcn_s_memtime(), tr_start = tr_t;
    auto lapt = [&](int i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tr_acc[i] += now - tr_t; tr_t = now; };
#else
    auto lapt = [](int) {};
#endif
    const int n0 = blockIdx.x * TILE_BN;                 // first output column
    const int m0 = blockIdx.y * BM;
    const int KB = g.K / KBLK;
    const int c = KB / (4 * g.kz);                       // k blocks per chunk
    // layer-major split of the gate GEMM (kernels.h, wave_mask; kz = 1, K0 = K1 = K / 2): 0x3 = the input half alone (chunks 0, 1:
    // EPI_XPART writes P = (c0 + c1) * scale), 0xC = the recurrent half alone on top of P (chunks 2, 3: ((P + c2) + c3) + bias).
    // The K-split kernels hand those halves to wave pairs; here the workgroup simply walks half of the k blocks.
    const bool half_x = (EPI == EPI_XPART || EPI == EPI_LSTM) && g.wave_mask == 0x3, half_h = EPI == EPI_LSTM && g.wave_mask == 0xC;
    const int T = (half_x || half_h) ? 2 * c : 4 * c * g.zs;      // k blocks of this workgroup (even)
    const int first_kb = half_h ? 2 * c : zg * T;
    const int nstage = T >> 1;

    // ---- BasicNorm scales of the tile's rows (EPI_HR: residual; EPI_SLOT_STORE: the whole sum), as in gemm_body: the partials
    // make one trip from global memory at kernel start and are added up after the K loop
    const RowScale &rsc = EPI == EPI_HR ? g.r_scale : g.x_scale;
    const bool NEED_SCL = (EPI == EPI_HR || EPI == EPI_SLOT_STORE || EPI == EPI_LSTM || EPI == EPI_XPART) && rsc.ssq != nullptr;
    float *scl = red + G::LDS_MAIN / 4;
    float stg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int TPR = NTH / BM;
    const int ppt = (rsc.groups + TPR - 1) / TPR;
    const bool staged = NEED_SCL && ppt <= 4;
    const int srow = threadIdx.x / TPR, sj0 = (threadIdx.x % TPR) * ppt;
    if (NEED_SCL && staged) {
        int r = m0 + srow;
        if (r >= g.M) r = g.M - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ppt && sj0 + k < rsc.groups) stg[k] = rsc.ssq[(size_t)r * rsc.groups + sj0 + k];
    }

    // ---- DMA pieces of this wave: piece P = wave + 4 i; P < 2 MT: activation rows 8 P .. 8 P + 7 (lane -> row P 8 + (lane >> 3),
    // 16-byte segment lane & 7 of the stage's 128 bytes, swizzled); otherwise weight piece (k block p, n tile nt) of the stage.
    // The activations may come in two K segments (gates: [y | h(slot)], K0 a multiple of the stage's 2 KBLK k): the piece
    // pointers switch to segment 1 at stage `seg1_stage` of this workgroup.
    const char *src[G::PPW], *src1[G::PPW];
    int inc[G::PPW], dst[G::PPW];
    const int k_begin = first_kb * KBLK;
    const bool two_seg = g.K1 > 0;
    const bool start_in_1 = two_seg && k_begin >= g.K0;
    const int seg1_stage = (two_seg && !start_in_1) ? (g.K0 - k_begin) / (2 * KBLK) : 0x7fffffff;
    {
        int arows[G::PPW], arows1[G::PPW];
#pragma unroll
        for (int i = 0; i < G::PPW; ++i) {
            const int P = wave + G::NW * i;
            int row = m0 + P * 8 + (lane >> 3);
            arows[i] = row >= g.M ? g.M - 1 : row;       // padding rows recompute the last row; never stored
            arows1[i] = arows[i];
        }
        if (g.aidx0) {                                    // row -> slot indirections, one round trip for all pieces
#pragma unroll
            for (int i = 0; i < G::PPW; ++i) if (wave + G::NW * i < 2 * MT) arows[i] = g.aidx0[arows[i]];
        }
        if (two_seg && g.aidx1) {
#pragma unroll
            for (int i = 0; i < G::PPW; ++i) if (wave + G::NW * i < 2 * MT) arows1[i] = g.aidx1[arows1[i]];
        }
#pragma unroll
        for (int i = 0; i < G::PPW; ++i) {
            const int P = wave + G::NW * i;
            src1[i] = nullptr;
            if (P < 2 * MT) {
                const int R = P * 8 + (lane >> 3);
                const int gseg = (lane & 7) ^ ((R >> 1) & 7);
                const char *p0 = reinterpret_cast<const char *>(g.a0) + ((size_t)arows[i] * g.lda0 + (size_t)k_begin) * AE + gseg * 16;
                if (two_seg) {
                    const char *p1 = reinterpret_cast<const char *>(g.a1) + (size_t)arows1[i] * g.lda1 * AE + gseg * 16;
                    src1[i] = p1;
                    if (start_in_1) p0 = p1 + (size_t)(k_begin - g.K0) * AE;
                }
                src[i] = p0;
                inc[i] = 128; dst[i] = P * 1024;
            } else {
                const int q = P - 2 * MT, p = q / NT, nt = q % NT;
                src[i] = reinterpret_cast<const char *>(g.wp) + ((size_t)(blockIdx.x * NT + nt) * KB + first_kb + p) * 1024 + lane * 16;
                inc[i] = 2048; dst[i] = G::A_BYTES + q * 1024;
            }
        }
    }
    int issued = 0;
    auto issue_begin = [&]() {                             // once per stage, before its pieces
        if (two_seg) {                                     // (selects, not a branch: the stage body stays one scheduling region)
            const bool sw = issued == seg1_stage;
#pragma unroll
            for (int i = 0; i < G::PPW; ++i) if (wave + G::NW * i < 2 * MT) src[i] = sw ? src1[i] : src[i];
        }
        ++issued;
    };
    auto issue_piece = [&](int i, int buf) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i]),
                                         (__attribute__((address_space(3))) void *)(lds + buf * G::STAGE_BYTES + dst[i]), 16, 0, 0);
        src[i] += inc[i];
    };
    auto issue = [&](int buf) {
#ifdef APRIL_GEMM_TRACE
        if (g.debug == 4) return;                          // measurement build: no DMA (the MFMA + LDS read loop alone, on stale LDS contents)
#endif
        issue_begin();
#pragma unroll
        for (int i = 0; i < G::PPW; ++i) issue_piece(i, buf);
    };

    // ---- what the row epilogue reads besides the sums: fetched before the K loop (as in gemm_body)
    constexpr int QROW = TILE_BN / 4, NQ = BM * QROW, QPT = (NQ + NTH - 1) / NTH;
    f32x4 e_bias[ROW_EPI ? QPT : 1], e_res[ROW_EPI ? QPT : 1];
    int e_slot[ROW_EPI ? QPT : 1];
    bool e_ok[ROW_EPI ? QPT : 1];
    if (ROW_EPI) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            int m = m0 + q / QROW;
            const int n = n0 + (q % QROW) * 4;
            e_ok[i] = q < NQ && m < g.M;
            if (m >= g.M) m = g.M - 1;
            const int qn = q < NQ ? n : n0;
            e_slot[i] = (EPI != EPI_RESID_SSQ && g.slot_idx) ? g.slot_idx[m] : m;
            if (EPI == EPI_SLOT_STORE && g.row_mask && !g.row_mask[m]) e_ok[i] = false;
            e_bias[i] = (EPI != EPI_HR) ? *reinterpret_cast<const f32x4 *>(g.bias + qn) : f32x4{0.f, 0.f, 0.f, 0.f};
            e_res[i] = (EPI == EPI_HR || (EPI == EPI_RESID_SSQ && g.resid)) ? *reinterpret_cast<const f32x4 *>(g.resid + (size_t)m * g.ldr + qn) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // EPI_LSTM: this thread's (row, hidden unit) pairs: slot -> previous cell value and the gate biases, fetched up front
    // (columns are unit-major: the 4-column quad = gates i, f, g, o of one unit)
    int l_unit[EPI == EPI_LSTM ? QPT : 1], l_m[EPI == EPI_LSTM ? QPT : 1];
    bool l_ok[EPI == EPI_LSTM ? QPT : 1];
    float *l_cptr[EPI == EPI_LSTM ? QPT : 1];
    float l_cprev[EPI == EPI_LSTM ? QPT : 1];
    f32x4 l_bias[(EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH) ? QPT : 1];
    if (EPI == EPI_LSTM) {
        int qslot[QPT];
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            int r = m0 + q / QROW;
            l_m[i] = r;
            l_ok[i] = q < NQ && r < g.M;
            if (r >= g.M) r = g.M - 1;
            qslot[i] = g.slot_idx[r];
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int n = n0 + (q % QROW) * 4;
            l_unit[i] = n >> 2;
            l_cptr[i] = g.c_state + (size_t)qslot[i] * g.hidden + l_unit[i];      // (padding rows point at the last row's cell: read, never stored)
            l_bias[i] = *reinterpret_cast<const f32x4 *>(g.bias + n);
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) l_cprev[i] = *l_cptr[i];
    }
    if (EPI == EPI_BIAS_DSWISH) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) l_bias[i] = *reinterpret_cast<const f32x4 *>(g.bias + n0 + ((threadIdx.x + i * NTH) % QROW) * 4);
    }

    // ---- fragment addresses inside a stage buffer
    const int mrow = lane & 15, kq = lane >> 4;
    int a_rd[2];                                          // k block p of the stage: row mrow, segment (4 p + kq) ^ ((mrow >> 1) & 7)
#pragma unroll
    for (int p = 0; p < 2; ++p) a_rd[p] = (wm * MTW * 16 + mrow) * 128 + (((p * 4 + kq) ^ ((mrow >> 1) & 7)) << 4);
    const int b_rd = G::A_BYTES + wn * NTW * 1024 + lane * 16;

    // chunk chain, slab sum, the three levels of the pairwise slab tree (named, not an array: a level array indexed under the
    // carry conditions is not promoted to registers and lands in scratch memory), the workgroup's result
    f32x4 acc[MTW][NTW], S[MTW][NTW], lvl0[MTW][NTW], lvl1[MTW][NTW], lvl2[MTW][NTW], res[MTW][NTW];
#define APRIL_TILE_EACH(expr) _Pragma("unroll") for (int mt = 0; mt < MTW; ++mt) _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) { expr; }
    APRIL_TILE_EACH(acc[mt][nt] = (f32x4{0.f, 0.f, 0.f, 0.f}); S[mt][nt] = acc[mt][nt]; res[mt][nt] = acc[mt][nt];
                    lvl0[mt][nt] = acc[mt][nt]; lvl1[mt][nt] = acc[mt][nt]; lvl2[mt][nt] = acc[mt][nt])
    int top = 0;
    while ((1 << top) < g.zs) ++top;
    int chunk_i = 0, slab_done = 0, in_chunk = 0;
    // binary16 operands, kz = 1 (gates, FFN up and the layer-major halves of the gate GEMM; round 6, kernels.h "fp16 one-chain rule"):
    // ONE MFMA chain over all k blocks in k order, the BasicNorm scale multiplied into the running sum where the y half of K ends
    // -- no chunk sums, no slab register set.  GM_PP (kernels_gemm_pp.hip) computes exactly this; the two are compared bitwise by
    // tools/pp_bench.  (fp32 operands keep the chunk / slab form of the K-split kernels.)
    constexpr bool ONE_CHAIN = WT == 1 && (EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH || EPI == EPI_XPART);
    if (EPI == EPI_LSTM && half_h && g.p_add) {
        // the slab starts as P (this lane's accumulator elements of the tile, fetched now, used at the first chunk end): S = P,
        // then S + c2, then S + c3 -- the canonical ((P + c2) + c3); one-chain form: the chain continues from P
        chunk_i = 2;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int row = m0 + (wm * MTW + mt) * 16 + (lane >> 4) * 4 + r;
                    if (row >= g.M) row = g.M - 1;
                    (ONE_CHAIN ? acc : S)[mt][nt][r] = g.p_add[(size_t)row * g.ldp + n0 + (wn * NTW + nt) * 16 + (lane & 15)];
                }
    }
    // EPI_LSTM with x_scale: x = y * scale(y) entered the GEMM as y, the first two chunks are exactly the y half of K (kz = 1,
    // K0 = K / 2, checked on the host): ((c0 + c1) * scale + c2) + c3, the scale of this lane's accumulator rows held in registers
    float xrs[MTW][4];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) xrs[mt][r] = 1.0f;
    const bool fold_scale = (EPI == EPI_LSTM || EPI == EPI_XPART) && NEED_SCL;
    auto chunk_end = [&]() {
        if constexpr (ONE_CHAIN) {
            if (fold_scale && chunk_i == 1) {              // the y half of K ends here: the running sum takes the row's BasicNorm scale
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[mt][nt][r] = acc[mt][nt][r] * xrs[mt][r];
            }
            ++chunk_i;
            return;
        }
        if (chunk_i == 0) { APRIL_TILE_EACH(S[mt][nt] = acc[mt][nt]) }
        else { APRIL_TILE_EACH(S[mt][nt] = S[mt][nt] + acc[mt][nt]) }
        if ((EPI == EPI_LSTM || EPI == EPI_XPART) && fold_scale && chunk_i == 1) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[mt][nt][r] = S[mt][nt][r] * xrs[mt][r];
        }
        APRIL_TILE_EACH(acc[mt][nt] = (f32x4{0.f, 0.f, 0.f, 0.f}))
        if (++chunk_i == 4) {                             // slab complete: S = ((c0 + c1) + c2) + c3 enters the pairwise tree (binary counter)
            chunk_i = 0;
            bool done = false;                            // true once the value has been parked in a level
            constexpr bool TREE = !(EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH);      // (those GEMMs have one slab: kz = 1)
            if (TREE && top > 0) {
                if (slab_done & 1) { APRIL_TILE_EACH(S[mt][nt] = lvl0[mt][nt] + S[mt][nt]) }
                else { APRIL_TILE_EACH(lvl0[mt][nt] = S[mt][nt]) done = true; }
            }
            if (TREE && !done && top > 1) {
                if (slab_done & 2) { APRIL_TILE_EACH(S[mt][nt] = lvl1[mt][nt] + S[mt][nt]) }
                else { APRIL_TILE_EACH(lvl1[mt][nt] = S[mt][nt]) done = true; }
            }
            if (TREE && !done && top > 2) {
                if (slab_done & 4) { APRIL_TILE_EACH(S[mt][nt] = lvl2[mt][nt] + S[mt][nt]) }
                else { APRIL_TILE_EACH(lvl2[mt][nt] = S[mt][nt]) done = true; }
            }
            if (!done) { APRIL_TILE_EACH(res[mt][nt] = S[mt][nt]) }      // all zs slabs of this workgroup are in
            ++slab_done;
        }
    };

    // the rows' BasicNorm scales: partials (in registers since the first instruction of the kernel) -> LDS, BM threads add them in
    // column order (the order of row_scale()); rows are padded to G + 1 floats (conflict-free column walks)
    auto compute_scl = [&]() {
        const int Gn = rsc.groups;
        float *part = scl + BM;
        if (staged) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < ppt && sj0 + k < Gn) part[srow * (Gn + 1) + sj0 + k] = stg[k];
        } else {
            for (int i = threadIdx.x; i < BM * Gn; i += NTH) {
                int r = m0 + i / Gn;
                if (r >= g.M) r = g.M - 1;
                part[(i / Gn) * (Gn + 1) + i % Gn] = rsc.ssq[(size_t)r * Gn + i % Gn];
            }
        }
        __syncthreads();
        if (threadIdx.x < BM) {
            float t = 0.0f;
            for (int j = 0; j < Gn; ++j) t += part[threadIdx.x * (Gn + 1) + j];
            scl[threadIdx.x] = __builtin_amdgcn_rsqf(t * rsc.inv_n + rsc.eps);
        }
        __syncthreads();
    };

    // ---- K loop.  Stage s = k blocks (2 s, 2 s + 1) of the workgroup's range, buffer s % NS.  The fragments of the NEXT k block
    // are read while the MFMAs of the current one issue, across the stage boundary too: the one barrier of a stage sits between its
    // two k blocks -- behind it every wave's pieces of stage s + 1 have landed (counted vmcnt before the barrier) and every wave has
    // issued its reads of stage s, so the buffer of stage s - 1 (read an iteration ago, consumed by MFMAs since) takes stage s + NS - 1.
    if (g.debug != 1) {
        constexpr int NS = G::NS;
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) if (i < nstage) issue(i);
        if ((EPI == EPI_LSTM || EPI == EPI_XPART) && fold_scale) {
            // the scales are needed INSIDE the loop (after the second chunk): reduce them now, behind the first DMA stages (the
            // compiler settles every outstanding memory operation here, which the first stage needs anyway)
            compute_scl();
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xrs[mt][r] = scl[(wm * MTW + mt) * 16 + (lane >> 4) * 4 + r];
        }
        // stage 0 landed: at most the NS - 2 younger stages may still be in flight
        if (nstage >= NS - 1) wait_vm<(NS - 2) * G::PPW>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        f32x4 a0[MTW], b0[NTW], a1[MTW], b1[NTW];
        auto read_frags = [&](const char *sb, int p, f32x4 (&a)[MTW], f32x4 (&b)[NTW]) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) a[mt] = *reinterpret_cast<const f32x4 *>(sb + a_rd[p] + mt * 2048);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) b[nt] = *reinterpret_cast<const f32x4 *>(sb + b_rd + (p * NT + nt) * 1024);
        };
        auto mfma_block = [&](const f32x4 (&a)[MTW], const f32x4 (&b)[NTW]) {
#ifdef APRIL_GEMM_TRACE
            if (g.debug == 5) {                            // measurement build: no MFMAs (DMA + barriers + LDS reads alone); the fragments stay live
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) asm volatile("" :: "v"(a[mt]));
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) asm volatile("" :: "v"(b[nt]));
                return;
            }
#endif
            // k step outermost: consecutive MFMAs go to different accumulators (a dependent MFMA issues 8 cycles late); per
            // accumulator the order is k = j, j + 4, j + 8, j + 12 inside the MFMA, j = 0..3 across MFMAs: the canonical chain
            if constexpr (WT == 1) {
                // one v_mfma_f32_16x16x32_f16 per tile and k block: lane (i, kq) holds k = 8 kq .. 8 kq + 7 of the block for A and B alike
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[mt]), __builtin_bit_cast(h8, b[nt]), acc[mt][nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
            }
        };
        read_frags(lds, 0, a0, b0);
        int buf = 0;
        lapt(5);
        int s = 0;
        // Fast path (chunks of whole stages: c even).  A wave issues in order, so whatever sits BETWEEN two MFMA blocks -- fragment
        // reads, DMA issue and its address arithmetic, loop control -- runs while the matrix pipe drains (measured with one workgroup
        // per CU: 1885 cycles per stage for 1024 cycles of MFMA, and 1032 with the MFMAs removed: additive).  Here the stage body is
        // branch-free (DMA issue and next-stage reads unconditional: the last NS - 1 stages and odd chunk lengths take the generic
        // loop below) and sched_group_barrier spreads the reads and the DMA issue through the MFMAs of the same scheduling region.
        if ((c & 1) == 0 && g.debug != 6) {
            constexpr int TILES = MTW * NTW, NM = TILES * (WT ? 1 : 4), NR = MTW + NTW, NV = G::PPW;      // MFMAs per k block, fragment reads, DMA pieces
            // MFMA q of a k block: k step q / TILES (fp32), tile q % TILES -- consecutive MFMAs on different accumulators, the k steps of
            // one accumulator in order (the canonical chain)
            auto mfma_one = [&](int q, const f32x4 (&a)[MTW], const f32x4 (&b)[NTW]) {
                const int t = q % TILES, mt = t / NTW, nt = t % NTW;
                if constexpr (WT == 1) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[mt]), __builtin_bit_cast(h8, b[nt]), acc[mt][nt], 0, 0, 0);
                else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][q / TILES], b[nt][q / TILES], acc[mt][nt], 0, 0, 0);
            };
            auto read_one = [&](int f, const char *sb, int p, f32x4 (&a)[MTW], f32x4 (&b)[NTW]) {
                if (f < MTW) a[f] = *reinterpret_cast<const f32x4 *>(sb + a_rd[p] + f * 2048);
                else b[f - MTW] = *reinterpret_cast<const f32x4 *>(sb + b_rd + (p * NT + (f - MTW)) * 1024);
            };
            // fillers are front-loaded, PF / PF2 behind each of the first MFMAs, so that the rest of the block covers their latency
            // (fp32: one per MFMA; fp16 blocks have fewer MFMAs than fillers)
            constexpr int PF = (2 * NR + NM - 1) / NM > 1 ? (2 * NR + NM - 1) / NM : 1, PF2 = (2 * (NV + NR) + NM - 1) / NM > 1 ? (2 * (NV + NR) + NM - 1) / NM : 1;
            const int spc = c >> 1, main_end = nstage - (NS - 1);
            while (s + spc <= main_end) {
                for (int j = 0; j < spc; ++j, ++s) {
                    const char *sb = lds + buf * G::STAGE_BYTES;
                    int nbuf = buf + 1; if (nbuf == NS) nbuf = 0;
                    int ib = buf - 1; if (ib < 0) ib += NS;
                    const char *nsb = lds + nbuf * G::STAGE_BYTES;
#ifdef APRIL_GEMM_TRACE
                    lapt(3); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lapt(6);      // [6] = time waiting for the fragments of k block 0
#endif
                    // k block 0 of the stage; the fragment reads of k block 1 go out between its MFMAs (pinned by sched_barrier: left
                    // to itself the scheduler clusters them at one end, and the solver behind sched_group_barrier reorders the chains)
#pragma unroll
                    for (int q = 0; q < NM; ++q) {
                        mfma_one(q, a0, b0);
#pragma unroll
                        for (int f = q * PF; f < (q + 1) * PF && f < NR; ++f) read_one(f, sb, 1, a1, b1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    lapt(0);
                    wait_vm<(NS - 3) * G::PPW>();          // stage s + 1 has landed (this wave's pieces); NS - 3 younger stages stay in flight
                    __builtin_amdgcn_s_barrier();
                    lapt(2);
#ifdef APRIL_GEMM_TRACE
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lapt(5);                // [5] = time waiting for the fragments of k block 1
#endif
                    issue_begin();
                    // k block 1; DMA pieces of stage s + NS - 1 first, then the reads of stage s + 1's k block 0
#pragma unroll
                    for (int q = 0; q < NM; ++q) {
                        mfma_one(q, a1, b1);
#pragma unroll
                        for (int f = q * PF2; f < (q + 1) * PF2 && f < NV + NR; ++f) {
                            if (f < NV) issue_piece(f, ib); else read_one(f - NV, nsb, 0, a0, b0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    buf = nbuf;
                    lapt(3);
                }
                chunk_end();
                lapt(1);
            }
        }
        for (; s < nstage; ++s) {
            const char *sb = lds + buf * G::STAGE_BYTES;
            int nbuf = buf + 1; if (nbuf == NS) nbuf = 0;
            read_frags(sb, 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);             // the reads go out BEFORE the MFMAs they overlap with (left alone, the scheduler sinks them behind)
            mfma_block(a0, b0);
            lapt(0);
            if (++in_chunk == c) { in_chunk = 0; chunk_end(); }
            lapt(1);
            if (s + 1 < nstage) {
                // stage s + 1 must have landed; younger stages still in flight: s + 2 .. min(s + NS - 2, nstage - 1)
                if (s + NS - 2 < nstage) wait_vm<(NS - 3) * G::PPW>();
                else if (NS > 4 && s + NS - 3 < nstage) wait_vm<(NS > 4 ? NS - 4 : 0) * G::PPW>();
                else wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
            lapt(2);
            if (s + NS - 1 < nstage) { int ib = buf - 1; if (ib < 0) ib += NS; issue(ib); }
            if (s + 1 < nstage) read_frags(lds + nbuf * G::STAGE_BYTES, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(a1, b1);
            lapt(3);
            if (++in_chunk == c) { in_chunk = 0; chunk_end(); }
            lapt(1);
            buf = nbuf;
        }
    }
#ifdef APRIL_GEMM_TRACE
    tr_acc[4] = __builtin_amdgcn_s_memtime() - tr_start - tr_acc[5];
    tr_t = __builtin_amdgcn_s_memtime();
#endif

    if constexpr (ONE_CHAIN) { APRIL_TILE_EACH(res[mt][nt] = acc[mt][nt]) }      // the one chain (EPI_XPART: its y half, scaled)
    else if (EPI == EPI_XPART) { APRIL_TILE_EACH(res[mt][nt] = S[mt][nt]) }      // (c0 + c1) * scale: half a slab, by design
    // ---- the workgroup's sums -> LDS plane (each wave owns its columns; no cross-wave addition) -> 4-column quads per thread
    __syncthreads();                                       // the last stage has been read by every wave
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[((wm * MTW + mt) * 16 + kq * 4 + r) * LDR + (wn * NTW + nt) * 16 + mrow] = res[mt][nt][r];
    __syncthreads();
    f32x4 v[QPT];
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int q = threadIdx.x + i * NTH;
        v[i] = q < NQ ? *reinterpret_cast<const f32x4 *>(red + (q / QROW) * LDR + (q % QROW) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if (NEED_SCL && EPI != EPI_LSTM && EPI != EPI_XPART) compute_scl();

    if (EPI == EPI_XPART) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int m = m0 + q / QROW;
            if (q < NQ && m < g.M) *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n0 + (q % QROW) * 4) = v[i];
        }
    } else if (EPI == EPI_PARTIAL) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int m = m0 + q / QROW;
            if (q < NQ && m < g.M)
                *reinterpret_cast<f32x4 *>(g.out + ((size_t)zg * g.m_stride + m) * g.N + n0 + (q % QROW) * 4) = v[i];
        }
    } else if (EPI == EPI_HR) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int m = m0 + q / QROW, n = n0 + (q % QROW) * 4;
            if (e_ok[i]) {
                const float rs = scl[q / QROW];
                const f32x4 o = e_res[i] * rs + v[i];
                *reinterpret_cast<f32x4 *>(g.state + (size_t)e_slot[i] * g.ld_state + n) = v[i];
                *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n) = o;
                if (g.state16) *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(g.state16) + (size_t)e_slot[i] * g.ld_state + n) = to_h4(v[i]);
                if (g.out16) *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(g.out16) + (size_t)m * g.ldo + n) = to_h4(o);
            }
        }
    } else if (EPI == EPI_RESID_SSQ) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int m = m0 + q / QROW, n = n0 + (q % QROW) * 4;
            const bool ok = e_ok[i];
            f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                y = v[i] + e_bias[i];
                if (g.resid) y = e_res[i] + y;
                *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n) = y;
                if (g.out16) *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(g.out16) + (size_t)m * g.ldo + n) = to_h4(y);
            }
            const float ss = granule_ssq(y);               // all lanes take part in the shuffles
            if (ok && (q & 7) == 0) g.ssq_out[(size_t)m * (g.N / SSQ_COLS) + n / SSQ_COLS] = ss;
        }
    } else if (EPI == EPI_SLOT_STORE) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int n = n0 + (q % QROW) * 4;
            if (e_ok[i])
                *reinterpret_cast<f32x4 *>(g.out + (size_t)e_slot[i] * g.ldo + n) = NEED_SCL ? v[i] * scl[q / QROW] + e_bias[i] : v[i] + e_bias[i];
        }
    } else if (EPI == EPI_BIAS_DSWISH) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int m = m0 + q / QROW, n = n0 + (q % QROW) * 4;
            if (q < NQ && m < g.M) {
                const f32x4 y = v[i] + l_bias[i];
                f32x4 o;
                o.x = y.x * fast_sigmoid(y.x - 1.0f); o.y = y.y * fast_sigmoid(y.y - 1.0f);
                o.z = y.z * fast_sigmoid(y.z - 1.0f); o.w = y.w * fast_sigmoid(y.w - 1.0f);
                if (g.out) *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n) = o;
                if (g.out16) *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(g.out16) + (size_t)m * g.ldo + n) = to_h4(o);
            }
        }
    } else {   // EPI_LSTM: the quad = gates i, f, g, o of one hidden unit; the BasicNorm scale of the y half was folded in after chunk 1
        // (every load this epilogue depends on was issued before the K loop: nothing below waits on memory while stores are in flight)
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const f32x4 gt = v[i] + l_bias[i];
            const float c_new = fast_sigmoid(gt.y) * l_cprev[i] + fast_sigmoid(gt.x) * fast_tanh(gt.z);
            const float u = fast_sigmoid(gt.w) * fast_tanh(c_new);
            if (l_ok[i]) {
                *l_cptr[i] = c_new;
                if (g.out) g.out[(size_t)l_m[i] * g.ldo + l_unit[i]] = u;
                if (g.out16) reinterpret_cast<_Float16 *>(g.out16)[(size_t)l_m[i] * g.ldo + l_unit[i]] = (_Float16)u;
            }
        }
    }
#ifdef APRIL_GEMM_TRACE
    lapt(6);
    if (g.trace && wave == 0 && lane == 0) {
        const size_t wg = blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        for (int i = 0; i < 7; ++i) g.trace[wg * 8 + i] = tr_acc[i];
        g.trace[wg * 8 + 7] = (unsigned long long)nstage;
    }
#endif
}

template <int MT, int EPI, int WT, int NT = 4, int NWM = 2, int NWN = 2, int NSB = TILE_STAGES>
__global__ __launch_bounds__(64 * NWM * NWN, NWM * NWN == 4 ? 2 : 1) void gemm_tile_kernel(GemmArgs g)
{
    if constexpr (EPI == EPI_LSTM) stamp_begin(g.stamp, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
    gemm_tile_body<MT, EPI, WT, NT, NWM, NWN, NSB>(g, (int)blockIdx.z);
    if constexpr (EPI == EPI_LSTM) stamp_end(g.stamp, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// n independent same-shape problems in one launch (see gemm_f32_zkernel): blockIdx.z / zdiv picks the argument block
template <int MT, int EPI, int WT, int NT = 4, int NWM = 2, int NWN = 2, int NSB = TILE_STAGES>
__global__ __launch_bounds__(64 * NWM * NWN, NWM * NWN == 4 ? 2 : 1) void gemm_tile_zkernel(const GemmArgs *__restrict__ zargs, int zdiv)
{
    const int zl = (int)blockIdx.z / zdiv;
    const GemmArgs g = zargs[zl];
    if constexpr (EPI == EPI_LSTM) stamp_begin(g.stamp, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
    gemm_tile_body<MT, EPI, WT, NT, NWM, NWN, NSB>(g, (int)blockIdx.z - zl * zdiv);
    if constexpr (EPI == EPI_LSTM) stamp_end(g.stamp, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

template <class G> size_t tile_lds_bytes(const GemmArgs &g)
{
    const int sg = g.epi == EPI_HR ? g.r_scale.groups : ((g.epi == EPI_SLOT_STORE || g.epi == EPI_LSTM || g.epi == EPI_XPART) && g.x_scale.ssq ? g.x_scale.groups : 0);
    return (size_t)G::LDS_MAIN + (size_t)(G::BM + (sg ? G::BM * (sg + 1) : 0)) * sizeof(float);
}

// fp16 operands: a stage is a few dozen SIMD cycles of MFMA against 12 .. 32 KB of DMA, so the depth of the DMA pipeline decides:
// four stage buffers (two to three stages in flight); fp32 stages are MFMA-bound and measured the same with three (less LDS)
template <int MT, int EPI, int WT, int NT = 4, int NWM = 2, int NWN = 2>
void launch_tile_one(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
#ifndef APRIL_TILE_STAGES_F16
#define APRIL_TILE_STAGES_F16 4
#endif
#ifndef APRIL_TILE_STAGES_F16_MT2
#define APRIL_TILE_STAGES_F16_MT2 4
#endif
    // (32-row four-wave tiles: 12 KB per stage, so two workgroups per CU could hold six each; 128 x 192 tiles: 40 KB per stage, three fit the LDS)
    constexpr int NSB = (WT && NT <= 8) ? ((MT == 2 && NWM * NWN == 4) ? APRIL_TILE_STAGES_F16_MT2 : APRIL_TILE_STAGES_F16) : TILE_STAGES;
    using G = TileGeom<MT, NT, NWM, NWN, NSB>;
    const int zdiv = g.kz / g.zs;
    dim3 grid((unsigned)(g.N / G::BN), (unsigned)((g.M + G::BM - 1) / G::BM), (unsigned)(zdiv * std::max(1, n)));
    const size_t lds = tile_lds_bytes<G>(g);
    // dynamic LDS beyond 64 KB has to be announced, per instantiation AND per device (one engine per GPU, each with its own
    // stepping thread: a bit per device id, set after the attribute calls)
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tile_kernel<MT, EPI, WT, NT, NWM, NWN, NSB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tile_zkernel<MT, EPI, WT, NT, NWM, NWN, NSB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    if (dev_args) APRIL_LAUNCH((gemm_tile_zkernel<MT, EPI, WT, NT, NWM, NWN, NSB>), grid, dim3(G::NTH), lds, s, dev_args, zdiv);
    else APRIL_LAUNCH((gemm_tile_kernel<MT, EPI, WT, NT, NWM, NWN, NSB>), grid, dim3(G::NTH), lds, s, g);
}

template <int MT, int WT>
bool dispatch_tile(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    switch (g.epi) {
    case EPI_PARTIAL: launch_tile_one<MT, EPI_PARTIAL, WT>(g, dev_args, n, s); return true;
    case EPI_HR: launch_tile_one<MT, EPI_HR, WT>(g, dev_args, n, s); return true;
    case EPI_RESID_SSQ: launch_tile_one<MT, EPI_RESID_SSQ, WT>(g, dev_args, n, s); return true;
    case EPI_SLOT_STORE: launch_tile_one<MT, EPI_SLOT_STORE, WT>(g, dev_args, n, s); return true;
    case EPI_LSTM: launch_tile_one<MT, EPI_LSTM, WT>(g, dev_args, n, s); return true;
    case EPI_BIAS_DSWISH: launch_tile_one<MT, EPI_BIAS_DSWISH, WT>(g, dev_args, n, s); return true;
    case EPI_XPART: if constexpr (WT == 1) { launch_tile_one<MT, EPI_XPART, WT>(g, dev_args, n, s); return true; } else return false;
    default: return false;
    }
}

}  // namespace

// launch of a GEMM whose plan (kernels_gemm.hip) chose GM_TILE: g.zs slabs per workgroup, tile rows 16 * mt
void launch_gemm_tile(const GemmArgs &g, int mt, int nt, const GemmArgs *dev_args, int n, hipStream_t s)
{
    bool ok = false;
    if (mt == 4 && nt == 8) {        // 64 x 128, eight waves (wave tile 32 x 32): the fp16 N = d_model GEMMs (projection, FFN down)
        if (g.wt == 1 && g.epi == EPI_PARTIAL) { launch_tile_one<4, EPI_PARTIAL, 1, 8, 2, 4>(g, dev_args, n, s); ok = true; }
        else if (g.wt == 1 && g.epi == EPI_HR) { launch_tile_one<4, EPI_HR, 1, 8, 2, 4>(g, dev_args, n, s); ok = true; }
        else if (g.wt == 1 && g.epi == EPI_RESID_SSQ) { launch_tile_one<4, EPI_RESID_SSQ, 1, 8, 2, 4>(g, dev_args, n, s); ok = true; }
    }
    else if (nt == 6) {                   // 64 x 96 / 32 x 96, four waves (wave tile 32 x 48 / 16 x 48): the fp16 N = d_model GEMMs where N is a multiple of 96 (plan_tile)
        if (g.wt == 1 && mt == 4) {
            if (g.epi == EPI_PARTIAL) { launch_tile_one<4, EPI_PARTIAL, 1, 6>(g, dev_args, n, s); ok = true; }
            else if (g.epi == EPI_HR) { launch_tile_one<4, EPI_HR, 1, 6>(g, dev_args, n, s); ok = true; }
            else if (g.epi == EPI_RESID_SSQ) { launch_tile_one<4, EPI_RESID_SSQ, 1, 6>(g, dev_args, n, s); ok = true; }
        } else if (g.wt == 1 && mt == 2) {
            if (g.epi == EPI_PARTIAL) { launch_tile_one<2, EPI_PARTIAL, 1, 6>(g, dev_args, n, s); ok = true; }
            else if (g.epi == EPI_HR) { launch_tile_one<2, EPI_HR, 1, 6>(g, dev_args, n, s); ok = true; }
            else if (g.epi == EPI_RESID_SSQ) { launch_tile_one<2, EPI_RESID_SSQ, 1, 6>(g, dev_args, n, s); ok = true; }
        }
    }
    else if (mt == 8 && nt == 12) {       // 128 x 192, eight waves (wave tile 64 x 48): 77 flop per operand byte; N a multiple of 192
        if (g.wt == 1 && g.epi == EPI_LSTM) { launch_tile_one<8, EPI_LSTM, 1, 12, 2, 4>(g, dev_args, n, s); ok = true; }
        else if (g.wt == 1 && g.epi == EPI_BIAS_DSWISH) { launch_tile_one<8, EPI_BIAS_DSWISH, 1, 12, 2, 4>(g, dev_args, n, s); ok = true; }
    }
    else if (mt == 8) {                   // 128 x 128, eight waves: the fp16 gates / FFN-up GEMMs
        if (g.wt == 1 && g.epi == EPI_LSTM) { launch_tile_one<8, EPI_LSTM, 1, 8, 2, 4>(g, dev_args, n, s); ok = true; }
        else if (g.wt == 1 && g.epi == EPI_BIAS_DSWISH) { launch_tile_one<8, EPI_BIAS_DSWISH, 1, 8, 2, 4>(g, dev_args, n, s); ok = true; }
        else if (g.wt == 1 && g.epi == EPI_XPART) { launch_tile_one<8, EPI_XPART, 1, 8, 2, 4>(g, dev_args, n, s); ok = true; }
        else if (g.wt == 0 && g.epi == EPI_LSTM) { launch_tile_one<8, EPI_LSTM, 0, 8, 2, 4>(g, dev_args, n, s); ok = true; }
        else if (g.wt == 0 && g.epi == EPI_BIAS_DSWISH) { launch_tile_one<8, EPI_BIAS_DSWISH, 0, 8, 2, 4>(g, dev_args, n, s); ok = true; }
    }
    else if (g.wt == 1) { if (mt == 4) ok = dispatch_tile<4, 1>(g, dev_args, n, s); else if (mt == 2) ok = dispatch_tile<2, 1>(g, dev_args, n, s); }
    else if (mt == 4) ok = dispatch_tile<4, 0>(g, dev_args, n, s);
    else if (mt == 2) ok = dispatch_tile<2, 0>(g, dev_args, n, s);
    if (!ok) { fprintf(stderr, "libapril(mi355x): launch_gemm_tile: no kernel for epi %d tile %d x %d (wt %d)\n", g.epi, 16 * mt, 16 * nt, g.wt); abort(); }
}

}  // namespace aprilx
