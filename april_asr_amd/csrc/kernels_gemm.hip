// MFMA "skinny" GEMM for gfx950: out[M,N] = A[M,K] x W[K,N], M = sessions
// stepped together (1 .. thousands), W streamed once per workgroup column from a
// pre-packed HBM layout (see kernels.h).  v_mfma_f32_16x16x4_f32 is an exact
// in-order fp32 FMA chain, so results do not depend on M or on the tile shape.
//
// Workgroup = 256 threads = 4 waves which split K; partial tiles meet in LDS and the
// epilogue runs on the summed tile.  Two schedules of the same canonical summation
// (kernels.h): GM_SLAB (waves split every slab, one LDS meet per slab, slabs across
// workgroups -> partial planes; the HBM-bound small-batch form) and GM_FULLK (every wave
// owns a contiguous K quarter, chunk chains and the slab tree are folded in registers,
// ONE LDS meet, no partial planes; the row work runs in the epilogue).  Epilogues:
//   EPI_PARTIAL      slab-tree sums -> workspace (row kernels finish bias/residual/...)
//   EPI_LSTM         LSTM cell: sigma/tanh gates, c' written in place, u = sigma(o) tanh(c')
//   EPI_BIAS_DSWISH  y = acc + b ; y * sigmoid(y - 1)
//   EPI_HR           h state write + residual (LSTM projection)
//   EPI_RESID_SSQ    bias + residual + sum-of-squares partials (BasicNorm is applied by the consumer's A prologue)
//   EPI_SLOT_STORE   bias + store into the session's slot row (encoder_proj, decoder_proj)
// WT = 1 reads fp16 weights and rounds A to fp16 on load (v_mfma_f32_16x16x16_f16, fp32
// accumulation, same lane<->k mapping and summation structure).  The K loop exists twice with
// identical arithmetic: compiler-scheduled C++ (all shapes) and a generated hand-scheduled
// version for the fused-epilogue 64x64 / 64x32 fp32 tiles (gemm_mainloop_asm.inc).
// Replaces the ORT MatMul/Gemm nodes of the encoder/decoder/joiner graphs
// (reference call sites src/april_session.c:145,160,176).
#include "kernels.h"
#include "device_utils.h"
// the head of the hand-scheduled K loop on an 8-byte boundary: every instruction of the loop is an 8-byte encoding, so the phase of the
// head is the phase of all of them (MI355X_MICROARCH.md: a hand-written stream loses up to 13 % at shifts of 4 mod 8; the round-5 build
// had the gates kernel's head at 4 mod 8)
#ifndef APRIL_ASM_LOOP_ALIGN
#define APRIL_ASM_LOOP_ALIGN ".p2align 3\n"
#endif
#include "gemm_mainloop_asm.inc"
#ifndef APRIL_ASM_NB
#define APRIL_ASM_NB 2      // operand buffers of the hand-scheduled K loop: 2 keeps two workgroups per CU, 3 needs > 256 registers
#endif
#include <algorithm>
#include <cstdlib>

namespace aprilx {


template <int MT, int NT, int NW = 4>
struct TileCfg {
    static constexpr int BM = MT * 16, BN = NT * 16, LDR = BN + 4;
    static constexpr int LDS_FLOATS = NW * BM * LDR;      // one partial plane per wave; BM row scales follow (LDS_FLOATS + BM floats are requested)
};

using h4 = __attribute__((ext_vector_type(4))) _Float16;
template <int WT> struct WQuad { using type = f32x4; };
template <> struct WQuad<1> { using type = h4; };

// ASM = 1: the K loop is the hand-scheduled one (fixed operand registers v112..v175, accumulators in AGPRs); a separate
// instantiation, so that neither loop's registers weigh on the other (both forms must stay within 256 registers: two
// workgroups per CU).  The compiler-scheduled 64-row tiles are told so (second launch-bound = waves per SIMD).
// NW = waves per workgroup: 4, or 8 on the full-K schedule of a kz = 8 layer (every wave owns ONE slab).  With only 16 x 32
// outputs per workgroup a wave has two accumulator tiles, i.e. chains of four DEPENDENT MFMAs back to back, which issue at
// half rate (measured: 520 cycles per k-block instead of 256); a second wave on the same SIMD fills the gaps.
template <int MT, int NT, int EPI, int AOP, int WT, int MODE, int ASM, int NW = 4>
__device__ __forceinline__ void gemm_body(const GemmArgs &g, const int zg, const unsigned wg_linear, const int bx, const int by, const bool one_m_block)
{
    using Cfg = TileCfg<MT, NT, NW>;
    constexpr int NTH = NW * 64;
    static_assert(NW == 4 || (NW == 8 && MODE == GM_FULLK), "8 waves only on the full-K schedule");
    using BQ = typename WQuad<WT>::type;       // one lane's four consecutive k values of a weight tile
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr bool FULLK = MODE == GM_FULLK;
    constexpr bool ROW_EPI = EPI == EPI_HR || EPI == EPI_RESID_SSQ || EPI == EPI_SLOT_STORE;
    // the slab tree (levels of pairwise sums) is only needed where a workgroup can own more than one slab
    constexpr bool TREE = !FULLK && (EPI == EPI_PARTIAL || ROW_EPI);       // (EPI_LSTM, EPI_BIAS_DSWISH, EPI_XPART: kz = 1)

    if (g.run_flag && *g.run_flag != g.run_gen) return;  // a joiner / decoder round nobody needs (all rows resolved earlier)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // measurement only (tools/gemm_bench built with -DAPRIL_GEMM_TRACE, run with GEMM_TRACE=1): wave 0 stamps s_memtime at
    // phase boundaries (uniform branch, all lanes store the same value); compiled out of the product
#ifdef APRIL_GEMM_TRACE
    auto stamp = [&](int i) { if (g.trace && wave == 0) g.trace[(size_t)wg_linear * 8 + i] = __builtin_amdgcn_s_memtime(); };
#else
    auto stamp = [](int) {};
#endif
    stamp(0);
#ifdef APRIL_GEMM_TRACE
    if (g.trace && wave == 0) {   // where this workgroup runs: HW_ID (wave/simd/cu/sh/se fields) and XCC_ID
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        g.trace[(size_t)wg_linear * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    // XCD-aware mapping: consecutive blockIdx.x land on different XCDs, so keep the
    // M-blocks that share one weight column on the same XCD (same x mod 8).
    const int nt0 = bx * NT;
    const int m0 = by * Cfg::BM;
    // zg (GM_SLAB): this workgroup owns slabs [zg*zs, (zg+1)*zs)
    first_round_skew(g.skew, (unsigned)wg_linear, (unsigned)g.skew_wgs);      // (measurement form, off by default: device_utils.h)
    const int KB = g.K >> 4;

    const int mrow = lane & 15, kq = lane >> 4;

    // Addressing = uniform 64-bit base (SGPRs; advances with the k block) + one 32-bit byte offset per lane and
    // m-tile (row start + this lane's k quarter), i.e. the saddr form of global_load: no 64-bit vector adds in the loop.
    // All operands are far below 4 GiB per array.
    // after the LDS meet every thread owns QPT groups of 4 consecutive columns ("quads") of the tile
    constexpr int QROW = Cfg::BN / 4, NQ = Cfg::BM * QROW, QPT = (NQ + NTH - 1) / NTH;
    // K blocks: KB = K/16 is a multiple of 4*kz (checked on the host): chunk = c blocks.
    //   GM_SLAB : slab z, wave w -> blocks [(4z + w) c, (4z + w + 1) c); a workgroup walks zs consecutive slabs.
    //   GM_FULLK: wave w of NW -> blocks [w T, (w + 1) T), T = 4 kz c / NW, chunk after chunk (kz / NW whole slabs).
    const int c = g.debug == 1 ? 0 : KB / (4 * g.kz);
    const bool wave_on = ((g.wave_mask >> wave) & 1) != 0;          // layer-major split of the gate GEMM: half of the waves sit a launch out
    const int T = !wave_on ? 0 : (FULLK ? (4 * g.kz / NW) * c : g.zs * c);   // blocks this wave processes in total
    const int first_kb = FULLK ? wave * T : (4 * (zg * g.zs) + wave) * c;

    // BasicNorm scales of the tile's rows, once per workgroup: the rows' sum-of-squares partials make ONE trip from global
    // memory (every lane fetching its own rows' partials cost the gate GEMM 8 us).  The loads are issued here, first thing,
    // and stay in registers during the K loop; after the meet 64 threads add them up through LDS (see the epilogue).
    const RowScale &rsc = EPI == EPI_HR ? g.r_scale : g.x_scale;
    const bool NEED_SCL = (EPI == EPI_HR || EPI == EPI_LSTM || EPI == EPI_SLOT_STORE || EPI == EPI_XPART) && rsc.ssq != nullptr;     // uniform
    float *scl = red + Cfg::LDS_FLOATS;
    float stg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // TPR threads share a row, each holds up to 4 consecutive partials of it (no runtime divisions on the way in)
    constexpr int TPR = NTH / Cfg::BM;
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

    // Row indirections (session slots) of all m-tiles are loaded back to back under uniform conditions: ONE memory round trip
    // before the first operand load instead of one per m-tile (the per-tile form compiled to load / s_waitcnt vmcnt(0) four
    // times in a row: ~1.5 us of a 22 us gate GEMM at 256 rows).  Padding rows recompute the last row; never stored.
    uint32_t aoff0[MT], aoff1[MT], aoffb[AOP == AOP_TANH_ADD ? MT : 1];
    int arow[MT], r0v[MT], r1v[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { const int row = m0 + mt * 16 + mrow; arow[mt] = row >= g.M ? g.M - 1 : row; }
    if (g.aidx0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) r0v[mt] = g.aidx0[arow[mt]];
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) r0v[mt] = arow[mt];
    }
    if (g.K1 > 0 && g.aidx1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) r1v[mt] = g.aidx1[arow[mt]];
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) r1v[mt] = arow[mt];
    }
    // EPI_LSTM: the slots of this thread's output rows (see below), fetched in the same round trip
    int qslot[EPI == EPI_LSTM ? QPT : 1];
    if (EPI == EPI_LSTM) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            int r = m0 + (int)(threadIdx.x + i * NTH) / QROW;
            if (r >= g.M) r = g.M - 1;
            qslot[i] = g.slot_idx[r];
        }
    }
    if (AOP == AOP_TANH_ADD) {
        int rbv[MT];
        if (g.same_idx_b) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rbv[mt] = r0v[mt];
        } else if (g.aidx0b) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rbv[mt] = g.aidx0b[arow[mt]];
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rbv[mt] = arow[mt];
        }
        if (g.ctx_state) {
            GreedyState st[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) st[mt] = g.ctx_state[rbv[mt]];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rbv[mt] = st[mt].ctx0 * g.ctx_vocab + st[mt].ctx1;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aoffb[mt] = (uint32_t)(((size_t)rbv[mt] * g.lda0 + kq * 4) * sizeof(float));
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        aoff0[mt] = (uint32_t)(((size_t)r0v[mt] * g.lda0 + kq * 4) * sizeof(float));
        aoff1[mt] = g.K1 > 0 ? (uint32_t)(((size_t)r1v[mt] * g.lda1 + kq * 4) * sizeof(float)) : 0u;
    }
    const uint32_t boff = (uint32_t)lane * sizeof(BQ);
    const bool stream_once = one_m_block;      // weights read by exactly one workgroup: bypass-friendly loads

    // pairwise (balanced-tree) slab accumulation (GM_SLAB): level b holds the sum of 2^b consecutive slabs.  A workgroup
    // that owns zs = 2^t slabs needs t levels; the 64x64 tile is capped at zs = 4 (2 levels, 32 registers) so that it stays
    // within 256 registers and two workgroups share a CU (launch_gemm / gemm_partials apply the same cap)
    constexpr int NLVL = !TREE ? 1 : ((MT == 4 && NT == 4) ? 2 : 3);
    f32x4 lvl[NLVL][QPT];
    int top = 0;
    if (TREE) while ((1 << top) < g.zs) ++top;
    constexpr int PLANE = Cfg::BM * Cfg::LDR;
    auto summed4 = [&](int o) {                        // GM_SLAB: ((p0+p1)+p2)+p3 of four consecutive columns; GM_FULLK: balanced tree over the waves' results
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + o), p1 = *reinterpret_cast<const f32x4 *>(red + PLANE + o);
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(red + 2 * PLANE + o), p3 = *reinterpret_cast<const f32x4 *>(red + 3 * PLANE + o);
        if constexpr (NW == 8) {
            const f32x4 p4 = *reinterpret_cast<const f32x4 *>(red + 4 * PLANE + o), p5 = *reinterpret_cast<const f32x4 *>(red + 5 * PLANE + o);
            const f32x4 p6 = *reinterpret_cast<const f32x4 *>(red + 6 * PLANE + o), p7 = *reinterpret_cast<const f32x4 *>(red + 7 * PLANE + o);
            return ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
        }
        if (FULLK) return (p0 + p1) + (p2 + p3);
        return ((p0 + p1) + p2) + p3;
    };

    auto load_a = [&](int kb, f32x4 (&a)[MT]) {
        const bool seg0 = kb * 16 < g.K0;                    // uniform: segment lengths are multiples of 16
        const char *sb = reinterpret_cast<const char *>(seg0 ? g.a0 : g.a1) + (ptrdiff_t)(seg0 ? kb * 16 : kb * 16 - g.K0) * (ptrdiff_t)sizeof(float);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint32_t off = seg0 ? aoff0[mt] : aoff1[mt];
            f32x4 v = *reinterpret_cast<const f32x4 *>(sb + off);
            if (AOP == AOP_TANH_ADD) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(g.a0b) + (ptrdiff_t)kb * 64 + aoffb[mt]);
                v.x = fast_tanh(v.x + w.x); v.y = fast_tanh(v.y + w.y); v.z = fast_tanh(v.z + w.z); v.w = fast_tanh(v.w + w.w);
            }
            a[mt] = v;
        }
    };
    auto load_b = [&](int kb, BQ (&b)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const char *sb = reinterpret_cast<const char *>(g.wp) + ((size_t)(nt0 + nt) * KB + kb) * (64 * sizeof(BQ));
            const BQ *pw = reinterpret_cast<const BQ *>(sb + boff);
            b[nt] = stream_once ? __builtin_nontemporal_load(pw) : *pw;
        }
    };

    // Register-staged software pipeline: DEPTH k-blocks of loads are in flight per wave while the MFMAs of the
    // oldest stage issue.  Small batches (MT == 1) are HBM-bound weight streams and want many bytes in flight per
    // CU (6 x 4 waves x (1+NT) KiB); large tiles are MFMA-bound and need only enough depth to cover L2 latency.
    // The prefetch stream runs ahead ACROSS chunk and slab boundaries, so the pipeline never restarts inside a workgroup;
    // loads in the main loop are unconditional (the fetch position parks on the last block) so the loop body is
    // straight-line code and the compiler keeps counted s_waitcnt vmcnt(N) instead of draining.
#ifndef APRIL_DEPTH4
#define APRIL_DEPTH4 2
#endif
    // The full-K tiles are small (16..64 x 32) and read everything through L2: they are bound by bytes in flight per CU
    // (measured 9 TB/s of L2 traffic at 6 stages x 3 KB per wave), so they run deeper.
#ifndef APRIL_FULLK_DEPTH1
#define APRIL_FULLK_DEPTH1 6     // measured: 12 stages are slower (whr 8.7 -> 9.9 us, FFN-down 11.9 -> 13.4 us at 256 rows)
#endif
#ifndef APRIL_DEPTH1
#define APRIL_DEPTH1 6
#endif
#ifndef APRIL_FULLK_DEPTH2
#define APRIL_FULLK_DEPTH2 4
#endif
    constexpr int DEPTH = FULLK ? ((MT == 1) ? APRIL_FULLK_DEPTH1 : (MT == 2 ? APRIL_FULLK_DEPTH2 : 3)) : ((MT == 1) ? APRIL_DEPTH1 : (MT == 2 ? 3 : APRIL_DEPTH4));
    f32x4 a_st[DEPTH][MT];
    BQ b_st[DEPTH][NT];
    int ld_base = first_kb, ld_off = 0, ld_cnt = 0;
    auto ld_next = [&]() {
        const int kb = g.debug == 3 ? 0 : ld_base + ld_off;     // debug 3 (measurement): every block re-reads block 0 (cache-resident operands)
        if (ld_cnt + 1 < T) {
            ++ld_cnt;
            if (FULLK) ++ld_off;                                   // contiguous quarter
            else if (++ld_off == c) { ld_off = 0; ld_base += 4 * c; }
        }
        return kb;
    };
    f32x4 acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto compute = [&](const f32x4 (&a)[MT], const BQ (&b)[NT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 av = a[mt];
            if constexpr (WT == 1) {
                // the same 16 k values per lane as four fp32 k-steps, in one v_mfma_f32_16x16x16_f16 (fp32 accumulate)
                const h4 ah = {(_Float16)av.x, (_Float16)av.y, (_Float16)av.z, (_Float16)av.w};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, b[nt], acc[mt][nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[nt].w, acc[mt][nt], 0, 0, 0);
                }
            }
        }
    };

    // ---- chunk ends
    // GM_FULLK: the chunk chain is closed in registers.  S = ((c0+c1)+c2)+c3 is a slab; with two slabs per wave (kz = 8)
    // R = S_first + S_second is the first level of the slab tree; the wave's result (a partial tile) is in R.
    f32x4 S[FULLK ? MT : 1][FULLK ? NT : 1], R[FULLK ? MT : 1][FULLK ? NT : 1];
    int chunk_i = 0, slab_i = 0;
    auto fold = [&]() {
        if constexpr (FULLK) {
            if (chunk_i == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) S[mt][nt] = acc[mt][nt];
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) S[mt][nt] = S[mt][nt] + acc[mt][nt];
            }
            zero_acc();
            if (++chunk_i == 4) {
                chunk_i = 0;
                if (slab_i == 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) R[mt][nt] = S[mt][nt];
                } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) R[mt][nt] = R[mt][nt] + S[mt][nt];
                }
                ++slab_i;
            }
        }
    };
    // partial tiles -> LDS (red[wave][row][col])
    auto to_lds = [&](const f32x4 (&t)[MT][NT]) {
        float *mine = red + (size_t)wave * PLANE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mine[(mt * 16 + kq * 4 + r) * Cfg::LDR + nt * 16 + mrow] = t[mt][nt][r];
            // one m-tile (16 accumulator registers) at a time: left alone, the scheduler copies the whole accumulator
            // file out of the AGPRs first and the kernel no longer fits two workgroups per CU
            if (MT == 4) __builtin_amdgcn_sched_barrier(0);
        }
    };
    int slab_done = 0;                                 // GM_SLAB: slabs finished so far by this workgroup
    f32x4 v[QPT];                                      // this thread's quads of the workgroup's total (valid after the last meet)
    // GM_SLAB: quarter chains meet in LDS and are added ((p0+p1)+p2)+p3; slab sums are combined pairwise in slab order
    // (balanced tree), so owning 1, 2, 4 or 8 slabs per workgroup -- chosen from the batch size -- yields the same bits
    auto meet = [&]() {
        if (slab_done > 0) __syncthreads();            // previous slab's reads of red[] are done
        to_lds(acc);
        __syncthreads();
        if (TREE) {
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                const int q = threadIdx.x + i * NTH;
                v[i] = q < NQ ? summed4((q / QROW) * Cfg::LDR + (q % QROW) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            bool carry = true;
#pragma unroll
            for (int b = 0; b < NLVL; ++b) {
                if (carry && b < top) {
                    if ((slab_done >> b) & 1) {
#pragma unroll
                        for (int i = 0; i < QPT; ++i) v[i] = lvl[b][i] + v[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < QPT; ++i) lvl[b][i] = v[i];
                        carry = false;
                    }
                }
            }
        }
        ++slab_done;
    };
    auto chunk_end = [&]() {
        if constexpr (FULLK) fold();
        else { meet(); if (slab_done < g.zs) zero_acc(); }
    };

    // EPI_LSTM: the thread's QPT (row, unit) pairs are known up front; their dependent global loads (row -> slot ->
    // previous cell value) and the gate biases are issued BEFORE the MFMA stream so the epilogue never waits on HBM
    int qm[QPT], qunit[QPT], qo[QPT];
    bool qok[QPT];
    float *cptr[QPT];
    float cprev[QPT];
    f32x4 qb[QPT];
    if (EPI == EPI_LSTM) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int row = q / QROW, ul = q % QROW;
            qm[i] = m0 + row;
            qok[i] = q < NQ && qm[i] < g.M;
            const int n = nt0 * 16 + ul * 4;
            qunit[i] = n >> 2;
            qo[i] = row * Cfg::LDR + ul * 4;
            cptr[i] = g.c_state + (size_t)qslot[i] * g.hidden + qunit[i];       // (padding rows point at the last row's cell: read, never stored)
            qb[i] = *reinterpret_cast<const f32x4 *>(g.bias + n);
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) cprev[i] = *cptr[i];
    }

    // Row epilogues: what the epilogue reads besides this kernel's own sums (bias, residual rows, the rows' slots) does not
    // depend on the K loop -- fetched here, into registers, so that the epilogue does not pay two or three dependent L2 round
    // trips after the meet (projection 16x32 at 256 rows: ~1.5 us of 8.7)
    f32x4 e_bias[ROW_EPI ? QPT : 1], e_res[ROW_EPI ? QPT : 1];
    int e_slot[ROW_EPI ? QPT : 1];
    bool e_ok[ROW_EPI ? QPT : 1];
    if (ROW_EPI) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            int m = m0 + q / QROW;
            const int n = nt0 * 16 + (q % QROW) * 4;
            e_ok[i] = q < NQ && m < g.M;
            if (m >= g.M) m = g.M - 1;
            const int qn = q < NQ ? n : nt0 * 16;                  // (idle threads of small tiles read a valid column)
            e_slot[i] = (EPI != EPI_RESID_SSQ && g.slot_idx) ? g.slot_idx[m] : m;
            if (EPI == EPI_SLOT_STORE && g.row_mask && !g.row_mask[m]) e_ok[i] = false;
            e_bias[i] = (EPI != EPI_HR) ? *reinterpret_cast<const f32x4 *>(g.bias + qn) : f32x4{0.f, 0.f, 0.f, 0.f};
            e_res[i] = (EPI == EPI_HR || (EPI == EPI_RESID_SSQ && g.resid)) ? *reinterpret_cast<const f32x4 *>(g.resid + (size_t)m * g.ldr + qn) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    zero_acc();
    stamp(1);
    // fused-epilogue GEMMs only (one slab, 5..16 blocks per wave): the split-K GEMMs walk several short slabs per
    // workgroup and rely on the cross-slab prefetch of the compiler-scheduled loop below (measured: no gain there)
    if constexpr (ASM) {
        static_assert(!FULLK && MT == 4 && (NT == 4 || NT == 2) && WT == 0 && AOP == AOP_NONE && (EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH), "no hand-scheduled loop for this form");
        {
            // hand-scheduled K loop (tools/gen_gemm_asm.py): same blocks, same order, same accumulation chains
            uint32_t boffs[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) boffs[nt] = boff + (uint32_t)nt * (uint32_t)KB * 1024u;
            for (int z = 0; z < g.zs; ++z) {
                const int kb0 = (4 * (zg * g.zs + z) + wave) * c;
                int done = wave_on ? 0 : c;
                while (done < c) {
                    const int kb = kb0 + done;
                    const bool seg0 = kb * 16 < g.K0;
                    int n = c - done;
                    if (seg0 && g.K0 / 16 - kb < n) n = g.K0 / 16 - kb;
                    const char *ap = reinterpret_cast<const char *>(seg0 ? g.a0 : g.a1) + (ptrdiff_t)(seg0 ? kb * 16 : kb * 16 - g.K0) * 4;
                    const char *bp = reinterpret_cast<const char *>(g.wp) + ((size_t)nt0 * KB + kb) * 1024;
                    const uint32_t o0 = seg0 ? aoff0[0] : aoff1[0], o1 = seg0 ? aoff0[1] : aoff1[1];
                    const uint32_t o2 = seg0 ? aoff0[2] : aoff1[2], o3 = seg0 ? aoff0[3] : aoff1[3];
#define APRIL_ASM_IN_A [aoff0] "v"(o0), [aoff1] "v"(o1), [aoff2] "v"(o2), [aoff3] "v"(o3), [ap] "s"(ap), [bp] "s"(bp), [nblk] "s"(n)
#define APRIL_ASM_ACC16 [c0] "+a"(acc[0][0]), [c1] "+a"(acc[0][1]), [c2] "+a"(acc[0][2]), [c3] "+a"(acc[0][3]), \
                        [c4] "+a"(acc[1][0]), [c5] "+a"(acc[1][1]), [c6] "+a"(acc[1][2]), [c7] "+a"(acc[1][3]), \
                        [c8] "+a"(acc[2][0]), [c9] "+a"(acc[2][1]), [c10] "+a"(acc[2][2]), [c11] "+a"(acc[2][3]), \
                        [c12] "+a"(acc[3][0]), [c13] "+a"(acc[3][1]), [c14] "+a"(acc[3][2]), [c15] "+a"(acc[3][3])
#define APRIL_ASM_ACC8 [c0] "+a"(acc[0][0]), [c1] "+a"(acc[0][1]), [c2] "+a"(acc[1][0]), [c3] "+a"(acc[1][1]), \
                       [c4] "+a"(acc[2][0]), [c5] "+a"(acc[2][1]), [c6] "+a"(acc[3][0]), [c7] "+a"(acc[3][1])
                    if constexpr (NT == 4) {
                        {
#if APRIL_ASM_NB == 3
                            asm volatile(APRIL_MAINLOOP3_TEXT
#else
                            asm volatile(APRIL_MAINLOOP2_TEXT
#endif
                                : APRIL_ASM_ACC16
                                : APRIL_ASM_IN_A, [boff0] "v"(boffs[0]), [boff1] "v"(boffs[1]), [boff2] "v"(boffs[NT - 2]), [boff3] "v"(boffs[NT - 1])
#if APRIL_ASM_NB == 3
                                : APRIL_MAINLOOP3_CLOBBERS);
#else
                                : APRIL_MAINLOOP2_CLOBBERS);
#endif
                        }
                    } else {
                        {
                            asm volatile(APRIL_MAINLOOP2_NT2_TEXT
                                : APRIL_ASM_ACC8
                                : APRIL_ASM_IN_A, [boff0] "v"(boffs[0]), [boff1] "v"(boffs[1])
                                : APRIL_MAINLOOP2_NT2_CLOBBERS);
                        }
                    }
#undef APRIL_ASM_IN_A
#undef APRIL_ASM_ACC16
#undef APRIL_ASM_ACC8
                    done += n;
                }
                stamp(2);
                meet();
                stamp(3);
                if (slab_done < g.zs) zero_acc();
            }
        }
    }
    else if (T > 0) {
        {
            int first[DEPTH];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) first[s] = ld_next();
            // same issue order as the steady state (stage by stage, weights then activations): the compiler merges
            // the wait counts of the loop entry and the back edge, and any other order here makes every iteration
            // wait for loads of the NEXT stage (vmcnt(7..4) instead of vmcnt(11..8) at DEPTH 2)
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                load_b(first[s], b_st[s]); load_a(first[s], a_st[s]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        int in_chunk = 0, i = 0;
        for (; i + DEPTH <= T; i += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                compute(a_st[s], b_st[s]);
                // pin the refill right behind its stage's MFMAs: left alone, the scheduler sinks all refills to the
                // loop end and the next iteration waits out a full memory latency
                __builtin_amdgcn_sched_barrier(0);
                const int nk = ld_next();
                load_b(nk, b_st[s]); load_a(nk, a_st[s]);
                __builtin_amdgcn_sched_barrier(0);
                if (++in_chunk == c) { in_chunk = 0; chunk_end(); }
            }
        }
#pragma unroll
        for (int s = 0; s < DEPTH - 1; ++s)
            if (i + s < T) {
                compute(a_st[s], b_st[s]);
                if (++in_chunk == c) { in_chunk = 0; chunk_end(); }
            }
    } else {
        if (FULLK) { for (int z = 0; z < 4 * g.kz / NW; ++z) fold(); }
        else for (int z = 0; z < g.zs; ++z) meet();    // measurement mode without the main loop
    }

    if constexpr (!ASM) stamp(2);
    if constexpr (FULLK) {
        // the ONE meet of the full-K schedule: (R0+R1)+(R2+R3) = the canonical slab tree
        to_lds(R);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            v[i] = q < NQ ? summed4((q / QROW) * Cfg::LDR + (q % QROW) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    if (NEED_SCL) {
        // the rows' scales: partials (in registers since the first instruction of the kernel) -> LDS, 64 threads add them in
        // column order (the order of row_scale()); rows are padded to G + 1 floats (conflict-free column walks)
        const int G = rsc.groups;
        float *part = scl + Cfg::BM;
        if (staged) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < ppt && sj0 + k < G) part[srow * (G + 1) + sj0 + k] = stg[k];
        } else {
            for (int i = threadIdx.x; i < Cfg::BM * G; i += NTH) {
                int r = m0 + i / G;
                if (r >= g.M) r = g.M - 1;
                part[(i / G) * (G + 1) + i % G] = rsc.ssq[(size_t)r * G + i % G];
            }
        }
        __syncthreads();
        if (threadIdx.x < Cfg::BM) {
            float t = 0.0f;
            for (int j = 0; j < G; ++j) t += part[threadIdx.x * (G + 1) + j];
            scl[threadIdx.x] = __builtin_amdgcn_rsqf(t * rsc.inv_n + rsc.eps);
        }
        __syncthreads();
    }
    if constexpr (!ASM) stamp(3);
    if (EPI == EPI_PARTIAL) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int m = m0 + q / QROW;
            if (q < NQ && m < g.M)
                *reinterpret_cast<f32x4 *>(g.out + ((size_t)(FULLK ? 0 : zg) * g.m_stride + m) * g.N + nt0 * 16 + (q % QROW) * 4) = v[i];
        }
    } else if (EPI == EPI_HR) {
        // LSTM projection: h' = acc goes to the session's state row, the layer continues with x + h' where
        // x = y * scale(y) is the (never materialised) BasicNorm output of the previous layer
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int m = m0 + q / QROW, n = nt0 * 16 + (q % QROW) * 4;
            if (e_ok[i]) {
                const float rs = scl[q / QROW];
                *reinterpret_cast<f32x4 *>(g.state + (size_t)e_slot[i] * g.ld_state + n) = v[i];
                *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n) = e_res[i] * rs + v[i];
            }
        }
    } else if (EPI == EPI_RESID_SSQ) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int m = m0 + q / QROW, n = nt0 * 16 + (q % QROW) * 4;
            const bool ok = e_ok[i];
            f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                y = v[i] + e_bias[i];
                if (g.resid) y = e_res[i] + y;
                *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n) = y;
            }
            const float ss = granule_ssq(y);           // all lanes take part in the shuffles
            if (ok && (q & 7) == 0) g.ssq_out[(size_t)m * (g.N / SSQ_COLS) + n / SSQ_COLS] = ss;
        }
    } else if (EPI == EPI_SLOT_STORE) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int n = nt0 * 16 + (q % QROW) * 4;
            if (e_ok[i])
                *reinterpret_cast<f32x4 *>(g.out + (size_t)e_slot[i] * g.ldo + n) = NEED_SCL ? v[i] * scl[q / QROW] + e_bias[i] : v[i] + e_bias[i];
        }
    } else if (EPI == EPI_XPART) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int row = q / QROW, col = (q % QROW) * 4;
            const int m = m0 + row, n = nt0 * 16 + col;
            if (q < NQ && m < g.M) {
                const int o = row * Cfg::LDR + col;
                const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + o), p1 = *reinterpret_cast<const f32x4 *>(red + PLANE + o);
                *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n) = NEED_SCL ? (p0 + p1) * scl[row] : (p0 + p1);
            }
        }
    } else if (EPI == EPI_BIAS_DSWISH) {
        // biases of all of this thread's quads first, settled once: a load inside the loop would make every quad wait for the
        // previous quad's store (see EPI_LSTM below)
        f32x4 bq[QPT];
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            bq[i] = *reinterpret_cast<const f32x4 *>(g.bias + nt0 * 16 + (q < NQ ? (q % QROW) * 4 : 0));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * NTH;
            const int row = q / QROW, col = (q % QROW) * 4;
            const int m = m0 + row, n = nt0 * 16 + col;
            if (q < NQ && m < g.M) {
                const f32x4 y = summed4(row * Cfg::LDR + col) + bq[i];
                f32x4 o;
                o.x = y.x * fast_sigmoid(y.x - 1.0f); o.y = y.y * fast_sigmoid(y.y - 1.0f);
                o.z = y.z * fast_sigmoid(y.z - 1.0f); o.w = y.w * fast_sigmoid(y.w - 1.0f);
                *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n) = o;
            }
        }
    } else {   // EPI_LSTM: each 4-column group = gates i,f,g,o of one hidden unit
        // (layer-major) the input halves of all of this thread's quads, fetched up front: a load inside the loop below makes
        // the compiler wait for ALL outstanding memory operations at the loop's merge points, i.e. for the previous quad's
        // STORES (~0.5 us each, three times per 64x64 tile) -- also on the streaming path, which loads nothing here
        f32x4 pq[QPT];
        if (g.p_add) {
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                int m = qm[i];
                if (m >= g.M) m = g.M - 1;
                pq[i] = *reinterpret_cast<const f32x4 *>(g.p_add + (size_t)m * g.ldp + qunit[i] * 4);
            }
        }
        // every load this epilogue depends on (previous cell values, biases, input halves) has been issued long ago: settle them
        // here, once, so that nothing in the loop waits on the memory counter while the previous quad's stores are in flight
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) only (expcnt / lgkmcnt fields at their maxima)
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            // x = y * scale(y) entered the GEMM as y: waves 0 and 1 hold the input half of the sum, which takes the row's scale here
            f32x4 gt;
            if (g.p_add || NEED_SCL) {
                const int o = qo[i];
                const f32x4 p2 = *reinterpret_cast<const f32x4 *>(red + 2 * PLANE + o), p3 = *reinterpret_cast<const f32x4 *>(red + 3 * PLANE + o);
                f32x4 xin;
                if (g.p_add) {           // layer-major: the input half was computed for all time steps at once (EPI_XPART)
                    xin = pq[i];
                } else {
                    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + o), p1 = *reinterpret_cast<const f32x4 *>(red + PLANE + o);
                    xin = (p0 + p1) * scl[(threadIdx.x + i * NTH) / QROW];
                }
                gt = ((xin + p2) + p3) + qb[i];
            } else {
                gt = summed4(qo[i]) + qb[i];
            }
            const float c_new = fast_sigmoid(gt.y) * cprev[i] + fast_sigmoid(gt.x) * fast_tanh(gt.z);
            const float u = g.debug == 2 ? gt.x + gt.y + gt.z + gt.w : fast_sigmoid(gt.w) * fast_tanh(c_new);
            if (qok[i]) { *cptr[i] = c_new; g.out[(size_t)qm[i] * g.ldo + qunit[i]] = u; }
        }
    }
    stamp(4);
}

template <int MT, int NT, int EPI, int AOP, int WT, int MODE, int ASM, int NW = 4>
__global__ __launch_bounds__(NW * 64, (MT == 4 && !ASM) ? 2 : 1) void gemm_f32_kernel(GemmArgs g)
{
    if constexpr (EPI == EPI_LSTM) stamp_begin(g.stamp, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
    gemm_body<MT, NT, EPI, AOP, WT, MODE, ASM, NW>(g, (int)blockIdx.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), (int)blockIdx.x, (int)blockIdx.y, gridDim.y == 1);
    if constexpr (EPI == EPI_LSTM) stamp_end(g.stamp, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// The same GEMM for `gridDim.z / zdiv` INDEPENDENT problems of one shape in one launch: blockIdx.z / zdiv selects the argument
// block (a device array written before the launch, read with scalar loads), blockIdx.z % zdiv is the slab group.  The
// offline wavefront schedule (engine.cc) batches the same step of all encoder layers this way: at one session a layer's
// recurrent step is a latency-bound launch, twelve of them in one launch cost about the same.
template <int MT, int NT, int EPI, int AOP, int WT, int MODE, int ASM>
__global__ __launch_bounds__(256, (MT == 4 && !ASM) ? 2 : 1) void gemm_f32_zkernel(const GemmArgs *__restrict__ zargs, int zdiv)
{
    const int zl = (int)blockIdx.z / zdiv;
    // a plain copy through the read-only, non-aliased kernel argument: uniform address -> scalar loads.  (The pointer members
    // are generic to the compiler -- flat_load / flat_store instead of global_load with an SGPR base -- and neither assumptions
    // nor address-space round trips change that; measured against the by-value kernel at one problem per launch: no difference.)
    const GemmArgs g = zargs[zl];
    if constexpr (EPI == EPI_LSTM) stamp_begin(g.stamp, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
    gemm_body<MT, NT, EPI, AOP, WT, MODE, ASM, 4>(g, (int)blockIdx.z - zl * zdiv, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), (int)blockIdx.x, (int)blockIdx.y, gridDim.y == 1);
    if constexpr (EPI == EPI_LSTM) stamp_end(g.stamp, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// Balanced form of the z-batched launch for tile counts that are not a whole number of rounds (three 256-row gate problems =
// 768 tiles on 512 resident workgroups): `slots` workgroups walk the tile list with that stride, tile id = x + gx (y + gy z),
// so a CU's two resident workgroups get three tiles between them (slot s takes s and s + slots: the same x, i.e. the same
// XCD and weight column stripe, as in the plain launch).  The plain launch leaves the placement of the last 256 workgroups to
// the dispatcher: they land wherever a slot frees first, some CUs take two of them and the launch waits for those (measured
// 58 us or 70 us per launch, at random; profiles/r04e_b256_streams_under_rocprof.txt).  Same body, same arguments per tile.
template <int MT, int NT, int EPI, int AOP, int WT, int MODE, int ASM>
__global__ __launch_bounds__(256, (MT == 4 && !ASM) ? 2 : 1) void gemm_f32_zkernel_walk(const GemmArgs *__restrict__ zargs, int zdiv, int gx, int gy, int ntiles)
{
    // (at most two tiles per workgroup, as two straight-line copies of the body: a loop around it costs 20 registers, which
    // takes the hand-scheduled 64 x 64 tile over 256 and the kernel down to one workgroup per CU)
    unsigned long long *stamp = nullptr;                         // (gates clock: the slot of the launch = of any of its problems)
    {
        const int tile = (int)blockIdx.x;
        const int bx = tile % gx, r = tile / gx, by = r % gy, bz = r / gy;
        const int zl = bz / zdiv;
        const GemmArgs g = zargs[zl];
        if constexpr (EPI == EPI_LSTM) { stamp = g.stamp; stamp_begin(stamp, blockIdx.x == 0); }
        gemm_body<MT, NT, EPI, AOP, WT, MODE, ASM, 4>(g, bz - zl * zdiv, (unsigned)tile, bx, by, gy == 1);
    }
    const int tile = (int)blockIdx.x + (int)gridDim.x;
    if (tile < ntiles) {
        __syncthreads();                                         // the first tile's epilogue is done with the LDS planes
        const int bx = tile % gx, r = tile / gx, by = r % gy, bz = r / gy;
        const int zl = bz / zdiv;
        const GemmArgs g = zargs[zl];
        gemm_body<MT, NT, EPI, AOP, WT, MODE, ASM, 4>(g, bz - zl * zdiv, (unsigned)tile, bx, by, gy == 1);
    }
    if constexpr (EPI == EPI_LSTM) stamp_end(stamp, gridDim.x, blockIdx.x);
}

// Mixed tiles for a z-batched launch whose 64 x 64 tiles are not a whole number of rounds (FFN up, three 256-row problems = 384 tiles: half
// of the CUs get two workgroups, half one, and the launch takes as long as four problems): the first n - 1 problems on 64 x 64 tiles, the
// last one on 64 x 32 -- 256 + 256 workgroups, every CU one of each.  One-dimensional grid, big tiles first (the dispatcher deals the
// first 256 workgroups one per CU).  Same bodies, same arguments per tile: the sums do not depend on the tile shape.
template <int EPI, int AOP, int MODE>
__global__ __launch_bounds__(256, 1) void gemm_f32_zkernel_mixed(const GemmArgs *__restrict__ zargs, int gx4, int gy, int nbig_problems)
{
    const int nbig = gx4 * gy * nbig_problems;
    int tile = (int)blockIdx.x;
    if (tile < nbig) {
        const int bx = tile % gx4, r = tile / gx4, by = r % gy, z = r / gy;
        const GemmArgs g = zargs[z];
        gemm_body<4, 4, EPI, AOP, 0, MODE, 1, 4>(g, 0, (unsigned)tile, bx, by, gy == 1);
    } else {
        tile -= nbig;
        const int gx2 = 2 * gx4, bx = tile % gx2, by = tile / gx2;
        const GemmArgs g = zargs[nbig_problems];
        gemm_body<4, 2, EPI, AOP, 0, MODE, 1, 4>(g, 0, (unsigned)(nbig + tile), bx, by, gy == 1);
    }
}

// ---------------------------------------------------------------- host side
struct TilePlan { int mt, nt, zs, mode; };

ProfileEvents &tl_profile_events() { static thread_local ProfileEvents pe; return pe; }
void gemm_profile_next_launch(hipEvent_t start, hipEvent_t stop) { ProfileEvents &pe = tl_profile_events(); pe.a = start; pe.b = stop; }
bool gemm_profile_pending() { return tl_profile_events().a != nullptr; }

static int env_int(const char *name, int def) { const char *v = getenv(name); return v && *v ? atoi(v) : def; }

// The full-K schedule pays once output tiles alone occupy a good part of the chip.  Tiles: at most 64x32 (three
// accumulator-sized register sets: chain, slab, tree level), at least 16x32 (a sum-of-squares granule is 32 columns).
static bool plan_fullk(int M, int N, int kz, TilePlan &t, bool force = false, int zcount = 1)
{
    static const int enabled = env_int("APRIL_FULLK", 1);
    static const int min_wgs = env_int("APRIL_FULLK_MIN_WGS", 96);
    if (!enabled || N % 32 != 0) return false;
    const int ncols = N / 32;
    int mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    // (zcount same-shape problems share a launch: the chip is filled by all of them together, so each can use larger tiles)
    while (mt > 1 && (long)ncols * ((M + 16 * mt - 1) / (16 * mt)) * zcount < 256) mt >>= 1;
    const long wgs = (long)ncols * ((M + 16 * mt - 1) / (16 * mt)) * zcount;
    if (wgs < min_wgs && !force) return false;
    t.mt = mt; t.nt = 2; t.zs = kz; t.mode = kz >= 4 ? GM_FULLK : GM_SLAB;    // kz 1 or 2: one workgroup walks the slabs (<= 2 meets)
    return true;
}

// GM_TILE (kernels_gemm_tile.hip): 64-column tiles of 32 (or 64) rows whose waves share the operands through LDS.
// Measured on MI355X (tools/tile_bench, profiles/r03_tile_bench.txt; us per launch, round-2 schedule -> GM_TILE):
//   FFN down 2048 rows x 2 problems 110 -> 83 (32-row tiles, all of K per workgroup), 1024 x 2: 60 -> 46, projection 2048 x 2:
//   64 -> 51, larger encoder 512 x 3: 105 -> 77 (K cut in four, row kernel);  256 rows x 1..3 problems: no gain (a launch of a few
//   hundred short workgroups is bound by its fixed costs -- pipeline fill, the second launch), so small launches keep the
//   round-2 schedules.  Rule: no GM_TILE below 256 32-row tiles per launch; 64-row tiles once the launch holds 512 of them,
//   32-row tiles otherwise; slabs per workgroup from a small cost model (below).
// APRIL_TILE_MT / APRIL_TILE_ZS (or gemm_tile_pin) pin the choice for measurements.
static int g_tile_pin_mt = env_int("APRIL_TILE_MT", 0), g_tile_pin_zs = env_int("APRIL_TILE_ZS", 0), g_tile_enable = -1;
void gemm_tile_pin(int enable, int mt, int zs) { g_tile_enable = enable; g_tile_pin_mt = mt; g_tile_pin_zs = zs; }

// GM_PP (kernels_gemm_pp.hip): the fp16 gates / FFN-up GEMMs on 256 (128) x 128 ping-pong tiles.  One workgroup per CU (144 KB of stage
// buffers), so a launch is ceil(tiles / 256) rounds of one tile time; 256-row tiles are the efficient ones (85 flop per operand byte,
// 16 MFMAs per 8 fragment reads), 128-row tiles fill the chip at fewer rows.  APRIL_GM_PP=0 keeps the round-3 GM_TILE forms;
// APRIL_PP_MT pins the tile rows / 16 (gemm_pp_pin for tools/pp_bench).
static int g_pp_enable = -1, g_pp_pin_mt = 0;
void gemm_pp_pin(int enable, int mt) { g_pp_enable = enable; g_pp_pin_mt = mt; }
static bool plan_pp(int M, int N, int zcount, TilePlan &t, bool wide_ok = false)
{
    static const int enabled = env_int("APRIL_GM_PP", 1), env_mt = env_int("APRIL_PP_MT", 0), min_tiles = env_int("APRIL_PP_MIN_TILES", 128);
    static const int wide = env_int("APRIL_PP_WIDE", 1);      // 256 x 192 tiles (kernels_gemm_pw.hip) where they save a round of workgroups
    if (!(g_pp_enable < 0 ? enabled : g_pp_enable) || N % 128 != 0) return false;
    const long zc = std::max(1, zcount);
    const long t16 = (long)(N / 128) * ((M + 255) / 256) * zc, t8 = (long)(N / 128) * ((M + 127) / 128) * zc;
    int mt = g_pp_pin_mt ? g_pp_pin_mt : env_mt;
    int nt = 8;
    if (mt == 12) { if (!wide_ok) return false; mt = 16; nt = 12; }      // (pinned: the wide form or nothing)
    else if (mt != 16 && mt != 8) {
        if (t8 < min_tiles) return false;                 // a handful of tiles: the small-batch forms stream the weights with more workgroups
        // rounds of one workgroup per CU; a 128-row tile costs ~0.6 of a 256-row one (half the MFMAs, the same weight pieces), a
        // 256 x 192 tile 1.5 of it: the larger encoder's gates at three 512-row problems are 288 tiles of 256 x 128 (two rounds) but 192
        // of 256 x 192 (one)
        const long c16 = ((t16 + 255) / 256) * 100, c8 = ((t8 + 255) / 256) * 60;
        mt = c16 <= c8 ? 16 : 8;
        if (wide && wide_ok && N % 192 == 0) {
            const long tw = (long)(N / 192) * ((M + 255) / 256) * zc, cw = ((tw + 255) / 256) * 150;
            if (cw < std::min(c16, c8)) { mt = 16; nt = 12; }
        }
    }
    t.mt = mt; t.nt = nt; t.zs = 1; t.mode = GM_PP;
    return true;
}

static bool plan_tile(int M, int N, int kz, int zcount, bool force_full, TilePlan &t, bool always = false, bool big_ok = false, bool wide_ok = false, int pp_ok = 0)
{
    static const int enabled = env_int("APRIL_GM_TILE", 1);
    static const int min_rows = env_int("APRIL_TILE_MIN_ROWS", 32);
    static const int fused_tiles = env_int("APRIL_TILE_FUSED_TILES", 512), split_tiles = env_int("APRIL_TILE_SPLIT_TILES", 256);
    const int pin_mt = g_tile_pin_mt, pin_zs = g_tile_pin_zs;
    if (N % 64 != 0) return false;
    if (!always && (!(g_tile_enable < 0 ? enabled : g_tile_enable) || M < min_rows)) return false;
    const long zc = std::max(1, zcount);
    const long tiles4 = (long)(N / 64) * ((M + 63) / 64) * zc, tiles2 = (long)(N / 64) * ((M + 31) / 32) * zc;
    // (fp16 tile engines: 64-row tiles for the N = d_model GEMMs measured no better than 32-row ones at 512 sessions -- 13.2 vs 12.7 ms
    // of projection + FFN time per 10 feeds -- so the same rule serves both precisions; APRIL_TILE_F16_MT pins it for measurements)
    static const int f16_mt = env_int("APRIL_TILE_F16_MT", 0);
    const int mt = pin_mt ? (pin_mt == 4 ? 4 : 2) : ((always && f16_mt) ? f16_mt : (tiles4 >= fused_tiles ? 4 : 2));
    const long tiles = mt == 4 ? tiles4 : tiles2;
    if ((pp_ok & 1) && kz == 1 && pin_mt == 0 && plan_pp(M, N, zcount, t, (pp_ok & 2) != 0)) return true;
    if (big_ok && kz == 1 && N % 128 == 0 && pin_mt == 0) {
        // fp16 gates / FFN up: 128 x 128 tiles (eight waves) once they give most CUs a workgroup -- twice the flops per operand byte
        static const int big = env_int("APRIL_TILE_BIG", 1), big_tiles = env_int("APRIL_TILE_BIG_TILES", 192), big_min_n = env_int("APRIL_TILE_BIG_MIN_N", 0);
        const long tiles8 = (long)(N / 128) * ((M + 127) / 128) * zc;
        if (big && tiles8 >= big_tiles && N >= big_min_n) {
            t.mt = 8; t.nt = 8; t.zs = 1; t.mode = GM_TILE;
            // APRIL_TILE_BIG_NT=12 (measurement): 128 x 192 tiles where N divides -- fewer, fatter workgroups: two 512-row problems of
            // the larger encoder's gates are 256 tiles = one per CU (128 x 128: 384 tiles = one and a half rounds of one workgroup per CU)
            static const int big_nt = env_int("APRIL_TILE_BIG_NT", 8);
            if (big_nt == 12 && N % 192 == 0) t.nt = 12;
            return true;
        }
    }
    int zs = kz;
    // fp16 N = d_model GEMMs (projection, FFN down) keep all of K in the workgroup, i.e. their fused row epilogue: the cost model below
    // prices a K cut at 5 % + 300, but the row kernel behind the planes and its launch boundary cost ~8 us at these sizes -- configs[4]
    // at 512 sessions 1.887 -> 1.820 ms per step with the cut forbidden (round 6; APRIL_TILE_F16_FULLK=0 restores the model's choice)
    // -- from 192 tiles per launch (one 512-row problem); below that the cut is what spreads the weights over the chip
    static const int f16_fullk = env_int("APRIL_TILE_F16_FULLK", 1), f16_fullk_tiles = env_int("APRIL_TILE_F16_FULLK_TILES", 192);
    if (always && f16_fullk && wide_ok && tiles >= f16_fullk_tiles) { }
    else if (pin_zs > 0) { if (!force_full) zs = std::min(kz, pin_zs); }
    else if (pin_mt > 0) { /* measurement: pinned tile rows, all of K */ }
    else {
        if (tiles2 < split_tiles && !always) return false;          // small launches keep the round-2 schedules
        if (!force_full) {
            // workgroups are dealt to the 256 CUs round robin, a CU works through its share at the MFMA rate: cost = (workgroups
            // per CU) x (stages per workgroup + ~3 stages of fill / epilogue), a K cut pays the row kernel on top (~5 %).  A slab is
            // taken as 8 stages (chunks of 4 k blocks); mt = 4 stages hold twice the MFMAs of mt = 2 stages (same for every zs).
            long best = -1;
            for (int z = kz; z >= 1; z >>= 1) {
                const long wgs = tiles * (kz / z), per_cu = (wgs + 255) / 256;
                long cost = per_cu * (8L * z + 3) * 100;
                if (z < kz) cost += cost / 20 + 300;
                if (best < 0 || cost < best) { best = cost; zs = z; }
            }
        }
    }
    t.mt = mt; t.nt = 4; t.zs = zs; t.mode = GM_TILE;
    if (always && wide_ok && zs == kz && N % 96 == 0 && pin_mt == 0) {
        // MEASUREMENT FORM, off (APRIL_TILE_NT6=1).  fp16 projection / FFN down, all of K in the workgroup: if these launches took as long as the
        // operand bytes of their busiest CU (tools/pp_bench probe: 32 x 64 tiles three per CU, 128 x 128 and 256 x 128 ping-pong tiles one per
        // CU all read ~63 GB/s per CU), 96-column tiles (N = 768 = 8 x 96: 192 / 256 tiles for three / two 512-row problems, one per CU,
        // where 64 columns give 576 / 384) would take a third off them.  Built (64 x 96 / 32 x 96, four waves), bit-identical
        // (tests/test_gpu_f16.py), measured: configs[4] 1.75 -> 1.81 ms per step (the N = d_model class 4.55 -> 4.80 ms per ten steps).
        static const int nt6 = env_int("APRIL_TILE_NT6", 0);
        if (nt6) {
            long best = -1;
            for (int cm : {2, 4}) for (int cn : {4, 6}) {
                const long tl = (long)(N / (16 * cn)) * ((M + 16 * cm - 1) / (16 * cm)) * zc;
                const long cost = ((tl + 255) / 256) * (16L * cm + 16L * cn) * 1000 + (cn == 4 && cm == mt ? 0 : 1);      // (ties keep the 64-column rule)
                if (best < 0 || cost < best) { best = cost; t.mt = cm; t.nt = cn; }
            }
        }
    }
    if (wide_ok && N % 128 == 0 && M >= 48 && pin_mt == 0 && t.nt == 4) {
        // fp16 projection / FFN down: 64 x 128 tiles, eight waves -- half the operand bytes per flop of 32 x 64; the cost model again,
        // on those tiles
        static const int wide = env_int("APRIL_TILE_WIDE", 0);      // measured: no gain (projection + FFN down 10.94 vs 11.02 ms per 10 feeds, more row-kernel work) -> off
        if (wide) {
            const long tw = (long)(N / 128) * ((M + 63) / 64) * zc;
            int zw = kz;
            if (!force_full && pin_zs == 0) {
                long best = -1;
                for (int z = kz; z >= 1; z >>= 1) {
                    const long wgs = tw * (kz / z), per_cu = (wgs + 255) / 256;
                    long cost = per_cu * (8L * z + 3) * 100;
                    if (z < kz) cost += cost / 20 + 300;
                    if (best < 0 || cost < best) { best = cost; zw = z; }
                }
            } else if (pin_zs > 0 && !force_full) zw = std::min(kz, pin_zs);
            t.mt = 4; t.nt = 8; t.zs = zw;
        }
    }
    return true;
}

bool gemm_tile_planned(int M, int N, int kz, int zcount) { TilePlan t; return plan_tile(M, N, kz, zcount, false, t); }

bool gemm_fullk(int M, int N, int kz, bool force, int zcount, int tile_ok)
{
    TilePlan t;
    // (tile_ok == 2 on a row-epilogue GEMM = the fp16 tile path: its wide tiles take part in the plan, here as in launch_gemm)
    if (tile_ok && plan_tile(M, N, kz, zcount, force, t, tile_ok == 2, false, tile_ok == 2)) return t.zs == kz;
    return plan_fullk(M, N, kz, t, force);
}

// Tile shape and slabs per workgroup.  Depends on M only through occupancy; numerics are tile-independent.
static TilePlan plan_tiles(int M, int N, int kz, int epi, bool force_fullk = false, int zcount = 1, int tile_ok = 0, int zcount_true = 1, bool f16 = false, int pp_ok = 0)
{
    // measurement knobs (default 0): 1/2 = smaller tiles for the fused-epilogue GEMMs (measured slower on MI355X:
    // B=256 gates 27 -> 32..36 us, the kernel is limited by operand loads per MFMA, not by occupancy);
    // 5 = 64x32 tiles for split-K GEMMs at M > 32
    static const int tune = env_int("APRIL_GEMM_TUNE", 0);
    TilePlan t;
    if (tile_ok && (epi == EPI_PARTIAL || epi == EPI_HR || epi == EPI_RESID_SSQ || epi == EPI_SLOT_STORE || epi == EPI_LSTM || epi == EPI_BIAS_DSWISH || (epi == EPI_XPART && tile_ok == 2))) {
        // the caller asked gemm_fullk first: a row epilogue arrives only when that plan keeps all of K in the workgroup
        if (plan_tile(M, N, kz, zcount_true, force_fullk || epi != EPI_PARTIAL, t, tile_ok == 2, (f16 || env_int("APRIL_TILE_BIG_F32", 0)) && tile_ok == 2 && (epi == EPI_LSTM || epi == EPI_BIAS_DSWISH || epi == EPI_XPART),
                      tile_ok == 2 && (epi == EPI_PARTIAL || epi == EPI_HR || epi == EPI_RESID_SSQ), (f16 && tile_ok == 2) ? pp_ok : 0)) return t;
        if (tile_ok == 2) { fprintf(stderr, "libapril(mi355x): launch_gemm: no GM_TILE plan for an always-tile GEMM (M=%d N=%d kz=%d)\n", M, N, kz); abort(); }
    }
    if (epi != EPI_LSTM && epi != EPI_BIAS_DSWISH && epi != EPI_XPART && plan_fullk(M, N, kz, t, force_fullk, zcount)) return t;
    const int ntiles = N / 16;
    t.mode = GM_SLAB;
    t.mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    int mblocks = (M + t.mt * 16 - 1) / (t.mt * 16);
    t.nt = 4;
    // (problems sharing a z-batched launch fill the chip together: wider tiles, i.e. the hand-scheduled 64x32 / 64x64 forms, at
    // fewer rows.  Only for 64-row tiles: the 16-row weight streams of the offline recurrence lose 12 % with 16x64 tiles.)
    const int zc = t.mt == 4 ? zcount : 1;
    while (t.nt > 1 && ((ntiles % t.nt) != 0 || (long)(ntiles / t.nt) * mblocks * kz * zc < 256)) t.nt >>= 1;
    // FFN up (one slab, no walking form): a launch whose 64 x 64 tiles are one and a half per CU (three 256-row problems = 384) leaves half of
    // the CUs with two workgroups and half with one; as 64 x 32 tiles it is three per CU.  APRIL_FF1_BALANCE (round 6): MEASUREMENT FORMS, off --
    // 1 (three-problem launches as 64 x 32 tiles): 1.334 vs 1.333 ms per 256-session step; 2 (two-problem launches too: two workgroups per CU): 1.351 vs 1.334
    static const int ff1_balance = env_int("APRIL_FF1_BALANCE", 0);
    if (ff1_balance && epi == EPI_BIAS_DSWISH && t.mt == 4 && t.nt == 4 && kz == 1) {
        const long tiles = (long)(ntiles / 4) * mblocks * zc;
        if (tiles > 256 && tiles < 512 && tiles % 256 != 0 && (2 * tiles) % 256 == 0) t.nt = 2;
        if (ff1_balance == 2 && tiles == 256) t.nt = 2;      // (two workgroups per CU instead of one)
    }
    if (tune && epi != EPI_PARTIAL && t.mt == 4 && (long)(ntiles / t.nt) * mblocks < 512) {
        if (tune == 1) { t.mt = 2; mblocks = (M + 31) / 32; }
        else if (tune == 2 && t.nt == 4) t.nt = 2;
    }
    if (tune == 5 && epi == EPI_PARTIAL && t.mt == 4 && t.nt == 4) t.nt = 2;
    if (tune == 3 && epi != EPI_PARTIAL && t.mt == 4 && t.nt == 4) t.nt = 2;      // measurement: 64x32 fused tiles at every size (three workgroups per CU with the compiler loop)
    t.zs = 1;
    if (epi == EPI_PARTIAL) {
        // slabs per workgroup grow once the output tiles alone fill the chip; the 64x64 tile has registers for two
        // tree levels only (NLVL in the kernel)
        const long tiles = (long)(ntiles / t.nt) * mblocks;
        const int zs_max = (t.mt == 4 && t.nt == 4) ? 4 : 8;
        while (t.zs < kz && t.zs < zs_max && tiles * (kz / (t.zs * 2)) >= 256) t.zs *= 2;
    }
    return t;
}

int gemm_partials(int M, int N, int kz, int zcount, int tile_ok)
{
    const TilePlan t = plan_tiles(M, N, kz, EPI_PARTIAL, false, 1, tile_ok, zcount);
    return t.mode == GM_FULLK ? 1 : kz / t.zs;
}

template <int MT, int NT, int EPI, int AOP, int MODE>
static void launch_one(const GemmArgs &g, hipStream_t s)
{
    constexpr bool HAS_ASM = MODE == GM_SLAB && MT == 4 && (NT == 4 || NT == 2) && AOP == AOP_NONE && (EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH);
    if constexpr (MODE == GM_FULLK) {
        static const int nw8 = env_int("APRIL_FULLK_NW8", 0);      // measured: no gain (the tiles are bound by L2 -> L1 bytes, ~24 B/clk/CU, not by MFMA issue)
        if (g.kz == 8 && nw8) {       // one slab per wave
            using Cfg8 = TileCfg<MT, NT, 8>;
            dim3 grid8((unsigned)(g.N / Cfg8::BN), (unsigned)((g.M + Cfg8::BM - 1) / Cfg8::BM), 1);
            const int sg8 = EPI == EPI_HR ? g.r_scale.groups : (EPI == EPI_SLOT_STORE && g.x_scale.ssq ? g.x_scale.groups : 0);
            const size_t lds8 = (size_t)(Cfg8::LDS_FLOATS + Cfg8::BM + (sg8 ? Cfg8::BM * (sg8 + 1) : 0)) * sizeof(float);
            if (g.wt == 1) APRIL_LAUNCH((gemm_f32_kernel<MT, NT, EPI, AOP, 1, MODE, 0, 8>), grid8, dim3(512), lds8, s, g);
            else APRIL_LAUNCH((gemm_f32_kernel<MT, NT, EPI, AOP, 0, MODE, 0, 8>), grid8, dim3(512), lds8, s, g);
            return;
        }
    }
    using Cfg = TileCfg<MT, NT>;
    dim3 grid((unsigned)(g.N / Cfg::BN), (unsigned)((g.M + Cfg::BM - 1) / Cfg::BM), (unsigned)(MODE == GM_FULLK ? 1 : g.kz / g.zs));
    static const int ldspad = env_int("APRIL_GEMM_LDSPAD", 0);   // measurement: KiB of LDS to request at least (> 80 forces one workgroup per CU)
    const int sg = EPI == EPI_HR ? g.r_scale.groups : ((EPI == EPI_LSTM || EPI == EPI_SLOT_STORE || EPI == EPI_XPART) && g.x_scale.ssq ? g.x_scale.groups : 0);
    const size_t lds = std::max((size_t)(Cfg::LDS_FLOATS + Cfg::BM + (sg ? Cfg::BM * (sg + 1) : 0)) * sizeof(float), (size_t)ldspad * 1024);
    if (g.wt == 1) { APRIL_LAUNCH((gemm_f32_kernel<MT, NT, EPI, AOP, 1, MODE, 0>), grid, dim3(256), lds, s, g); return; }
    if constexpr (HAS_ASM) {
        if (g.asm_loop && g.debug != 1) { APRIL_LAUNCH((gemm_f32_kernel<MT, NT, EPI, AOP, 0, MODE, 1>), grid, dim3(256), lds, s, g); return; }
    }
    APRIL_LAUNCH((gemm_f32_kernel<MT, NT, EPI, AOP, 0, MODE, 0>), grid, dim3(256), lds, s, g);
}

// (epilogue, prologue, schedule) combinations that exist; everything else is a programming error
template <int MT, int NT>
static bool dispatch(const GemmArgs &g, hipStream_t s)
{
#define CASE(E, A, MD) if (g.epi == E && g.a_op == A && g.mode == MD) { launch_one<MT, NT, E, A, MD>(g, s); return true; }
    if constexpr (NT == 2) {        // the full-K schedule uses 16/32/64 x 32 tiles only
        CASE(EPI_PARTIAL, AOP_TANH_ADD, GM_FULLK)
        CASE(EPI_HR, AOP_NONE, GM_FULLK) CASE(EPI_RESID_SSQ, AOP_NONE, GM_FULLK) CASE(EPI_SLOT_STORE, AOP_NONE, GM_FULLK)
        CASE(EPI_HR, AOP_NONE, GM_SLAB) CASE(EPI_RESID_SSQ, AOP_NONE, GM_SLAB) CASE(EPI_SLOT_STORE, AOP_NONE, GM_SLAB)
    }
    CASE(EPI_PARTIAL, AOP_NONE, GM_SLAB) CASE(EPI_PARTIAL, AOP_TANH_ADD, GM_SLAB)
    CASE(EPI_LSTM, AOP_NONE, GM_SLAB) CASE(EPI_XPART, AOP_NONE, GM_SLAB)
    CASE(EPI_BIAS_DSWISH, AOP_NONE, GM_SLAB)
#undef CASE
    return false;
}

// GM_KW (kernels_gemm_kw.hip) takes over a fused full-K plan of the K-split kernels -- the row-epilogue GEMMs on 16..64 x 32
// GM_FULLK / GM_SLAB tiles, and (APRIL_KW_FF1) the FFN-up GEMM on its hand-scheduled slab tiles -- when the operands allow it
// (gemm_kw_waves) and the launch is a few hundred rows: below APRIL_KW_MIN_ROWS the launch is latency-bound either way, above
// the GM_TILE threshold plan_tile has already taken it.  The decision never changes a caller-visible property of the plan (all
// of K in the workgroup, row work in the epilogue), so gemm_fullk / gemm_partials need not know.
static int g_kw_enable = -1, g_kw_pin_mt = 0, g_kw_ff1 = -1;
void gemm_kw_pin(int enable, int mt, int ff1) { g_kw_enable = enable; g_kw_pin_mt = mt; g_kw_ff1 = ff1; }
static void plan_kw(const GemmArgs &g, TilePlan &t, int zc)
{
    static const int enabled = env_int("APRIL_GM_KW", 1), min_rows = env_int("APRIL_KW_MIN_ROWS", 3), max_rows = env_int("APRIL_KW_MAX_ROWS", 1 << 30);
    static const int ff1 = env_int("APRIL_KW_FF1", 0), env_mt = env_int("APRIL_KW_MT", 0);
    if (!(g_kw_enable < 0 ? enabled : g_kw_enable) || t.zs != g.kz) return;
    if (t.mode == GM_TILE) {
        // a FUSED GM_TILE plan (all of K in the workgroup: the same caller-visible shape) keeps the launch unless its tiles fill the 512
        // resident slots badly: half a round or less, or a small remainder behind whole rounds (tools/kw_bench, FFN down: 512 x 2 rows
        // 31.3 (GM_TILE) vs 29.6 us (GM_KW); 768 x 3: 64.8 vs 59.0; 2300 x 2: 117 vs 112; but 1024 x 2: 47 vs 54, 2048 x 3: 123 vs 147)
        static const int over_tile = env_int("APRIL_KW_OVER_TILE", 1);
        if (!over_tile || (g.epi != EPI_HR && g.epi != EPI_RESID_SSQ) || t.nt != 4) return;
        const long tiles = (long)(g.N / 64) * ((g.M + 16 * t.mt - 1) / (16 * t.mt)) * zc;
        const long rem = tiles % 512;
        if (!(tiles <= 256 || (tiles > 512 && rem > 0 && rem <= 128))) return;
    }
    if (g.M < min_rows || g.M > max_rows) return;
    if (g.epi == EPI_LSTM) {
        // MEASUREMENT FORM, off by default (APRIL_KW_GATES=1): the gates GEMM with its activation rows through the wave-private LDS rings, four
        // waves = the four chunks, 32 / 64 x 32 tiles.  The idea: at one 64-row problem per launch the K-split 64 x 16 tiles spend 19.8 k cycles
        // in a K loop of 8.2 k cycles of MFMA on their 16-row x 64-byte fragment loads (profiles/r05_gates_ffup_phase_trace.txt).  Bit-identical
        // (tools/kw_bench), but the z-batched launches of the engine already run 64 x 32 / 64 x 64 hand-scheduled tiles that are within 25 % of
        // their MFMA time: 64 rows x 1 / 2 / 3 problems 12.8 / 14.9 / 24.0 us (K-split) vs 9.5 / 15.0 / 23.8 (best GM_KW tile), 128 x 2:
        // 22.8 vs 30.3, 256 x 2: 40.1 vs 55.3 -- a gain only for one-problem launches, a loss from 96 rows up.
        static const int gates = env_int("APRIL_KW_GATES", 0), gates_max = env_int("APRIL_KW_GATES_MAX_ROWS", 96);
        if (!gates || g.M < 17 || g.M > gates_max || gemm_kw_waves(g) != 4) return;
        const int gmt = g_kw_pin_mt ? g_kw_pin_mt : (g.M > 32 ? 4 : 2);
        if (!gemm_kw_has_kernel(g, gmt, 2)) return;         // (a pinned tile shape without a kernel: the previous plan stays)
        t.mt = gmt; t.nt = 2; t.zs = 1; t.mode = GM_KW;
        return;
    }
    if (g.epi == EPI_BIAS_DSWISH ? !(g_kw_ff1 < 0 ? ff1 : g_kw_ff1) : (g.epi != EPI_HR && g.epi != EPI_RESID_SSQ)) return;
    const int nw = gemm_kw_waves(g);
    if (!nw) return;
    int mt = g_kw_pin_mt ? g_kw_pin_mt : env_mt;
    const int nt = (g.epi == EPI_BIAS_DSWISH && g.N % 64 == 0) ? 4 : 2;
    if (!mt) {
        // 32-row tiles are the efficient ones (one memory instruction per four MFMAs; 16-row tiles: three per eight), 16-row tiles the
        // finer grain: the launch takes ceil(tiles / 256 CUs) rounds, a 16-row round costs ~0.55 of a 32-row one (tools/kw_bench)
        const long t32 = (long)(g.N / (16 * nt)) * ((g.M + 31) / 32) * zc, t16 = (long)(g.N / (16 * nt)) * ((g.M + 15) / 16) * zc;
        mt = (nw == 8 && ((t16 + 255) / 256) * 55 < ((t32 + 255) / 256) * 100) ? 1 : 2;
    }
    if (nw == 4 && mt == 1) mt = 2;
    const int knt = (mt == 4 && nw == 8 && g.N % 64 == 0) ? 4 : nt;      // (pinned 64-row tiles: 64 x 64, a wave tile of one memory instruction per eight MFMAs)
    // environment knobs and pins can ask for shapes that were never instantiated (APRIL_KW_FF1 with N % 64 != 0, 64-row tiles on four
    // waves or at N % 64 != 0 ...): the previous plan stays instead of an abort in launch_gemm_kw
    if (!gemm_kw_has_kernel(g, mt, knt)) return;
    t.mt = mt; t.nt = knt; t.zs = g.kz; t.mode = GM_KW;
}

// plan + checks + measurement knobs: everything launch_gemm decides on the host
static TilePlan finalize_gemm(GemmArgs &g)
{
    static const int dbg = env_int("APRIL_GEMM_DEBUG", 0);
    g.debug = dbg;
    static const int skew = env_int("APRIL_GEMM_SKEW", 0);        // first-round start skew, x 4096 cycles (below; measured neutral, off)
    static const int asm_loop = env_int("APRIL_GEMM_ASM", 1);     // 0 = compiler-scheduled loop everywhere (A/B)
    static const int z_tiles = env_int("APRIL_Z_TILES", 2);      // A/B: 0 = plan z-batched problems as if each had the chip to itself, 1 = hint everywhere, 2 = fused-epilogue slab tiles only, 3 = full-K tiles only
    const int zc = std::max(1, g.zcount);
    const bool is_slab_epi = g.epi == EPI_LSTM || g.epi == EPI_BIAS_DSWISH || g.epi == EPI_XPART;
    // GM_TILE eligibility: planner-rule GEMMs (tile_ok 1) are the fp32 row-epilogue / partial GEMMs over one A segment; always-tile
    // GEMMs (tile_ok 2, fp16 tile engines) may also carry two segments and the LSTM / DoubleSwish epilogues
    // (the fp16 tile kernels also have the layer-major halves of the gate GEMM: wave_mask 0x3 + EPI_XPART, 0xC + p_add + EPI_LSTM)
    const bool lm_half = g.tile_ok == 2 && g.wt == 1 && ((g.epi == EPI_XPART && g.wave_mask == 0x3 && !g.p_add) || (g.epi == EPI_LSTM && g.wave_mask == 0xC && g.p_add));
    const bool plain = g.a_op == AOP_NONE && g.N % 64 == 0 && ((g.wave_mask == 0xF && !g.p_add) || lm_half);
    const int tile_ok = !plain ? 0 : (g.tile_ok == 2 ? 2 : ((g.tile_ok == 1 && g.K1 == 0 && g.wt == 0 && g.epi != EPI_LSTM && g.epi != EPI_BIAS_DSWISH) ? 1 : 0));
    if (g.tile_ok == 2 && !tile_ok) { fprintf(stderr, "libapril(mi355x): launch_gemm: always-tile GEMM with a prologue / wave mask / odd N\n"); abort(); }
    TilePlan t = plan_tiles(g.M, g.N, g.kz, g.epi, g.force_fullk != 0, (z_tiles == 1 || (z_tiles == 2 && is_slab_epi) || (z_tiles == 3 && !is_slab_epi)) ? zc : 1, tile_ok, zc, g.wt == 1,
                            (g.wt == 1 && is_slab_epi && gemm_pp_ok(g, 16)) ? (1 | (gemm_pw_ok(g) ? 2 : 0)) : 0);
    plan_kw(g, t, zc);
    const bool row_epi = g.epi == EPI_HR || g.epi == EPI_RESID_SSQ || g.epi == EPI_SLOT_STORE;
    if (row_epi && t.zs != g.kz) { fprintf(stderr, "libapril(mi355x): launch_gemm: row epilogue %d needs the full-K plan (M=%d N=%d kz=%d)\n", g.epi, g.M, g.N, g.kz); abort(); }
    g.zs = t.zs; g.mode = t.mode;
    // EPI_LSTM with x_scale scales the partial sums of waves 0 and 1: they must hold exactly A segment 0
    if ((g.epi == EPI_LSTM || g.epi == EPI_XPART) && (g.x_scale.ssq || g.p_add || g.wave_mask != 0xF) && (g.kz != 1 || g.K0 * 2 != g.K)) { fprintf(stderr, "libapril(mi355x): launch_gemm: x_scale needs K0 == K / 2 and kz == 1\n"); abort(); }
    // hand-scheduled K loop wherever it exists (fused-epilogue 64x64 / 64x32 fp32 tiles).  Round-2 measurements
    // (tools/gemm_bench, gates [M,1024]x[1024,4096] + LSTM cell, us per launch at M = 512 / 1024 / 2048):
    //   hand loop, no skew 42.2 / 85.6 / 159.2   hand loop, skew 2: 85.5 / 174.3   compiler loop, skew 2 (round-1 choice at two
    //   workgroups per CU): 91.3 / 170.4   compiler loop, no skew: 56.5 / 100.7 / 180.2;  FFN-up [2048,512]x[512,2048]: 42.3 vs 45.7
    g.asm_loop = asm_loop != 0;
    // first-round start skew (device_utils.h first_round_skew; measured neutral, off): launches of several rounds of the fused-epilogue
    // K-split tiles and of the four-wave GM_TILE tiles.  APRIL_GEMM_SKEW = delay in units of 4096 cycles, APRIL_SKEW_MIN_WGS = smallest launch.
    {
        static const int skew_min_wgs = env_int("APRIL_SKEW_MIN_WGS", 1536), skew_slots = env_int("APRIL_SKEW_SLOTS", 512);
        const long wgs = (long)(g.N / (16 * t.nt)) * ((g.M + 16 * t.mt - 1) / (16 * t.mt)) * (t.mode == GM_FULLK ? 1 : g.kz / g.zs) * zc;
        const bool several_per_cu = (t.mode == GM_TILE && t.nt == 4 && (t.mt == 4 || t.mt == 2)) || ((t.mode == GM_SLAB || t.mode == GM_FULLK) && is_slab_epi);
        g.skew = (several_per_cu && wgs >= skew_min_wgs) ? skew : 0;
        g.skew_wgs = skew_slots;
    }
    static const int kw_skew = env_int("APRIL_KW_SKEW", 0);      // GM_KW: start delay of the second half of a workgroup's waves, x 64 cycles (measured: no effect; kernels_gemm_kw.hip)
    if (t.mode == GM_KW) g.skew = kw_skew;
    if (t.mode == GM_PP) g.skew = 0;
    static const int kw_xcd = env_int("APRIL_KW_XCD", 0);        // GM_KW: 2 = the 2 x 4 XCD order of the tiles (kernels_gemm_kw.hip kw_tile_of)
    g.xcd_rc = t.mode == GM_KW ? kw_xcd : 0;
    return t;
}

// GM_KW instead of the weight-stream row kernel of kernels_recur.hip for the FFN-down GEMM from APRIL_KW_MIN_ROWS (3) rows:
// 16 x 32 tiles on eight waves against two 16-column tiles on sixteen waves -- 4 / 8 / 16 sessions 0.499 / 0.537 / 0.602 -> 0.475 / 0.496 / 0.527 ms
// per 100 ms feed, one or two rows the same within the noise (they stay on the stream kernels, as does the recurrent pair of a long feed)
static bool kw_before_recur(const GemmArgs &g)
{
    static const int enabled = env_int("APRIL_GM_KW", 1), min_rows = env_int("APRIL_KW_MIN_ROWS", 3);
    // (FFN down only: 9.5 .. 12.8 -> 8.2 us per launch at 8 .. 16 rows; the projection's stream kernel is the faster one there, 4.7 .. 5.4 vs 5.8 us)
    return (g_kw_enable < 0 ? enabled : g_kw_enable) && min_rows <= 16 && g.M >= min_rows && g.M <= 16 && g.epi == EPI_RESID_SSQ && gemm_kw_waves(g) != 0;
}

void launch_gemm(const GemmArgs &g_in, hipStream_t s)
{
    GemmArgs g = g_in;
    g.ksplit = kw_before_recur(g) ? 1 : recur_ksplit(g, 1);
    if (!kw_before_recur(g)) if (const int rf = recur_form(g)) { launch_recur(g, rf, nullptr, 1, s); return; }
    const TilePlan t = finalize_gemm(g);
    if (t.mode == GM_PP) { if (t.nt == 12) launch_gemm_pw(g, nullptr, 0, s); else launch_gemm_pp(g, t.mt, nullptr, 0, s); return; }
    if (t.mode == GM_TILE) { launch_gemm_tile(g, t.mt, t.nt, nullptr, 0, s); return; }
    if (t.mode == GM_KW) { launch_gemm_kw(g, t.mt, t.nt, nullptr, 0, s); return; }
    const int mt = t.mt, nt = t.nt;
    bool ok = false;
    if (mt == 1) { if (nt == 4) ok = dispatch<1, 4>(g, s); else if (nt == 2) ok = dispatch<1, 2>(g, s); else ok = dispatch<1, 1>(g, s); }
    else if (mt == 2) { if (nt == 4) ok = dispatch<2, 4>(g, s); else if (nt == 2) ok = dispatch<2, 2>(g, s); else ok = dispatch<2, 1>(g, s); }
    else { if (nt == 4) ok = dispatch<4, 4>(g, s); else if (nt == 2) ok = dispatch<4, 2>(g, s); else ok = dispatch<4, 1>(g, s); }
    if (!ok) { fprintf(stderr, "libapril(mi355x): launch_gemm: no kernel for epi %d a_op %d mode %d tile %dx%d\n", g.epi, g.a_op, g.mode, mt, nt); abort(); }
}

// ---- n independent problems of one shape in one launch (gemm_f32_zkernel)
template <int MT, int NT, int EPI, int AOP, int MODE>
static void launch_one_z(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    using Cfg = TileCfg<MT, NT>;
    const int zdiv = MODE == GM_FULLK ? 1 : g.kz / g.zs;
    dim3 grid((unsigned)(g.N / Cfg::BN), (unsigned)((g.M + Cfg::BM - 1) / Cfg::BM), (unsigned)(zdiv * n));
    const int sg = EPI == EPI_HR ? g.r_scale.groups : ((EPI == EPI_LSTM || EPI == EPI_SLOT_STORE || EPI == EPI_XPART) && g.x_scale.ssq ? g.x_scale.groups : 0);
    const size_t lds = (size_t)(Cfg::LDS_FLOATS + Cfg::BM + (sg ? Cfg::BM * (sg + 1) : 0)) * sizeof(float);
    constexpr bool HAS_ASM = MODE == GM_SLAB && MT == 4 && (NT == 4 || NT == 2) && AOP == AOP_NONE && (EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH);
    if (g.wt == 1) { APRIL_LAUNCH((gemm_f32_zkernel<MT, NT, EPI, AOP, 1, MODE, 0>), grid, dim3(256), lds, s, dev_args, zdiv); return; }
    if constexpr (HAS_ASM) {
        if (g.asm_loop && g.debug != 1) {
            if constexpr (EPI == EPI_LSTM && MT == 4 && NT == 4) {
                // APRIL_GEMM_WALK (1): 512 walking workgroups when the launch is more than one round of the chip's 512 resident
                // 64 x 64 gate workgroups but not a whole number of rounds
                static const int walk = env_int("APRIL_GEMM_WALK", 1);
                const long ntiles = (long)grid.x * grid.y * grid.z;
                if (walk && ntiles > 512 && ntiles < 1024) {
                    APRIL_LAUNCH((gemm_f32_zkernel_walk<MT, NT, EPI, AOP, 0, MODE, 1>), dim3(512), dim3(256), lds, s, dev_args, zdiv, (int)grid.x, (int)grid.y, (int)ntiles);
                    return;
                }
            }
            if constexpr (EPI == EPI_BIAS_DSWISH && MT == 4 && NT == 4 && MODE == GM_SLAB) {
                // APRIL_FF1_MIXED (round 6): see gemm_f32_zkernel_mixed
                static const int mixed = env_int("APRIL_FF1_MIXED", 1);      // 256 sessions: 1.341 -> 1.320 ms per step (three alternations on one box)
                const long t4 = (long)grid.x * grid.y, total = t4 * n;
                if (mixed && zdiv == 1 && n >= 2 && total % 256 != 0 && ((long)(n - 1) * t4) % 256 == 0 && (2 * t4) % 256 == 0 && g.N % 32 == 0) {
                    APRIL_LAUNCH((gemm_f32_zkernel_mixed<EPI, AOP, MODE>), dim3((unsigned)((n - 1) * t4 + 2 * t4)), dim3(256), lds, s, dev_args, (int)grid.x, (int)grid.y, n - 1);
                    return;
                }
            }
            APRIL_LAUNCH((gemm_f32_zkernel<MT, NT, EPI, AOP, 0, MODE, 1>), grid, dim3(256), lds, s, dev_args, zdiv); return;
        }
    }
    APRIL_LAUNCH((gemm_f32_zkernel<MT, NT, EPI, AOP, 0, MODE, 0>), grid, dim3(256), lds, s, dev_args, zdiv);
}

template <int MT, int NT>
static bool dispatch_z(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
#define CASE(E, MD) if (g.epi == E && g.a_op == AOP_NONE && g.mode == MD) { launch_one_z<MT, NT, E, AOP_NONE, MD>(g, dev_args, n, s); return true; }
    if constexpr (NT == 2) {
        CASE(EPI_HR, GM_FULLK) CASE(EPI_RESID_SSQ, GM_FULLK)
        CASE(EPI_HR, GM_SLAB) CASE(EPI_RESID_SSQ, GM_SLAB)
    }
    CASE(EPI_LSTM, GM_SLAB) CASE(EPI_XPART, GM_SLAB) CASE(EPI_BIAS_DSWISH, GM_SLAB)
#undef CASE
    return false;
}

void stage_gemm_z(const GemmArgs *items, int n, GemmArgs *staged)
{
    TilePlan t0{0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        staged[i] = items[i];
        staged[i].zcount = n;
        staged[i].ksplit = 1;
        const TilePlan t = finalize_gemm(staged[i]);
        if (i == 0) t0 = t;
        const GemmArgs &a = staged[i], &b = staged[0];
        if (t.mt != t0.mt || t.nt != t0.nt || t.zs != t0.zs || t.mode != t0.mode || a.M != b.M || a.N != b.N || a.K != b.K || a.kz != b.kz || a.epi != b.epi || a.wt != b.wt ||
            a.a_op != AOP_NONE || a.x_scale.groups != b.x_scale.groups || a.r_scale.groups != b.r_scale.groups || (a.x_scale.ssq == nullptr) != (b.x_scale.ssq == nullptr)) {
            fprintf(stderr, "libapril(mi355x): stage_gemm_z: the problems of one launch must have one shape\n"); abort();
        }
    }
    // the K cut of the <= 16-row stream kernels (kernels_recur.hip): one value for the launch, only when every problem brings its workspace
    if (n > 0 && !kw_before_recur(staged[0])) {
        int S = recur_ksplit(staged[0], n);
        for (int i = 1; i < n && S > 1; ++i) if (recur_ksplit(staged[i], n) != S) S = 1;
        for (int i = 0; i < n; ++i) staged[i].ksplit = S;
    }
}

void launch_gemm_z(const GemmArgs *staged, int n, const GemmArgs *dev_args, hipStream_t s)
{
    if (n <= 0) return;
    const GemmArgs &g = staged[0];
    if (!kw_before_recur(g)) if (const int rf = recur_form(g)) { launch_recur(g, rf, dev_args, n, s); return; }      // (stage_gemm_z checked that the n problems have one shape)
    GemmArgs probe = g;
    const TilePlan t = finalize_gemm(probe);
    if (t.mode == GM_PP) {
        // MEASUREMENT FORM, off by default (APRIL_PP_SPLIT=1).  One workgroup per CU, so a launch costs whole rounds of one tile time; when
        // the n problems give more than one round of 256-row tiles but n - 1 of them exactly one (the larger encoder's gates at 512
        // sessions: 96 tiles per problem, three problems = 288), the launch can be cut in two: n - 1 problems on 256-row tiles, the last
        // one on the planner's choice for it alone (192 128-row tiles).  tools/pp_bench: 45.0 us against 50.6 (128-row tiles for all
        // three) -- but inside the engine the three-problem launch already runs in 47.9 us and the pair costs 31.8 + 19: nothing gained
        // (configs[4] step 1.881 vs 1.888 ms), so one launch stays the rule.
        static const int split = env_int("APRIL_PP_SPLIT", 0);
        const long per16 = (long)(g.N / 128) * ((g.M + 255) / 256);
        if (split && t.nt != 12 && g_pp_pin_mt == 0 && n >= 2 && per16 * n > 256 && per16 * (n - 1) <= 256 && per16 * (n - 1) >= 160) {
            launch_gemm_pp(g, 16, dev_args, n - 1, s);
            TilePlan t1;
            if (!plan_pp(g.M, g.N, 1, t1)) { t1.mt = 8; }
            launch_gemm_pp(staged[n - 1], t1.mt, dev_args + (n - 1), 1, s);
            return;
        }
        if (t.nt == 12) launch_gemm_pw(g, dev_args, n, s);
        else launch_gemm_pp(g, t.mt, dev_args, n, s);
        return;
    }
    if (t.mode == GM_TILE) { launch_gemm_tile(g, t.mt, t.nt, dev_args, n, s); return; }
    if (t.mode == GM_KW) { launch_gemm_kw(g, t.mt, t.nt, dev_args, n, s); return; }
    const int mt = t.mt, nt = t.nt;
    bool ok = false;
    if (mt == 1) { if (nt == 4) ok = dispatch_z<1, 4>(g, dev_args, n, s); else if (nt == 2) ok = dispatch_z<1, 2>(g, dev_args, n, s); else ok = dispatch_z<1, 1>(g, dev_args, n, s); }
    else if (mt == 2) { if (nt == 4) ok = dispatch_z<2, 4>(g, dev_args, n, s); else if (nt == 2) ok = dispatch_z<2, 2>(g, dev_args, n, s); else ok = dispatch_z<2, 1>(g, dev_args, n, s); }
    else { if (nt == 4) ok = dispatch_z<4, 4>(g, dev_args, n, s); else if (nt == 2) ok = dispatch_z<4, 2>(g, dev_args, n, s); else ok = dispatch_z<4, 1>(g, dev_args, n, s); }
    if (!ok) { fprintf(stderr, "libapril(mi355x): launch_gemm_z: no kernel for epi %d mode %d tile %dx%d\n", g.epi, g.mode, mt, nt); abort(); }
}

}  // namespace aprilx
