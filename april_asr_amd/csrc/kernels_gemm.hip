// MFMA "skinny" GEMM for gfx950: out[M,N] = A[M,K] x W[K,N], M = sessions
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
// WT = 1 reads fp16 weights and rounds A to fp16 on load (v_mfma_f32_16x16x16_f16, fp32
// accumulation, same lane<->k mapping and summation structure).  The K loop exists twice with
// identical arithmetic: compiler-scheduled C++ (all shapes) and a generated hand-scheduled
// version for the fused-epilogue 64x64 / 64x32 fp32 tiles (gemm_mainloop_asm.inc).
// Replaces the ORT MatMul/Gemm nodes of the encoder/decoder/joiner graphs
// (reference call sites src/april_session.c:145,160,176).
#include "kernels.h"
#include "device_utils.h"
#include "gemm_mainloop_asm.inc"
#ifndef APRIL_ASM_NB
#define APRIL_ASM_NB 2      // operand buffers of the hand-scheduled K loop: 2 keeps two workgroups per CU, 3 needs > 256 registers
#endif
#include <algorithm>
#include <cstdlib>

namespace aprilx {


template <int MT, int NT>
struct TileCfg {
    static constexpr int BM = MT * 16, BN = NT * 16, LDR = BN + 4;
    static constexpr int LDS_FLOATS = 4 * BM * LDR;
};

using h4 = __attribute__((ext_vector_type(4))) _Float16;
template <int WT> struct WQuad { using type = f32x4; };
template <> struct WQuad<1> { using type = h4; };

template <int MT, int NT, int EPI, int AOP, int WT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g)
{
    using Cfg = TileCfg<MT, NT>;
    using BQ = typename WQuad<WT>::type;       // one lane's four consecutive k values of a weight tile
    extern __shared__ __attribute__((aligned(16))) float red[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // measurement only (tools/gemm_bench built with -DAPRIL_GEMM_TRACE, run with GEMM_TRACE=1): wave 0 stamps s_memtime at
    // phase boundaries (uniform branch, all lanes store the same value); compiled out of the product
#ifdef APRIL_GEMM_TRACE
    auto stamp = [&](int i) { if (g.trace && wave == 0) g.trace[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + i] = __builtin_amdgcn_s_memtime(); };
#else
    auto stamp = [](int) {};
#endif
    stamp(0);
#ifdef APRIL_GEMM_TRACE
    if (g.trace && wave == 0) {   // where this workgroup runs: HW_ID (wave/simd/cu/sh/se fields) and XCC_ID
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        g.trace[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    // XCD-aware mapping: consecutive blockIdx.x land on different XCDs, so keep the
    // M-blocks that share one weight column on the same XCD (same x mod 8).
    const int nt0 = blockIdx.x * NT;
    const int m0 = blockIdx.y * Cfg::BM;
    const int zg = blockIdx.z;                 // this workgroup owns slabs [zg*zs, (zg+1)*zs)
    if (g.skew > 0) {
        // Two workgroups share a CU.  Dispatched together they run in lock-step: both stream MFMAs (halving each
        // other's rate), then both sit in their epilogues while the matrix pipe idles.  Delaying every second
        // "generation" of workgroups once, by about one epilogue, interleaves the phases for the rest of the launch.
        // Placement is only a heuristic here (speed, never correctness).
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if ((lin >> 8) & 1) for (int i = 0; i < g.skew; ++i) __builtin_amdgcn_s_sleep(64);
    }
    const int KB = g.K >> 4;

    const int mrow = lane & 15, kq = lane >> 4;

    // Addressing = uniform 64-bit base (SGPRs; advances with the k block) + one 32-bit byte offset per lane and
    // m-tile (row start + this lane's k quarter), i.e. the saddr form of global_load: no 64-bit vector adds in the loop.
    // All operands are far below 4 GiB per array.
    // after the LDS meet every thread owns QPT groups of 4 consecutive columns ("quads") of the tile
    constexpr int QROW = Cfg::BN / 4, NQ = Cfg::BM * QROW, QPT = (NQ + 255) / 256;
    uint32_t aoff0[MT], aoff1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = m0 + mt * 16 + mrow;
        if (row >= g.M) row = g.M - 1;                        // padding rows recompute the last row; never stored
        const int r0 = g.aidx0 ? g.aidx0[row] : row;
        aoff0[mt] = (uint32_t)(((size_t)r0 * g.lda0 + kq * 4) * sizeof(float));
        aoff1[mt] = 0;
        if (g.K1 > 0) { const int r1 = g.aidx1 ? g.aidx1[row] : row; aoff1[mt] = (uint32_t)(((size_t)r1 * g.lda1 + kq * 4) * sizeof(float)); }
    }
    const uint32_t boff = (uint32_t)lane * sizeof(BQ);
    const bool stream_once = gridDim.y == 1;   // weights read by exactly one workgroup: bypass-friendly loads

    // pairwise (balanced-tree) slab accumulation: level b holds the sum of 2^b consecutive slabs.  A workgroup that owns
    // zs = 2^t slabs needs t levels; the 64x64 tile is capped at zs = 4 (2 levels, 32 registers) so that it stays
    // within 256 registers and two workgroups share a CU (launch_gemm / gemm_partials apply the same cap)
    constexpr int NLVL = (MT == 4 && NT == 4) ? 2 : 3;
    f32x4 lvl[NLVL][QPT];
    int top = 0;
    while ((1 << top) < g.zs) ++top;
    constexpr int PLANE = Cfg::BM * Cfg::LDR;
    auto summed4 = [&](int o) {                        // ((p0+p1)+p2)+p3 of four consecutive columns
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + o), p1 = *reinterpret_cast<const f32x4 *>(red + PLANE + o);
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(red + 2 * PLANE + o), p3 = *reinterpret_cast<const f32x4 *>(red + 3 * PLANE + o);
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
                const f32x4 w = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(g.a0b) + (ptrdiff_t)kb * 64 + off);
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
    // K blocks: KB = K/16 is a multiple of 4*kz (checked on the host), so every wave owns exactly c blocks of every
    // slab: slab z, wave w -> blocks [(4z + w) c, (4z + w + 1) c).  A workgroup walks zs consecutive slabs.
    const int c = g.debug == 1 ? 0 : KB / (4 * g.kz);
    const int T = g.zs * c;                            // blocks this wave processes in total

    // Register-staged software pipeline: DEPTH k-blocks of loads are in flight per wave while the MFMAs of the
    // oldest stage issue.  Small batches (MT == 1) are HBM-bound weight streams and want many bytes in flight per
    // CU (6 x 4 waves x (1+NT) KiB); large tiles are MFMA-bound and need only enough depth to cover L2 latency.
    // The prefetch stream runs ahead ACROSS slab boundaries, so the pipeline never restarts inside a workgroup;
    // loads in the main loop are unconditional (the fetch position parks on the last block) so the loop body is
    // straight-line code and the compiler keeps counted s_waitcnt vmcnt(N) instead of draining.
#ifndef APRIL_DEPTH4
#define APRIL_DEPTH4 2
#endif
    constexpr int DEPTH = (MT == 1) ? 6 : (MT == 2 ? 3 : APRIL_DEPTH4);
    f32x4 a_st[DEPTH][MT];
    BQ b_st[DEPTH][NT];
    int ld_base = (4 * (zg * g.zs) + wave) * c, ld_off = 0, ld_cnt = 0;
    auto ld_next = [&]() {
        const int kb = g.debug == 3 ? 0 : ld_base + ld_off;     // debug 3 (measurement): every block re-reads block 0 (cache-resident operands)
        if (ld_cnt + 1 < T) { ++ld_cnt; if (++ld_off == c) { ld_off = 0; ld_base += 4 * c; } }
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
        if constexpr (WT == 1) {
            // the same 16 k values per lane as four fp32 k-steps, in one v_mfma_f32_16x16x16_f16 (fp32 accumulate)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h4 ah = {(_Float16)a[mt].x, (_Float16)a[mt].y, (_Float16)a[mt].z, (_Float16)a[mt].w};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, b[nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b[nt].w, acc[mt][nt], 0, 0, 0);
                }
        }
    };
    int slab_done = 0;                                 // slabs finished so far by this workgroup
    // quarter chains meet in LDS (red[wave][row][col]) and are added ((p0+p1)+p2)+p3; slab sums are combined
    // pairwise in slab order (balanced tree), so owning 1, 2, 4 or 8 slabs per workgroup -- chosen from the batch
    // size -- yields the same bits
    auto meet = [&]() {
        if (slab_done > 0) __syncthreads();            // previous slab's reads of red[] are done
        float *mine = red + (size_t)wave * PLANE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mine[(mt * 16 + kq * 4 + r) * Cfg::LDR + nt * 16 + mrow] = acc[mt][nt][r];
            // one m-tile (16 accumulator registers) at a time: left alone, the scheduler copies the whole accumulator
            // file out of the AGPRs first and the kernel no longer fits two workgroups per CU
            if (MT == 4) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (EPI == EPI_PARTIAL) {
            f32x4 v[QPT];
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                const int q = threadIdx.x + i * 256;
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
            if (slab_done + 1 == g.zs) {               // the carry ran through every level: v is the total of the zs slabs
#pragma unroll
                for (int i = 0; i < QPT; ++i) {
                    const int q = threadIdx.x + i * 256;
                    const int m = m0 + q / QROW;
                    if (q < NQ && m < g.M)
                        *reinterpret_cast<f32x4 *>(g.out + ((size_t)zg * g.m_stride + m) * g.N + nt0 * 16 + (q % QROW) * 4) = v[i];
                }
            }
        }
        ++slab_done;
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
            const int q = threadIdx.x + i * 256;
            const int row = q / QROW, ul = q % QROW;
            qm[i] = m0 + row;
            qok[i] = q < NQ && qm[i] < g.M;
            const int n = nt0 * 16 + ul * 4;
            qunit[i] = n >> 2;
            qo[i] = row * Cfg::LDR + ul * 4;
            const int slot = qok[i] ? g.slot_idx[qm[i]] : 0;
            cptr[i] = g.c_state + (size_t)slot * g.hidden + qunit[i];
            qb[i] = *reinterpret_cast<const f32x4 *>(g.bias + n);
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) cprev[i] = qok[i] ? *cptr[i] : 0.0f;
    }

    zero_acc();
    stamp(1);
    bool by_hand = false;
    // fused-epilogue GEMMs only (one slab, 5..16 blocks per wave): the split-K GEMMs walk several short slabs per
    // workgroup and rely on the cross-slab prefetch of the compiler-scheduled loop below (measured: no gain there)
    if constexpr (MT == 4 && (NT == 4 || NT == 2) && WT == 0 && AOP == AOP_NONE && EPI != EPI_PARTIAL) {
        if (g.asm_loop && c > 0) {
            // hand-scheduled K loop (tools/gen_gemm_asm.py): same blocks, same order, same accumulation chains
            by_hand = true;
            uint32_t boffs[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) boffs[nt] = boff + (uint32_t)nt * (uint32_t)KB * 1024u;
            for (int z = 0; z < g.zs; ++z) {
                const int kb0 = (4 * (zg * g.zs + z) + wave) * c;
                int done = 0;
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
                    if constexpr (NT == 4) {
#if APRIL_ASM_NB == 3
                        asm volatile(APRIL_MAINLOOP3_TEXT
#else
                        asm volatile(APRIL_MAINLOOP2_TEXT
#endif
                            : [c0] "+a"(acc[0][0]), [c1] "+a"(acc[0][1]), [c2] "+a"(acc[0][2]), [c3] "+a"(acc[0][3]),
                              [c4] "+a"(acc[1][0]), [c5] "+a"(acc[1][1]), [c6] "+a"(acc[1][2]), [c7] "+a"(acc[1][3]),
                              [c8] "+a"(acc[2][0]), [c9] "+a"(acc[2][1]), [c10] "+a"(acc[2][2]), [c11] "+a"(acc[2][3]),
                              [c12] "+a"(acc[3][0]), [c13] "+a"(acc[3][1]), [c14] "+a"(acc[3][2]), [c15] "+a"(acc[3][3])
                            : APRIL_ASM_IN_A, [boff0] "v"(boffs[0]), [boff1] "v"(boffs[1]), [boff2] "v"(boffs[NT - 2]), [boff3] "v"(boffs[NT - 1])
#if APRIL_ASM_NB == 3
                            : APRIL_MAINLOOP3_CLOBBERS);
#else
                            : APRIL_MAINLOOP2_CLOBBERS);
#endif
                    } else {
                        asm volatile(APRIL_MAINLOOP2_NT2_TEXT
                            : [c0] "+a"(acc[0][0]), [c1] "+a"(acc[0][1]), [c2] "+a"(acc[1][0]), [c3] "+a"(acc[1][1]),
                              [c4] "+a"(acc[2][0]), [c5] "+a"(acc[2][1]), [c6] "+a"(acc[3][0]), [c7] "+a"(acc[3][1])
                            : APRIL_ASM_IN_A, [boff0] "v"(boffs[0]), [boff1] "v"(boffs[1])
                            : APRIL_MAINLOOP2_NT2_CLOBBERS);
                    }
#undef APRIL_ASM_IN_A
                    done += n;
                }
                stamp(2);
                meet();
                stamp(3);
                if (slab_done < g.zs) zero_acc();
            }
        }
    }
    if (by_hand) {
        // K loop done above
    } else if (T > 0) {
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
        int in_slab = 0, i = 0;
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
                if (++in_slab == c) { in_slab = 0; meet(); if (slab_done < g.zs) zero_acc(); }
            }
        }
#pragma unroll
        for (int s = 0; s < DEPTH - 1; ++s)
            if (i + s < T) {
                compute(a_st[s], b_st[s]);
                if (++in_slab == c) { in_slab = 0; meet(); if (slab_done < g.zs) zero_acc(); }
            }
    } else {
        for (int z = 0; z < g.zs; ++z) meet();         // measurement mode without the main loop
    }

    if (EPI == EPI_PARTIAL) {
        // stored by the last meet()
    } else if (EPI == EPI_BIAS_DSWISH) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = threadIdx.x + i * 256;
            const int row = q / QROW, col = (q % QROW) * 4;
            const int m = m0 + row, n = nt0 * 16 + col;
            if (q < NQ && m < g.M) {
                const f32x4 y = summed4(row * Cfg::LDR + col) + *reinterpret_cast<const f32x4 *>(g.bias + n);
                f32x4 o;
                o.x = y.x * fast_sigmoid(y.x - 1.0f); o.y = y.y * fast_sigmoid(y.y - 1.0f);
                o.z = y.z * fast_sigmoid(y.z - 1.0f); o.w = y.w * fast_sigmoid(y.w - 1.0f);
                *reinterpret_cast<f32x4 *>(g.out + (size_t)m * g.ldo + n) = o;
            }
        }
    } else {   // EPI_LSTM: each 4-column group = gates i,f,g,o of one hidden unit
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const f32x4 gt = summed4(qo[i]) + qb[i];
            const float c_new = fast_sigmoid(gt.y) * cprev[i] + fast_sigmoid(gt.x) * fast_tanh(gt.z);
            const float u = g.debug == 2 ? gt.x + gt.y + gt.z + gt.w : fast_sigmoid(gt.w) * fast_tanh(c_new);
            if (qok[i]) { *cptr[i] = c_new; g.out[(size_t)qm[i] * g.ldo + qunit[i]] = u; }
        }
    }
    stamp(4);
}

template <int MT, int NT>
static void dispatch(const GemmArgs &g, hipStream_t s)
{
    using Cfg = TileCfg<MT, NT>;
    dim3 grid((unsigned)(g.N / Cfg::BN), (unsigned)((g.M + Cfg::BM - 1) / Cfg::BM), (unsigned)(g.kz / g.zs));
    static const int ldspad = getenv("APRIL_GEMM_LDSPAD") ? atoi(getenv("APRIL_GEMM_LDSPAD")) : 0;   // measurement: KiB of LDS to request at least (> 80 forces one workgroup per CU)
    const size_t lds = std::max((size_t)Cfg::LDS_FLOATS * sizeof(float), (size_t)ldspad * 1024);
#define LAUNCH2(E, A) do { if (g.wt == 1) hipLaunchKernelGGL((gemm_f32_kernel<MT, NT, E, A, 1>), grid, dim3(256), lds, s, g); \
                           else hipLaunchKernelGGL((gemm_f32_kernel<MT, NT, E, A, 0>), grid, dim3(256), lds, s, g); } while (0)
    if (g.epi == EPI_PARTIAL) { if (g.a_op == AOP_TANH_ADD) LAUNCH2(EPI_PARTIAL, AOP_TANH_ADD); else LAUNCH2(EPI_PARTIAL, AOP_NONE); }
    else if (g.epi == EPI_LSTM) LAUNCH2(EPI_LSTM, AOP_NONE);
    else LAUNCH2(EPI_BIAS_DSWISH, AOP_NONE);
#undef LAUNCH2
}

struct TilePlan { int mt, nt, zs; };

// Tile shape and slabs per workgroup.  Depends on M only through occupancy; numerics are tile-independent.
static TilePlan plan_tiles(int M, int N, int kz, int epi)
{
    // measurement knobs (default 0): 1/2 = smaller tiles for the fused-epilogue GEMMs (measured slower on MI355X:
    // B=256 gates 27 -> 32..36 us, the kernel is limited by operand loads per MFMA, not by occupancy);
    // 5 = 64x32 tiles for split-K GEMMs at M > 32
    static const int tune = getenv("APRIL_GEMM_TUNE") ? atoi(getenv("APRIL_GEMM_TUNE")) : 0;
    const int ntiles = N / 16;
    TilePlan t;
    t.mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    int mblocks = (M + t.mt * 16 - 1) / (t.mt * 16);
    t.nt = 4;
    while (t.nt > 1 && ((ntiles % t.nt) != 0 || (long)(ntiles / t.nt) * mblocks * kz < 256)) t.nt >>= 1;
    if (tune && epi != EPI_PARTIAL && t.mt == 4 && (long)(ntiles / t.nt) * mblocks < 512) {
        if (tune == 1) { t.mt = 2; mblocks = (M + 31) / 32; }
        else if (tune == 2 && t.nt == 4) t.nt = 2;
    }
    if (tune == 5 && epi == EPI_PARTIAL && t.mt == 4 && t.nt == 4) t.nt = 2;
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

int gemm_partials(int M, int N, int kz) { return kz / plan_tiles(M, N, kz, EPI_PARTIAL).zs; }

void launch_gemm(const GemmArgs &g_in, hipStream_t s)
{
    GemmArgs g = g_in;
    static const int dbg = getenv("APRIL_GEMM_DEBUG") ? atoi(getenv("APRIL_GEMM_DEBUG")) : 0;
    g.debug = dbg;
    static const int skew = getenv("APRIL_GEMM_SKEW") ? atoi(getenv("APRIL_GEMM_SKEW")) : 2;
    static const int asm_loop = getenv("APRIL_GEMM_ASM") ? atoi(getenv("APRIL_GEMM_ASM")) : 1;     // 0 = compiler-scheduled loop everywhere (A/B)
    const TilePlan t = plan_tiles(g.M, g.N, g.kz, g.epi);
    g.zs = t.zs;
    // hand-scheduled K loop: measured gains with one workgroup per CU (gates at B <= 256: 24.8 -> 22.9 us) and for the
    // bias+DoubleSwish GEMMs at any size (FFN-up at B = 1024: 26.5 -> 23.3 us); the LSTM-cell GEMM with two
    // co-resident workgroups per CU is faster with the compiler-scheduled loop (B = 1024: 83 vs 91 us)
    const long wgs = (long)(g.N / 64) * ((g.M + 63) / 64);
    g.asm_loop = asm_loop == 1 ? !(g.epi == EPI_LSTM && wgs >= 512) : (asm_loop != 0);
    g.skew = (long)(g.N / (16 * t.nt)) * ((g.M + 16 * t.mt - 1) / (16 * t.mt)) * (g.kz / g.zs) >= 512 ? skew : 0;   // two workgroups per CU
    const int mt = t.mt, nt = t.nt;
    if (mt == 1) { if (nt == 4) dispatch<1, 4>(g, s); else if (nt == 2) dispatch<1, 2>(g, s); else dispatch<1, 1>(g, s); }
    else if (mt == 2) { if (nt == 4) dispatch<2, 4>(g, s); else if (nt == 2) dispatch<2, 2>(g, s); else dispatch<2, 1>(g, s); }
    else { if (nt == 4) dispatch<4, 4>(g, s); else if (nt == 2) dispatch<4, 2>(g, s); else dispatch<4, 1>(g, s); }
}

}  // namespace aprilx
