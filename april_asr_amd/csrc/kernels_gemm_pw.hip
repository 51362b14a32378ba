// GM_PP, 256 x 192 tiles (TilePlan.nt = 12) -- the wide form of the ping-pong schedule of kernels_gemm_pp.hip, round 6, BASELINE
// configs[4] (fp16 MFMA path).
//
// Why a second tile shape: GM_PP holds one workgroup per CU, so a launch costs whole rounds of one tile time.  The larger encoder's
// gates at 512 sessions are 96 tiles of 256 x 128 per layer problem; the three-problem launches of the feed wavefront (the common
// case: three chunk steps per feed) are 288 tiles = TWO rounds on 256 CUs, 50.6 us against 27.7 us for two problems (tools/pp_bench).
// At 256 x 192 the same launch is 192 tiles: one round, every tile 1.5 x the work.  N = 4 x cell = 6144 = 32 x 192.
//
// Schedule: as GM_PP -- eight waves = two groups of four running one phase apart (load phase: DMA pieces only; compute phase: the
// MFMAs of k block j with the fragment reads of k block j + 1 spread between them into a second register set), group g owns rows
// [128 g, 128 g + 128), its four waves split them 2 x 2: a wave tile is 64 x 96 = 24 v_mfma_f32_16x16x32_f16 per k block against
// 10 fragment reads (GM_PP: 16 against 8).  What differs is the LDS ring: a 64-k stage of this tile is 56 KB and three of them do not
// fit 160 KB, so the ring holds FIVE single k blocks (32 k: 256 rows x 64 B of activations + 12 weight pieces = 28 KB each, 140 KB):
//
//     barrier index   0      1      2      3      4     ...
//     group 0         | L0   | C0   | L1   | C1   | L2   ...        L_j: DMA pieces of k block j + 4 -> buffer (j + 4) % 5
//     group 1         | -    | L0   | C0   | L1   | C1   ...        C_j: MFMAs of k block j, fragment reads of k block j + 1
//
// Buffer (j + 4) % 5 = (j - 1) % 5 was last read in C_{j-2} (k block j - 1), whose reads group 1 retired (lgkmcnt(0)) in front of
// barrier 2 j - 1; L_j begins behind barrier 2 j (group 0) / 2 j + 1 (group 1).  K block j + 1 has landed when it is first read (C_j
// of group 0, behind barrier 2 j + 1): group 0 waits for its own pieces of it at the end of L_j (three younger k blocks stay in
// flight: counted vmcnt), group 1 at the end of C_{j-1} (two younger ones) -- both in front of that barrier.  A k block is issued
// 3.5 phases pairs before its first read (GM_PP: two 64-k stages = the same bytes in flight).
//
// LDS image of one k block:  A: [256 rows][64 B], 16-byte segment s of row R stored at segment s ^ f((R >> 2) & 3), f = 0, 2, 3, 1:
// ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32), i.e. per bank quarter (R & 3) a group
// holds rows R >> 2 = 0, 3 of one k segment and 1, 2 of the next -- f makes their four slots distinct, so a group covers the 64 banks
// once (the plain XOR with R >> 2 does not); the DMA writes LDS linearly, so the swizzle is applied to the per-lane SOURCE address.
// B: [12 n tiles][1 KB] in the packed weight order (launch_repack_x32).
//
// Summation: the fp16 one-chain rule (kernels.h) -- one MFMA chain over all k blocks in k order, multiplied by the row's BasicNorm
// scale where the y half of K ends.  Bit-identical to GM_TILE / GM_PP at every other tile shape (tools/pp_bench,
// tests/test_gpu_gemm_pp.py).  Epilogue: the sums go through an LDS plane 128 rows at a time (256 x 196 floats do not fit beside
// nothing either), group by group; EPI_LSTM (two A segments [y16 | h16(slot)], cell update) and EPI_BIAS_DSWISH.
// Replaces the ORT MatMul nodes of the encoder's LSTM / FFN blocks (reference call site src/april_session.c:131-148).
#include "kernels.h"
#include "device_utils.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace aprilx {

namespace {

using h4 = __attribute__((ext_vector_type(4))) _Float16;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ h4 to_h4(const f32x4 &v) { return h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; }

struct PWGeom {
    static constexpr int NT = 12, NB = 5, NW = 8, NTH = 64 * NW;
    static constexpr int BM = 256, BN = 16 * NT, LDR = BN + 4;
    static constexpr int A_BYTES = BM * 64, B_BYTES = NT * 1024, KB_BYTES = A_BYTES + B_BYTES;      // one k block (32 k)
    static constexpr int RING_BYTES = NB * KB_BYTES;
    static constexpr int MTW = 4, NTW = 6;                         // MFMA tiles per wave: 64 x 96
    static constexpr int PLANE_BYTES = (BM / 2) * LDR * 4;         // the epilogue's plane: one group's 128 rows
    static constexpr int LDS_BYTES = RING_BYTES + BM * 4;          // + the rows' BasicNorm scales
    static_assert(PLANE_BYTES <= RING_BYTES, "the plane reuses the ring");
};

template <int N> __device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T *gp(T *p) { return (__attribute__((address_space(1))) T *)p; }
template <int N> using ic = std::integral_constant<int, N>;
__device__ __forceinline__ int seg_swz(int rq) { return (0x78 >> (2 * rq)) & 3; }      // f of the LDS image: 0, 2, 3, 1

template <int EPI, int DEF>
__device__ __forceinline__ void gemm_pw_body(const GemmArgs &g)
{
    using G = PWGeom;
    constexpr int BM = G::BM, BN = G::BN, NT = G::NT, MTW = G::MTW, NTW = G::NTW, NTH = G::NTH, LDR = G::LDR, NB = G::NB, KBB = G::KB_BYTES;
    static_assert(EPI == EPI_LSTM || EPI == EPI_BIAS_DSWISH, "GM_PP 256 x 192: gates, FFN up");
    extern __shared__ __attribute__((aligned(1024))) float red[];
    char *lds = reinterpret_cast<char *>(red);

    if (g.run_flag && *gp(g.run_flag) != g.run_gen) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int wrow = (grp * 2 + wm) * 64;                 // first tile row of this wave
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int KB = g.K / 32;                              // k blocks (even, >= 4: checked on the host)

    // ---- BasicNorm scales of the tile's rows: the sum-of-squares partials make one trip from global memory at kernel start
    const RowScale &rsc = g.x_scale;
    const bool fold_scale = EPI == EPI_LSTM && rsc.ssq != nullptr;
    float *scl = red + G::RING_BYTES / 4;
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

    // ---- the epilogue's thread -> cell map: thread t owns the 4-column quad t % 16 of each of the three 64-column blocks, in rows
    // t / 16 + 32 i of either 128-row half (24 quads); EPI_LSTM: the rows' slots, loaded FIRST, in front of every DMA piece
    const int qc = threadIdx.x & 15, rr = threadIdx.x >> 4;
    int l_slot[EPI == EPI_LSTM ? 8 : 1];
    if (EPI == EPI_LSTM) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int r = m0 + rr + 32 * i;
            if (r >= g.M) r = g.M - 1;                    // (padding rows read the last row's cell: never stored)
            l_slot[i] = gp(g.slot_idx)[r];
        }
    }

    // ---- DMA pieces of this wave and k block: activation pieces P = wave, wave + 8 (rows 16 P .. 16 P + 15: lane -> row 16 P +
    // (lane >> 2), LDS segment lane & 3, i.e. source segment (lane & 3) ^ f((row >> 2) & 3)); weight pieces (n tile) wave and, for the
    // waves of group 0, 8 + wave.  Sources are wave-uniform buffer descriptors + 32-bit lane offsets + scalar k offsets.
    const bool two_seg = g.K1 > 0;
    const int seg1_kb = two_seg ? g.K0 / 32 : 0x7fffffff;
    unsigned aoff[2];
    int arows1[2];
    unsigned gsegv;
    {
        int arows[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = m0 + (wave + 8 * i) * 16 + (lane >> 2);
            arows[i] = row >= g.M ? g.M - 1 : row;       // padding rows recompute the last row; never stored
            arows1[i] = arows[i];
        }
        if (g.aidx0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) arows[i] = gp(g.aidx0)[arows[i]];
        }
        if (two_seg && g.aidx1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) arows1[i] = gp(g.aidx1)[arows1[i]];
        }
        gsegv = (unsigned)(((lane & 3) ^ seg_swz((lane >> 4) & 3)) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) aoff[i] = (unsigned)arows[i] * (unsigned)g.lda0 * 2u + gsegv;
    }
    constexpr int RSRC_FLAGS = 0x00020000;                 // raw buffer, 32-bit data format (gfx9 dword 3)
    const char *wbase = reinterpret_cast<const char *>(g.wp) + (size_t)(blockIdx.x * NT + wave) * KB * 1024;
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(g.a0)), 0, -1, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(two_seg ? g.a1 : g.a0)), 0, -1, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wbase), 0, -1, RSRC_FLAGS);
    const int woff = lane * 16;
    const int w2 = 8 * KB * 1024;                          // n tile 8 + wave
    int issued = 0, seg_base = 0;
    __amdgpu_buffer_rsrc_t rs_a = rs_a0;
    auto issue_kb = [&](int buf) {                         // this wave's pieces of k block `issued`, into ring buffer buf
        if (issued == seg1_kb) {                           // (once per tile: the activation pieces move on to segment 1)
            rs_a = rs_a1; seg_base = seg1_kb;
#pragma unroll
            for (int i = 0; i < 2; ++i) aoff[i] = (unsigned)arows1[i] * (unsigned)g.lda1 * 2u + gsegv;
        }
        char *db = lds + buf * KBB;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void *)(db + (wave + 8 * i) * 1024), 16, (int)aoff[i], (issued - seg_base) * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(db + G::A_BYTES + wave * 1024), 16, woff, issued * 1024, 0, 0);
        if (grp == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(db + G::A_BYTES + (8 + wave) * 1024), 16, woff, w2 + issued * 1024, 0, 0);
        ++issued;
    };

    // ---- fragment addresses inside a ring buffer
    const int mrow = lane & 15, kq = lane >> 4;
    const int a_rd = (wrow + mrow) * 64 + ((kq ^ seg_swz((mrow >> 2) & 3)) << 4);
    const int b_rd = G::A_BYTES + wn * NTW * 1024 + lane * 16;

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int scale_at = fold_scale ? KB / 2 : -1;         // the k block in front of which the y half is complete
    auto apply_scale = [&]() {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const f32x4 xr = *reinterpret_cast<const f32x4 *>(scl + wrow + mt * 16 + kq * 4);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = acc[mt][nt][r] * xr[r];
        }
    };

    // ---- k blocks 0 .. 3 are on their way before anything else of the prologue waits for memory
    issue_kb(0); issue_kb(1); issue_kb(2); issue_kb(3);

    // EPI_LSTM: the previous cell values of this thread's cells in the FIRST 128-row half, behind the first DMA pieces (their slots were
    // fetched in front of them: memory operations retire in order, so waiting for the slots does not wait for the DMA); the second half's
    // and the rest of the first are fetched behind the K loop, in front of the plane's barriers (24 registers across the loop would spill)
    constexpr int PRE = 2;                                 // rows (of 4) of the first half fetched here
    constexpr int CPV = EPI == EPI_LSTM ? 3 * PRE : 0;
    const int l_unit0 = (n0 >> 2) + qc;                    // + 16 cb
    float l_cprev[2][EPI == EPI_LSTM ? 12 : 1];
    if (EPI == EPI_LSTM) {
#pragma unroll
        for (int i = 0; i < PRE; ++i)
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) l_cprev[0][i * 3 + cb] = gp(g.c_state)[(size_t)l_slot[i] * g.hidden + l_unit0 + 16 * cb];
    }

    if (fold_scale) {
        // partials (in registers since the first instruction of the kernel) -> LDS, BM threads add them in column order (the order of
        // row_scale()); rows padded to Gn + 1 floats.  Staged in ring buffer 4, which takes its first DMA piece behind barrier 0.
        const int Gn = rsc.groups;
        float *part = reinterpret_cast<float *>(lds + 4 * KBB);
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
        // (raw barriers: __syncthreads() would drain every DMA piece and the cell prefetch in flight)
        wait_lgkm0(); __builtin_amdgcn_s_barrier();
        if (threadIdx.x < BM) {
            float t = 0.0f;
            for (int j = 0; j < Gn; ++j) t += part[threadIdx.x * (Gn + 1) + j];
            scl[threadIdx.x] = __builtin_amdgcn_rsqf(t * rsc.inv_n + rsc.eps);
        }
        wait_lgkm0(); __builtin_amdgcn_s_barrier();
    }
    // k block 0 has landed (this wave's pieces); younger than it: k blocks 1 .. 3 and the cell prefetch.  Group 1 waits for k block 1
    // as well: group 0 reads it behind barrier 1, and group 1's first counted wait (end of C_0) is for k block 2
    if (grp == 0) wait_vm<3 * 4 + CPV>(); else wait_vm<2 * 3 + CPV>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();                          // barrier 0

    f32x4 fa0[MTW], fb0[NTW], fa1[MTW], fb1[NTW];
    auto read_frags = [&](const char *sb, f32x4 (&fa)[MTW], f32x4 (&fb)[NTW]) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) fa[mt] = *reinterpret_cast<const f32x4 *>(sb + a_rd + mt * 1024);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) fb[nt] = *reinterpret_cast<const f32x4 *>(sb + b_rd + nt * 1024);
    };
    read_frags(lds, fa0, fb0);                             // k block 0
    wait_lgkm0();
    if (grp == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one phase behind group 0

    int rb = 1, ib = 4, jkb = 0;                           // ring buffers of k block j + 1 (read) and j + 4 (issue)
    // One k block of one group: load phase | barrier | compute phase | barrier.  AH0 / AH1: the k blocks of this wave's pieces that may
    // stay in flight behind group 0's wait (end of the load phase: k block j + 1 landed) / group 1's (end of the compute phase: k block
    // j + 2 landed); EX: other loads in flight among them (the cell prefetch, during the first k blocks)
    constexpr int NR = MTW + NTW;
    // DEF (0 = the product, 2 = measurement form, see launch_gemm_pw) rows of a wave's 4 x 6 MFMA tiles are DEFERRED: their MFMAs of k block j
    // issue in the load phase of k block j + 1 (the fragments of k block j are still in their register set: the reads of k block j + 2
    // that overwrite it issue behind the barrier that ends that load phase), so both waves of a SIMD issue MFMAs in every phase.  The
    // chain of every accumulator keeps its k order.  Measured: no gain.
    constexpr int MC = MTW - DEF;                          // rows whose MFMAs stay in the compute phase
    auto deferred = [&](f32x4 (&fa)[MTW], f32x4 (&fb)[NTW]) {
#pragma unroll
        for (int mt = MC; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[mt]), __builtin_bit_cast(h8, fb[nt]), acc[mt][nt], 0, 0, 0);
    };
    auto kblock = [&](const bool do_issue, auto first, auto ah0, auto ah1, auto ex0, auto ex1, f32x4 (&fa)[MTW], f32x4 (&fb)[NTW], f32x4 (&fan)[MTW], f32x4 (&fbn)[NTW]) {
        constexpr int AH0 = decltype(ah0)::value, AH1 = decltype(ah1)::value, EX0 = decltype(ex0)::value, EX1 = decltype(ex1)::value;
        if (do_issue) issue_kb(ib);
        if constexpr (DEF > 0 && !decltype(first)::value) deferred(fan, fbn);      // (the set that was current one k block ago)
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) wait_vm<AH0 * 4 + EX0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (jkb == scale_at) { apply_scale(); __builtin_amdgcn_sched_barrier(0); }
        // (the reads are unconditional -- behind the last k block they fetch a buffer nobody needs -- so that the phase is ONE scheduling region)
        read_frags(lds + rb * KBB, fan, fbn);
#pragma unroll
        for (int mt = 0; mt < MC; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[mt]), __builtin_bit_cast(h8, fb[nt]), acc[mt][nt], 0, 0, 0);
        // issue order: MFMA(s), read, ... (every accumulator takes one MFMA per k block: their order is free)
        constexpr int NMC = MC * NTW, PER = NMC >= 2 * NR ? 2 : 1, NI = NMC / PER < NR ? NMC / PER : NR;
#pragma unroll
        for (int i = 0; i < NI; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, PER, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        if constexpr (NMC > PER * NI) __builtin_amdgcn_sched_group_barrier(0x008, NMC - PER * NI, 0);
        if constexpr (NR > NI) __builtin_amdgcn_sched_group_barrier(0x100, NR - NI, 0);
        __builtin_amdgcn_sched_barrier(0);                 // (the MFMAs stay IN FRONT of the wait: they do not depend on the reads in flight)
        wait_lgkm0();                                      // the fragments of k block j + 1 are in (and this wave's reads of their buffer are done)
        if (grp == 1) wait_vm<AH1 * 3 + EX1>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ++jkb;
        rb = rb == NB - 1 ? 0 : rb + 1;
        ib = ib == NB - 1 ? 0 : ib + 1;
    };
    using T1 = std::true_type; using F0 = std::false_type;
    // k blocks 0 .. 3: the cell prefetch sits between k block 3 and k block 4 in the memory pipe.  Group 0's wait in L_j lets k blocks
    // j + 2 .. j + 4 stay out (j = 0, 1, 2: the prefetch is among them), group 1's in C_j k blocks j + 3, j + 4 (j = 0, 1).
    kblock(true, T1{}, ic<3>{}, ic<2>{}, ic<CPV>{}, ic<CPV>{}, fa0, fb0, fa1, fb1);
    kblock(true, F0{}, ic<3>{}, ic<2>{}, ic<CPV>{}, ic<CPV>{}, fa1, fb1, fa0, fb0);
    kblock(true, F0{}, ic<3>{}, ic<2>{}, ic<CPV>{}, ic<0>{}, fa0, fb0, fa1, fb1);
    kblock(true, F0{}, ic<3>{}, ic<2>{}, ic<0>{}, ic<0>{}, fa1, fb1, fa0, fb0);
    for (int j = 4; j + 4 < KB; j += 2) {
        kblock(true, F0{}, ic<3>{}, ic<2>{}, ic<0>{}, ic<0>{}, fa0, fb0, fa1, fb1);
        kblock(true, F0{}, ic<3>{}, ic<2>{}, ic<0>{}, ic<0>{}, fa1, fb1, fa0, fb0);
    }
    // the last four k blocks issue nothing: what may stay in flight shrinks
    kblock(false, F0{}, ic<2>{}, ic<1>{}, ic<0>{}, ic<0>{}, fa0, fb0, fa1, fb1);      // j = KB - 4: group 0 needs KB - 3 (KB - 2, KB - 1 out), group 1 KB - 2 (KB - 1 out)
    kblock(false, F0{}, ic<1>{}, ic<0>{}, ic<0>{}, ic<0>{}, fa1, fb1, fa0, fb0);
    kblock(false, F0{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, fa0, fb0, fa1, fb1);
    kblock(false, F0{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, fa1, fb1, fa0, fb0);
    if constexpr (DEF > 0) deferred(fa1, fb1);             // the deferred rows of the last k block
    if (grp == 0) __builtin_amdgcn_s_barrier();
    if (jkb == scale_at) apply_scale();

    // ---- epilogue: the sums -> LDS plane (128 rows = one group's, each wave its own part) -> 4-column quads per thread
    if (EPI == EPI_LSTM) {
#pragma unroll
        for (int i = PRE; i < 8; ++i)
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) l_cprev[i >> 2][(i & 3) * 3 + cb] = gp(g.c_state)[(size_t)l_slot[i] * g.hidden + l_unit0 + 16 * cb];
    }
    f32x4 l_bias[3];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) l_bias[cb] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(gp(g.bias) + n0 + cb * 64 + qc * 4);
    __syncthreads();                                       // the last k block has been read by every wave
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        if (grp == ps) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[(wm * 64 + mt * 16 + kq * 4 + r) * LDR + (wn * NTW + nt) * 16 + mrow] = acc[mt][nt][r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = rr + 32 * i, m = m0 + ps * 128 + rl;
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(red + rl * LDR + cb * 64 + qc * 4);
                const int ncol = n0 + cb * 64 + qc * 4;
                if (EPI == EPI_BIAS_DSWISH) {
                    if (m < g.M) {
                        const f32x4 y = v + l_bias[cb];
                        f32x4 o;
                        o.x = y.x * fast_sigmoid(y.x - 1.0f); o.y = y.y * fast_sigmoid(y.y - 1.0f);
                        o.z = y.z * fast_sigmoid(y.z - 1.0f); o.w = y.w * fast_sigmoid(y.w - 1.0f);
                        if (g.out) *reinterpret_cast<__attribute__((address_space(1))) f32x4 *>(gp(g.out) + (size_t)m * g.ldo + ncol) = o;
                        if (g.out16) *reinterpret_cast<__attribute__((address_space(1))) h4 *>(gp(reinterpret_cast<_Float16 *>(g.out16)) + (size_t)m * g.ldo + ncol) = to_h4(o);
                    }
                } else {   // EPI_LSTM: the quad = gates i, f, g, o of one hidden unit; the BasicNorm scale of the y half was folded in at scale_at
                    const f32x4 gt = v + l_bias[cb];
                    const int l_unit = ncol >> 2;
                    const float c_new = fast_sigmoid(gt.y) * l_cprev[ps][i * 3 + cb] + fast_sigmoid(gt.x) * fast_tanh(gt.z);
                    const float u = fast_sigmoid(gt.w) * fast_tanh(c_new);
                    if (m < g.M) {
                        gp(g.c_state)[(size_t)l_slot[ps * 4 + i] * g.hidden + l_unit] = c_new;
                        if (g.out) gp(g.out)[(size_t)m * g.ldo + l_unit] = u;
                        if (g.out16) gp(reinterpret_cast<_Float16 *>(g.out16))[(size_t)m * g.ldo + l_unit] = (_Float16)u;
                    }
                }
            }
        }
        if (ps == 0) __syncthreads();                      // the plane is rewritten by group 1
    }
}

template <int EPI, int DEF>
__global__ __launch_bounds__(512, 1) void gemm_pw_kernel(GemmArgs g)
{
    if constexpr (EPI == EPI_LSTM) stamp_begin(g.stamp, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
    gemm_pw_body<EPI, DEF>(g);
    if constexpr (EPI == EPI_LSTM) stamp_end(g.stamp, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// n independent same-shape problems in one launch (see gemm_f32_zkernel): blockIdx.z picks the argument block
template <int EPI, int DEF>
__global__ __launch_bounds__(512, 1) void gemm_pw_zkernel(const GemmArgs *__restrict__ zargs)
{
    const GemmArgs g = zargs[blockIdx.z];
    if constexpr (EPI == EPI_LSTM) stamp_begin(g.stamp, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
    gemm_pw_body<EPI, DEF>(g);
    if constexpr (EPI == EPI_LSTM) stamp_end(g.stamp, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

template <int EPI, int DEF>
void launch_pw_one(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    using G = PWGeom;
    dim3 grid((unsigned)(g.N / G::BN), (unsigned)((g.M + G::BM - 1) / G::BM), (unsigned)(dev_args ? n : 1));
    // dynamic LDS beyond 64 KB has to be announced, per instantiation AND per device (one engine per GPU)
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_pw_kernel<EPI, DEF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_pw_zkernel<EPI, DEF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    if (dev_args) APRIL_LAUNCH((gemm_pw_zkernel<EPI, DEF>), grid, dim3(G::NTH), (size_t)G::LDS_BYTES, s, dev_args);
    else APRIL_LAUNCH((gemm_pw_kernel<EPI, DEF>), grid, dim3(G::NTH), (size_t)G::LDS_BYTES, s, g);
}

}  // namespace

// can this GEMM run on the 256 x 192 form of GM_PP?  (operand shapes only; the planner decides whether it should)
bool gemm_pw_ok(const GemmArgs &g)
{
    if (g.wt != 1 || g.kz != 1 || g.a_op != AOP_NONE || g.wave_mask != 0xF) return false;
    if (g.epi != EPI_LSTM && g.epi != EPI_BIAS_DSWISH) return false;
    if (g.N % 192 != 0 || g.K % 64 != 0 || g.K < 256) return false;          // an even number of k blocks, the last four peeled, the first four in the prologue
    if (g.K1 > 0 && (g.K0 != g.K1 || g.K0 % 32 != 0 || g.K0 + g.K1 != g.K)) return false;
    if (g.K1 == 0 && g.K0 != g.K) return false;
    if (g.epi == EPI_LSTM && g.x_scale.ssq) {
        if (g.K1 == 0) return false;                                         // (the scale belongs to the y half)
        if ((size_t)PWGeom::BM * (g.x_scale.groups + 1) * 4 > (size_t)PWGeom::KB_BYTES) return false;      // the partials are staged in one ring buffer
    }
    if ((uint64_t)g.M * (uint64_t)g.lda0 * 2 >= (1ull << 31)) return false;
    return true;
}

void launch_gemm_pw(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s)
{
    if (!gemm_pw_ok(g)) { fprintf(stderr, "libapril(mi355x): launch_gemm_pw: no kernel for epi %d (wt %d kz %d N %d K %d)\n", g.epi, g.wt, g.kz, g.N, g.K); abort(); }
    // APRIL_PW_DEFER=2: MEASUREMENT FORM (off).  Half of a wave's MFMAs of k block j issue in the load phase of k block j + 1, so that both
    // waves of a SIMD feed the matrix pipe in every phase (two waves sustain 0.87 of it, one 0.73: tools/mfma_peak).  Bit-identical; measured
    // 36.1 / 37.4 / 41.2 us against 35.6 / 36.6 / 39.7 (gates, 512 rows x 1 / 2 / 3): no gain -- the phases are not set by the issue rate
    // of a lone wave (deferring one row of four: 39.9 / 41.5 / 44.5, with spills)
    static const int defer = [] { const char *e = getenv("APRIL_PW_DEFER"); return e && *e && atoi(e) == 2 ? 2 : 0; }();
    if (g.epi == EPI_LSTM) {
        if (defer == 2) launch_pw_one<EPI_LSTM, 2>(g, dev_args, n, s);
        else launch_pw_one<EPI_LSTM, 0>(g, dev_args, n, s);
    } else {
        if (defer == 2) launch_pw_one<EPI_BIAS_DSWISH, 2>(g, dev_args, n, s);
        else launch_pw_one<EPI_BIAS_DSWISH, 0>(g, dev_args, n, s);
    }
}

}  // namespace aprilx
