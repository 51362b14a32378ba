// Device kernel interface (gfx950 only).  Launch wrappers are implemented in the
// kernels_*.hip files; engine.cc calls them on its stream.
//
// Every floating-point reduction below has a FIXED order that does not depend
// on the batch size or on the position of a row inside the batch, so a session
// stepped alone produces bit-identical state and logits to the same session
// stepped inside a batch of thousands (SURVEY.md section 7 "determinism rule").
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace aprilx {

// ---------------------------------------------------------------- GEMM
// out[M,N] = A[M,K] x W[K,N]  with W pre-packed for v_mfma_f32_16x16x4_f32:
//   Wp[((ntile*KB + kb)*64 + lane)*4 + j] = W[kb*16 + (lane>>4)*4 + j][ntile*16 + (lane&15)]
// so one 16-byte load per lane feeds four MFMA k-steps.
//
// Canonical summation per output element (a property of the LAYER, never of the batch):
//   K is cut into `kz` slabs, every slab into 4 contiguous chunks; a chunk is ONE in-order fp32 FMA chain
//   (that is what the MFMA computes); a slab is ((c0+c1)+c2)+c3; slabs are combined PAIRWISE in slab order
//   (balanced tree).  Two schedules produce exactly this:
//     GM_SLAB   the 4 waves of a workgroup own the 4 chunks of a slab and meet in LDS once per slab; a workgroup
//               walks zs consecutive slabs (1, 2, 4 or 8 as the batch grows), the slab tree lives in registers;
//               what is left (kz / zs planes) goes to a workspace and a row kernel finishes the same tree.
//               This is the small-batch schedule: kz x more workgroups stream the weights from HBM.
//     GM_FULLK  (kz >= 4) wave w owns the contiguous K quarter = slabs [w kz/4, (w+1) kz/4): it closes every chunk
//               chain in registers (S = ((c0+c1)+c2)+c3, then the first tree level for kz = 8), and the four waves
//               meet ONCE as (R0+R1)+(R2+R3).  No partial planes; the row work (state write, residual, bias,
//               sum of squares, slot store) runs in the GEMM epilogue.  Used as soon as output tiles alone give
//               enough workgroups.
//   fp16 one-chain rule (round 6; binary16 operands, wt == 1, on the GM_TILE / GM_PP schedules, kz = 1: gates, FFN up, the layer-major
//   halves of the gate GEMM): the sum is ONE MFMA chain over all 32-k blocks in k order -- each v_mfma_f32_16x16x32_f16 adds its 32
//   products to the running fp32 sum -- and where the y half of K ends (K0) the running sum is multiplied by the row's BasicNorm scale:
//   s * sum_y + sum_h.  EPI_XPART leaves s * sum_y, EPI_LSTM + p_add continues the chain from it.  No chunk sums: the accumulator is the
//   only register set, which is what lets GM_PP hold 256 x 128 tiles with double-buffered fragments.  Still a property of the layer:
//   every batch size, streamed or layer-major, runs the same chain.  fp32 operands keep the chunk / slab / tree form above.
enum GemmEpilogue {
    EPI_PARTIAL = 0,      // ws[z][m][n] = tree sum of this workgroup's slabs      (consumer: row kernels; FULLK: one plane)
    EPI_LSTM = 1,         // columns are unit-major (unit*4 + gate i,f,g,o): cell update, c in place, u out
    EPI_BIAS_DSWISH = 2,  // out = y * sigmoid(y - 1), y = acc + bias
    EPI_HR = 3,           // LSTM projection: state[slot] = acc ; out = resid * rowscale(resid) + acc
    EPI_RESID_SSQ = 4,    // y = resid + (acc + bias) (resid optional) ; out = y ; ssq[m][n/32] = sum of y^2 over 32 columns
    EPI_SLOT_STORE = 5,   // out[slot] = acc + bias, rows with row_mask[m] == 0 skipped (mask optional)
    EPI_XPART = 6         // layer-major schedule: out = (p0 + p1) (* x_scale) = the input half of the gate pre-activations
};
// A-operand prologues
enum GemmAOp { AOP_NONE = 0, AOP_TANH_ADD = 1 };
// BasicNorm never runs as a kernel.  A layer leaves y and its per-32-column sums of squares (EPI_RESID_SSQ); the consumers
// of x = y * scale(y), scale = (mean(y^2) + eps)^-1/2, fold the row scale in where it is cheapest:
//   - a GEMM over x (LSTM gates' input half, encoder_proj) runs over y and multiplies the finished partial sum by the row's
//     scale in its epilogue (`x_scale`): scale * sum_k(y_k w_k), one rounding away from sum_k((scale y_k) w_k);
//   - the residual x + h' reads y and multiplies (`r_scale`).
enum GemmMode { GM_SLAB = 0, GM_FULLK = 1, GM_TILE = 2, GM_KW = 3, GM_PP = 4 };
//     GM_TILE   (kernels_gemm_tile.hip) the four waves split the OUTPUT tile and share both operands through LDS; every wave
//               folds chunk -> slab -> tree in registers over the workgroup's zs slabs.  zs == kz: row epilogue fused;
//               zs < kz: kz / zs partial planes, finished by the row kernels exactly as for GM_SLAB.
//     GM_KW     (kernels_gemm_kw.hip, round 5) the waves split K as in GM_FULLK (eight waves: one slab each at kz = 8), but every wave's
//               activation rows arrive in full 128-byte lines through a wave-private LDS ring (no barrier in the K loop); 32 x 32 tiles.
//               The N = d_model GEMMs (and FFN up) at a few hundred rows per launch.

//     GM_PP     (kernels_gemm_pp.hip, round 6) the fp16 gates / FFN-up GEMMs (kz = 1) from a few hundred rows per launch: 256 / 128 x 128
//               tiles on eight waves = two groups of four that run one phase apart (one group's MFMAs beside the other's DMA issue and
//               fragment reads); same chains and folds as GM_TILE.

constexpr int SSQ_COLS = 32;       // columns per sum-of-squares partial (one granule = 8 consecutive 4-column quads)

struct RowScale {                  // BasicNorm scale of a row from its sum-of-squares partials (see row_scale() in device_utils.h)
    const float *ssq = nullptr;    // [rows][groups]
    int groups = 0;                // N / SSQ_COLS
    float inv_n = 0.0f;            // 1 / N
    float eps = 0.0f;
};

// Per-slot search state (reference AprilASRSession_i: context tensor, last_emission_time_ms, the class of the
// last active token; src/april_session.h:32-73); see "greedy search on the device" below.
struct GreedyState { int32_t ctx0, ctx1; int32_t last_tok; uint32_t last_emit_ms; };

constexpr int STAMP_WORDS = 144, STAMP_ENDS = 8, STAMP_NENDS = 64, STAMP_SUBS = 72;      // layout of a gates-clock slot (GemmArgs::stamp; device_utils.h)

struct GemmArgs {
    // A operand as up to two K segments with optional row indirection (slot ids)
    const float *a0 = nullptr; int lda0 = 0; const int *aidx0 = nullptr; int K0 = 0;
    const float *a1 = nullptr; int lda1 = 0; const int *aidx1 = nullptr; int K1 = 0;
    const float *a0b = nullptr;            // AOP_TANH_ADD second addend (same ld as a0; row index aidx0b, or aidx0 when same_idx_b)
    const int *aidx0b = nullptr; int same_idx_b = 1;
    // decoder-output table (engine.h): when set, the second addend's row is table row ctx0 * ctx_vocab + ctx1 of the search
    // state of the row's slot (the index aidx0b / aidx0 would have given), i.e. a0b = the table
    const GreedyState *ctx_state = nullptr; int ctx_vocab = 0;
    int a_op = AOP_NONE;
    RowScale x_scale;                      // optional (ssq != null).  EPI_LSTM: the A-segment-0 half of the sum, ((p0+p1) * scale + p2) + p3;
                                           // EPI_SLOT_STORE: the whole sum, out = acc * scale + bias
    const void *wp = nullptr;              // packed weights (fp32, or fp16 in the same element order when wt == 1)
    int wt = 0;                            // 0: fp32 operands, v_mfma_f32_16x16x4_f32; 1: fp16 weights, A rounded to fp16 on load,
                                           //    v_mfma_f32_16x16x16_f16 with fp32 accumulation (BASELINE configs[4])
    int M = 0, N = 0, K = 0;               // N multiple of 16, K multiple of 64
    int kz = 1;                            // K slabs (power of two); canonical summation unit
    int zs = 1;                            // slabs handled per workgroup (set by launch_gemm)
    int mode = GM_SLAB;                    // set by launch_gemm
    int epi = EPI_PARTIAL;
    float *out = nullptr; int ldo = 0;     // EPI_PARTIAL: workspace [planes][m_stride][N]; others: [M][ldo] (EPI_SLOT_STORE: [slots][ldo])
    // fp16-operand engines (GM_TILE, wt == 1): activations are READ as binary16 (a0 / a1 point at binary16 rows, lda in elements)
    // and the producing epilogues leave binary16 copies beside (or instead of: out == null) the fp32 rows, same leading dimensions
    void *out16 = nullptr;                 // EPI_LSTM (u), EPI_BIAS_DSWISH (ff), EPI_HR (x + h'), EPI_RESID_SSQ (y)
    void *state16 = nullptr;               // EPI_HR: binary16 copy of the h state rows [slots][ld_state]
    int m_stride = 0;
    const float *bias = nullptr;
    float *c_state = nullptr;              // EPI_LSTM: [slots][hidden] for this layer
    const int *slot_idx = nullptr;         // EPI_LSTM / EPI_HR / EPI_SLOT_STORE: row -> slot
    int hidden = 0;
    float *state = nullptr; int ld_state = 0;     // EPI_HR: h state of this layer [slots][ld_state]
    const float *resid = nullptr; int ldr = 0;    // EPI_HR / EPI_RESID_SSQ: residual rows [M][ldr]
    RowScale r_scale;                      // EPI_HR: the residual is y * scale(y)
    float *ssq_out = nullptr;              // EPI_RESID_SSQ: [M][N / SSQ_COLS]
    const int *row_mask = nullptr;         // EPI_SLOT_STORE: optional
    const int *run_flag = nullptr;         // optional device word: the kernel returns at once unless it holds run_gen
    int run_gen = 1;
    // layer-major schedule (the encoder input of T chunks is known up front): the gate GEMM is split in two launches with the
    // SAME chains -- waves 0,1 own the input half of K, waves 2,3 the recurrent half (kz = 1, K0 = K1):
    //   wave_mask 0b0011 + EPI_XPART : P = (p0 + p1) * scale for all T x sessions rows at once
    //   wave_mask 0b1100 + EPI_LSTM + p_add : ((P + p2) + p3) + bias per time step -- bit-identical to the one-launch form
    int wave_mask = 0xF;
    const float *p_add = nullptr; int ldp = 0;
    int force_fullk = 0;                   // take the full-K plan even when it yields few workgroups (latency-bound sequential steps: one launch
                                           // instead of split-K + row kernel matters more than filling the chip)
    int zcount = 1;                        // same-shape problems sharing the launch (set by stage_gemm_z): an occupancy hint for the tile planner
    int tile_ok = 0;                       // the caller planned this GEMM with gemm_fullk / gemm_partials(..., tile_ok): 1 = GM_TILE may be chosen by
                                           // the planner's occupancy rule (fp32 A in one K segment, N % 64 == 0, row epilogue or partial planes);
                                           // 2 = GM_TILE always (the fp16 tile path of an fp16 engine: every batch size runs the same chains)
    // (measurement form, APRIL_RECUR_KSPLIT, off by default) the row forms of kernels_recur.hip (<= 16 rows, EPI_HR / EPI_RESID_SSQ) cut K across `ksplit` workgroups per column granule when the caller
    // lends them a workspace: ks_ws = [N / granule columns][kz][16][granule columns] floats, ks_cnt = one zeroed word per granule, both
    // private to this problem among everything that can run beside it (the engine: per layer).  ksplit is set by launch_gemm / stage_gemm_z.
    float *ks_ws = nullptr; unsigned *ks_cnt = nullptr; int ksplit = 1;
    int skew = 0;                          // first-round start skew (x 4096 cycles; APRIL_GEMM_SKEW, off: device_utils.h first_round_skew; GM_KW: its own meaning, APRIL_KW_SKEW)
    int skew_wgs = 0;                      // workgroups of the first round (the chip's slots for this kernel); set by launch_gemm with skew
    int xcd_rc = 0;                        // GM_KW: 2 = tiles dealt to the XCDs as 2 row halves x 4 column quarters (APRIL_KW_XCD; 0 = column tiles round robin)
    int asm_loop = 0;                      // != 0: hand-scheduled K loop (gemm_mainloop_asm.inc) in the fused-epilogue 64x64 fp32 tiles
    unsigned long long *trace = nullptr;   // measurement only: per-workgroup s_memtime stamps [wg][8] (wave 0, lane 0)
    // Gates launches only (EPI_LSTM; the engine's "gates clock", aprilx_model_profile(model, 2)): one slot per LAUNCH (all problems
    // of a z-batched launch point at the same one).  Every workgroup stamps real time (s_memrealtime, 10 ns) when it starts and when it has
    // finished; the last one to finish adds (latest end - earliest start) to the slot's sum and re-arms it -- the launch's duration under
    // whatever launch path is in use (graph replay), with no event packet near the kernel.  null (the product's case): two scalar branches.
    unsigned long long *stamp = nullptr;   // slot of 128 words (device_utils.h stamp_begin / stamp_end): [0] start, [1] arrivals, [2] sum of durations (ticks), [3] launches, [8..71] end stamps
    int debug = 0;                         // measurement only: 1 = skip the MFMA main loop, 2 = skip the epilogue math, 3 = all k blocks read block 0
};
void launch_gemm(const GemmArgs &g, hipStream_t s);
// Measurement (the engine's profiling mode, bench.py's roofline clock): the NEXT GEMM kernel this thread launches records its own start
// and stop into these events -- hipExtLaunchKernel puts them into the dispatch packet, i.e. the time stamps rocprofv3 reports as the
// kernel's duration, with no event packets around the kernel.  Consumed (cleared) by that launch; gemm_profile_pending() tells whether
// a launch has taken them.
void gemm_profile_next_launch(hipEvent_t start, hipEvent_t stop);
bool gemm_profile_pending();
// (internal) GM_TILE launch, called by launch_gemm / launch_gemm_z once the plan is made: tile 16 * mt rows x 16 * nt columns; dev_args != null: n z-batched problems
void launch_gemm_tile(const GemmArgs &g, int mt, int nt, const GemmArgs *dev_args, int n, hipStream_t s);
// (internal) GM_KW launch (kernels_gemm_kw.hip): tile 16 * mt rows x 16 * nt columns, all of K in the workgroup; dev_args != null: n z-batched
// problems.  gemm_kw_waves: 8 / 4 = the waves a GM_KW workgroup would use for this GEMM (operands, epilogue, chunk structure), 0 = not eligible
int gemm_kw_waves(const GemmArgs &g);
// (internal) GM_PP launch (kernels_gemm_pp.hip): tile 16 * mt (256 / 128) rows x 128 columns; dev_args != null: n z-batched problems.
// gemm_pp_ok: the operands of g fit the schedule (binary16 operands, kz = 1, LSTM / DoubleSwish / XPART epilogue, whole stages)
bool gemm_pp_ok(const GemmArgs &g, int mt);
void launch_gemm_pp(const GemmArgs &g, int mt, const GemmArgs *dev_args, int n, hipStream_t s);
// (internal) the 256 x 192 form of GM_PP (kernels_gemm_pw.hip; TilePlan.nt = 12): N a multiple of 192, EPI_LSTM / EPI_BIAS_DSWISH, all of K
bool gemm_pw_ok(const GemmArgs &g);
void launch_gemm_pw(const GemmArgs &g, const GemmArgs *dev_args, int n, hipStream_t s);
// measurement only (tools/pp_bench): enable -1 = environment default (APRIL_GM_PP), 0 / 1 = off / on; mt = 0 (planner) or pinned 16 / 8
void gemm_pp_pin(int enable, int mt);
bool gemm_kw_has_kernel(const GemmArgs &g, int mt, int nt);      // a GM_KW kernel exists for this GEMM on 16 mt x 16 nt tiles
void launch_gemm_kw(const GemmArgs &g, int mt, int nt, const GemmArgs *dev_args, int n, hipStream_t s);
// measurement only (tools/kw_bench): enable / ff1 (FFN up on GM_KW): -1 = environment default, 0 / 1 = off / on; mt = 0 (planner) or pinned tile rows / 16
void gemm_kw_pin(int enable, int mt, int ff1);
// (internal) the recurrent GEMMs of a long feed at <= 16 rows as weight streams (kernels_recur.hip): recur_form says whether g is
// one of them (1 gates h-half + cell, 2 projection), launch_recur runs n same-shape problems (dev_args) or g itself (dev_args == null)
int recur_form(const GemmArgs &g);
void recur_ksplit_pin(int s);                    // measurement only (tools/kw_bench): -1 = environment default (APRIL_RECUR_KSPLIT), 0 = off, 1 = planner, > 1 = that cut
int recur_ksplit(const GemmArgs &g, int n);      // workgroups per column granule for the row forms (1 = whole K in one workgroup)
void launch_recur(const GemmArgs &g, int form, const GemmArgs *dev_args, int n, hipStream_t s);
// n independent GEMMs of ONE shape (same M, N, K, kz, epilogue; any pointers) in one launch.  stage_gemm_z finalizes the
// argument blocks on the host; launch_gemm_z launches once they are in device memory at dev_args (in stream order).
// Fused-epilogue forms only (EPI_LSTM, EPI_XPART, EPI_BIAS_DSWISH, and EPI_HR / EPI_RESID_SSQ where gemm_fullk says so).
void stage_gemm_z(const GemmArgs *items, int n, GemmArgs *staged);
void launch_gemm_z(const GemmArgs *staged, int n, const GemmArgs *dev_args, hipStream_t s);
// number of partial planes launch_gemm will write for an EPI_PARTIAL GEMM of this shape
// (zcount = same-shape problems sharing a z-batched launch; tile_ok = the GM_TILE schedule is allowed for this call site:
// callers pass the same values here and in GemmArgs::zcount / tile_ok so that both sides make the same plan)
int gemm_partials(int M, int N, int kz, int zcount = 1, int tile_ok = 0);
// true when launch_gemm runs a GEMM of this shape on the full-K schedule, i.e. the caller may (must, for the row
// epilogues EPI_HR / EPI_RESID_SSQ / EPI_SLOT_STORE) fuse the row work; false: EPI_PARTIAL + row kernel
bool gemm_fullk(int M, int N, int kz, bool force = false, int zcount = 1, int tile_ok = 0);
// true when a tile_ok GEMM of this shape runs on the GM_TILE schedule (whose K split is a matter of occupancy, not of need)
bool gemm_tile_planned(int M, int N, int kz, int zcount = 1);
// measurement only (tools/tile_bench): enable -1 = environment default, 0 / 1 = off / on; mt, zs = 0 (planner's choice) or pinned
void gemm_tile_pin(int enable, int mt, int zs);

// ---------------------------------------------------------------- row kernels (one workgroup per row; small-batch path)
enum RowMode {
    ROW_HR = 0,          // s = tree(ws); state[slot] = s; out = resid * rowscale(resid) + s
    ROW_RESID_SSQ = 1,   // y = resid + (s + bias) (resid optional); out = y; ssq partials of y
    ROW_SLOT_STORE = 2   // out[slot] = s (* rowscale, optional) + bias (rows with row_mask == 0 skipped)
};
struct RowArgs {
    int mode = ROW_HR;
    const float *ws = nullptr; int parts = 1; int m_stride = 0; int N = 0; int M = 0;
    const float *bias = nullptr;
    const float *resid = nullptr; int ldr = 0;
    RowScale r_scale;                      // ROW_HR: scale of the residual; ROW_SLOT_STORE: optional scale of the sum
    float *out = nullptr; int ldo = 0;
    const int *slot_idx = nullptr;         // row -> slot for state / slot-indexed outputs
    float *state = nullptr; int ld_state = 0;   // ROW_HR: h state of this layer
    float *ssq_out = nullptr;              // ROW_RESID_SSQ
    void *out16 = nullptr, *state16 = nullptr;   // fp16 tile engines: binary16 copies of out (ROW_HR, ROW_RESID_SSQ) and of the h state rows (ROW_HR)
    const int *row_mask = nullptr;         // ROW_SLOT_STORE
    const int *run_flag = nullptr;         // optional device word: the kernel returns at once unless it holds run_gen
    int run_gen = 1;
};
void launch_row(const RowArgs &r, hipStream_t s);
// n row problems of one mode and shape (same M, N, parts) in one launch: blockIdx.y picks the argument block (device array)
void launch_row_z(const RowArgs *host_args, int n, const RowArgs *dev_args, hipStream_t s);

// ---------------------------------------------------------------- greedy search on the device
// Per-slot search state (reference AprilASRSession_i: context tensor, last_emission_time_ms, the class of the
// last active token; src/april_session.h:32-73).  The host keeps the token list and runs the callbacks; the
// device keeps what the NEXT network call depends on, so a chunk needs no host decision.
// One joiner round of one session.  flags: 1 = valid (the round ran), 2 = resolved to blank, 4 = context changed
struct StepRecord { int32_t idx; float max_val; float blank_val; uint32_t flags; };
enum { REC_VALID = 1, REC_BLANK = 2, REC_CTX = 4 };
enum TokClassBits { TKC_WORD_START = 1, TKC_SENT_END = 2, TKC_COMMA = 4, TKC_DOT = 8, TKC_DIGIT_START = 16 };

struct DecEmbedParams {
    const float *emb = nullptr;            // [vocab][d]
    const float *conv_w = nullptr;         // [d][d/groups][context]
    const float *conv_b = nullptr;         // optional [d]
    int d = 0, groups = 0, context = 0, vocab = 0;
};

// Masked arg-max over the joiner logits + the blank / non-blank decision of src/april_session.c:306-429 that
// drives the data path (context push, decoder re-run, silence reset) + the decoder front end for rows whose context
// changed.  One workgroup per row.
struct DecideArgs {
    const float *ws = nullptr; int parts = 1; int m_stride = 0; int N = 0; int M = 0;   // logits = tree(ws) + bias
    int n_valid = 0;                       // vocabulary size (<= N)
    const float *bias = nullptr;
    int blank = 0;
    float early_emit = 0.0f;               // 1.0 in the first round of a chunk, 0.0 afterwards (:449-454)
    const int *slot_idx = nullptr;         // row -> slot
    const int *now_ms = nullptr;           // row -> session time of this chunk
    int *active = nullptr;                 // row -> `gen` while the row is still searching in this chunk (in/out; round 0 treats every row as active)
    int gen = 1;                           // generation of this chunk inside the step (1 for a one-chunk step, t + 1 in a layer-major step)
    int rec_slot = 0;                      // record index of this round inside the step's record block (chunk * 3 + round)
    int *dirty = nullptr;                  // row -> context changed in this round (out; the decoder projection's row mask)
    const uint8_t *tok_class = nullptr;    // [vocab] TokClassBits
    GreedyState *state = nullptr;          // [slots]
    StepRecord *rec = nullptr;             // [M] records of this round, or null (see rec_ring)
    StepRecord *rec_ring = nullptr;        // records go to rec_ring[*rec_off + round * M + row] when rec == null
    const int *rec_off = nullptr; int round = 0;
    float *logits_dump = nullptr;          // optional [M][n_valid]
    DecEmbedParams dec;
    float *de_out = nullptr; int ld_de = 0;   // relu(conv(emb[ctx])) rows for dirty rows
    // round flags (device words, zeroed by the advance kernel): run[r] == gen: some row still searches in round r (r > 0),
    // rerun[r] == gen: some row's context changed in round r.  A round nobody needs costs three empty launches.
    int *run_flags = nullptr;              // [3]
    int *rerun_flags = nullptr;            // [3]
};
void launch_decide(const DecideArgs &a, hipStream_t s);

// Decoder front end for listed slots, context taken from the device state.  op 1 first applies the end-of-flush
// reset (last token forgotten; context cleared to [blank, blank] unless it already starts with blank,
// src/april_session.c:296-301,561-563).
struct DecRowsArgs {
    const int *slot_idx = nullptr; int M = 0;
    int op = 0; int blank = 0;
    GreedyState *state = nullptr;
    DecEmbedParams dec;
    float *de_out = nullptr; int ld_de = 0;
};
void launch_dec_rows(const DecRowsArgs &a, hipStream_t s);

// debug / parity entry point: decoder front end with explicit contexts
struct DecEmbedArgs {
    DecEmbedParams dec;
    const int *ctx = nullptr;              // [M][context]
    int M = 0;
    float *out = nullptr; int ldo = 0;
};
void launch_dec_embed(const DecEmbedArgs &a, hipStream_t s);

// start of a step inside a launch chain whose arguments are fixed (hipGraph): fetches the step's index block
// (n_arrays arrays of `len[a]` ints: slots | ring tails | session times | row slots) from the pinned host ring through the
// device step counter
struct AdvanceArgs {
    const int *host_ring = nullptr;        // pinned, device-readable
    const int *host_step_off = nullptr;    // pinned: [step] -> offset of the step's block in host_ring
    const int *host_rec_off = nullptr;     // pinned: [step] -> offset of the step's records in the record ring
    int *counter = nullptr;                // device: next step (runs on for the life of the engine; table index = counter & index_mask)
    int index_mask = 0x7fffffff;
    int *dst = nullptr; int dst_stride = 0;   // device: [n_arrays][dst_stride]
    int n_arrays = 3; int len[4] = {0, 0, 0, 0};
    int *rec_off = nullptr;                // device: record offset of the current step
    int *flags = nullptr; int n_flags = 0; // device: zeroed
};
void launch_advance(const AdvanceArgs &a, hipStream_t s);

// slot reset (aas_free / session create): recurrent state, encoder/decoder outputs and the search state of one slot
struct ZeroSlotArgs {
    float *h = nullptr, *c = nullptr; int n_layers = 0; size_t slots = 0; int d_model = 0, hidden = 0;
    float *eout = nullptr, *dout = nullptr; int joiner = 0;
    GreedyState *state = nullptr; int blank = 0;
    int slot = 0;
    void *h16 = nullptr;                   // fp16 tile engines: the binary16 copy of h, zeroed too
    int n_list = 0; int list[64] = {0};    // n_list > 0: the listed slots in ONE launch (grid.y), `slot` unused
};
void launch_zero_slot(const ZeroSlotArgs &a, hipStream_t s);

// Search bookkeeping of one BLOCK of time steps of a layer-major step, so that the search's launch chain has arguments that do
// not depend on the block (one captured graph serves every block): the block's session times and encoder-output row numbers
// are copied to fixed places, the record offset of the block is derived from the step's, the round flags are cleared.
struct BlockSetupArgs {
    const int *now_src = nullptr; int *now_dst = nullptr;     // [count] session times of the block's rows
    int *rows_dst = nullptr; int row0 = 0; int count = 0;     // rows_dst[i] = row0 + i
    const int *rec_off_step = nullptr; int *rec_off_block = nullptr; int rec_add = 0;
    int *flags = nullptr; int n_flags = 0;
};
void launch_block_setup(const BlockSetupArgs &a, hipStream_t s);

// ---------------------------------------------------------------- encoder front (conv 1+2, im2col for conv 3)
struct ConvEmbedArgs {
    const float *ring = nullptr;           // [slots][ring_frames][mel]
    int ring_frames = 0, mel = 0, seg = 0;
    const int *slot_idx = nullptr;         // row -> slot
    const int *ring_tail = nullptr;        // row -> first ring row of the chunk
    const float *w[3] = {nullptr, nullptr, nullptr};
    const float *b[3] = {nullptr, nullptr, nullptr};
    int ch[3] = {0, 0, 0};
    int ch1_per_group = 0;                 // second-conv channels per workgroup (grid.y = ch[1] / this)
    int stride[3] = {1, 2, 2};
    float *out = nullptr; int ldo = 0;     // im2col rows of the third conv: [M * f_out][ldo], k = ci*9 + i*3 + j
    int M = 0;
    const float *x_direct = nullptr;       // debug path: x given as [M][seg][mel] instead of the ring
    const float *w0t = nullptr, *w1t = nullptr;      // optional: the first two convs' weights transposed (launch_conv_weight_transpose): [9][ch0], [ch0 * 9][ch1] -- the many-chunk form needs them
};
void launch_conv_embed(const ConvEmbedArgs &a, hipStream_t s);
void launch_conv_weight_transpose(const float *w0, const float *w1, int c0, int c1, float *w0t, float *w1t, hipStream_t s);

// fp32 -> fp16 (round to nearest even), elementwise; used once at load for the fp16 weight copies
void launch_cvt_f16(const float *src, void *dst, size_t n, hipStream_t s);

// Weight prefetch into the memory-side cache (round 6).  The layer weights of a model do not fit the 256 MB Infinity Cache (aprilv0 fp32:
// 0.33 GB, the larger encoder in binary16: 0.5 GB), so every launch of a step streams its weights from HBM and starts with the
// latency of that (tools/pp_bench `cold`: +3 .. 4 us per launch against weights the launch before left behind).  launch_prefetch
// reads one dword of every 128-byte line of n regions -- the NEXT launch's weights -- from a side stream while the current launch
// computes: HBM is idle then (a launch's weights are in after its first third), the lines land in the memory-side cache, and the
// next launch's first DMA stages arrive at cache latency.  No data dependency: a hint, joined only to close the graph capture.
// MEASURED AND NOT KEPT (engine.cc, APRIL_PREFETCH=1): the cross-stream edges it needs inside the captured graph cost more than it saves.
struct PrefetchItem { const void *ptr = nullptr; unsigned long long bytes = 0; };
void launch_prefetch(const PrefetchItem *dev_items, int n, hipStream_t s);
// fp32 packed weights (16-k blocks, kernels.h top) -> binary16 in the order of v_mfma_f32_16x16x32_f16's B fragment:
//   dst[((ntile * (K / 32) + kb) * 64 + lane) * 8 + j] = W[kb * 32 + (lane >> 4) * 8 + j][ntile * 16 + (lane & 15)]
// (one 16-byte read per lane and 32-k block; K a multiple of 32)
void launch_repack_x32(const float *src_packed, void *dst, int K, int N, hipStream_t s);

// ---------------------------------------------------------------- fbank
struct FbankTables {                       // device pointers
    const float *window = nullptr;         // [padded]
    const double *tw[16] = {nullptr};      // per factor twiddles (pocketfft layout)
    const double *tws[16] = {nullptr};     // factors above 5: the generic pass's roots of unity (cos, sin)(2 pi i / ip)
    int fct[16] = {0}; int nfct = 0;
    const float *mel = nullptr;            // [nbins][padded/2]
    const int *mel_lo = nullptr;           // [nbins] first non-zero fft bin
    const int *mel_hi = nullptr;           // [nbins] one past the last non-zero fft bin
    int padded = 0, nbins = 0;
};
struct FbankFrameDesc { int32_t slot; int32_t ring_row; int32_t pcm_off; };   // pcm_off < 0: padding row log(kEps)
struct FbankArgs {
    FbankTables t;
    const int16_t *pcm = nullptr;          // staged samples
    const FbankFrameDesc *desc = nullptr;
    int n_frames = 0;
    float *ring = nullptr; int ring_frames = 0;
    float pad_value = 0.0f;
};
void launch_fbank(const FbankArgs &a, hipStream_t s);

}  // namespace aprilx
