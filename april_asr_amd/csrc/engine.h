// Per-GPU engine: packed weights + slot-indexed persistent session state in HBM
// and the batched step functions that replace the three ORT Run() calls of the
// reference (src/april_session.c:131-179) and its per-session fbank
// (src/fbank.c:174-306).
//
// HBM layout (fp32, row-major):
//   h      [L][slots][d_model]      LSTM projected hidden state      (reference h tensor (L,1,d))
//   c      [L][slots][hidden]       LSTM cell state                  (reference c tensor (L,1,H))
//   ring   [slots][ring_frames][mel]  log-mel feature ring           (reference OnlineFBank ring)
//   eout   [slots][joiner]          projected encoder output of the last chunk
//   dout   [slots][joiner]          projected decoder output of the current token context
//   gstate [slots]                  greedy-search state (token context, last emission time, last token)
// plus per-step work buffers sized for `max_batch` rows.  State never leaves HBM
// between feed calls; per joiner round 16 bytes per session come back (StepRecord), once per flight.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <deque>
#include <array>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include "fbank_tables.h"
#include "kernels.h"
#include "host_pool.h"
#include "model_loader.h"

namespace aprilx {

// Packed, device-layout weights as ONE contiguous blob (so they can be broadcast with a
// single collective and uploaded with a single copy).  Offsets are in floats.
struct PackedLayout {
    NetDims dims;
    size_t conv_w[3], conv_b[3];
    int k3 = 0;                              // K of the third conv as a GEMM (conv_ch[1]*9 rounded up to 64)
    size_t w_embed, b_embed;
    struct Layer { size_t wg, bg, whr, wff1, bff1, wff2, bff2; };
    std::vector<Layer> layers;
    size_t w_encproj, b_encproj, emb, dec_conv, dec_conv_b, w_decproj, b_decproj, w_out, b_out;
    size_t total = 0;
    int vocab_pad = 0;                       // joiner N padded to 16
    bool has_dec_conv_b = false;
    std::vector<float> norm_eps;             // [L]
    float embed_eps = 0;
};
void plan_layout(const NetDims &d, bool has_dec_conv_b, PackedLayout &L);
// (offset, count) of every MFMA-packed Linear / LSTM weight matrix inside the blob, in blob order: the sections that get
// binary16 copies in fp16-operand mode and are stored as binary16 in the fp16 cache file
std::vector<std::pair<size_t, size_t>> gemm_sections(const PackedLayout &L);
// fills `blob` (L.total floats) from the neutral host weights
void pack_weights(const HostModel &m, PackedLayout &L, std::vector<float> &blob);

// Process-wide lock between graph captures and everything that uses the LEGACY stream or allocates (hipMemcpy, hipMemset, hipMalloc,
// hipFree ...).  While any stream of the process is being captured, such a call from another thread fails
// ("operation would make the legacy stream depend on a capturing blocking stream" / hipErrorStreamCaptureUnsupported) and poisons the
// capture -- measured on ROCm 7.2 also for non-blocking streams and relaxed-mode captures -- and client threads make such calls: a
// second model being loaded, aprilx_session_read_frames / aprilx_session_context on an idle session, blob export.  Captures hold the
// lock for their few milliseconds, the legacy-stream users for theirs.  Recursive: the debug entry points hold it and may capture.
std::recursive_mutex &hip_legacy_mutex();
struct HipLegacyLock { std::lock_guard<std::recursive_mutex> g; HipLegacyLock() : g(hip_legacy_mutex()) {} };

struct EngineConfig {
    int device = 0;
    int max_slots = 4096;
    int max_batch = 2048;
    int precision = 0;                       // 0: fp32 GEMMs; 1: fp16 weights + fp16-rounded activations, fp32 accumulate (APRIL_PRECISION=f16)
};

struct KernelTiming { double ms = 0; long launches = 0; };

class Engine {
public:
    Engine(const EngineConfig &cfg, const PackedLayout &layout, const float *blob_host, const float *blob_device,
           const ModelParams &params, const FbankHostTables &ft, const std::vector<uint8_t> &tok_class);
    ~Engine();

    int device() const { return cfg_.device; }
    int max_batch() const { return cfg_.max_batch; }
    int max_slots() const { return cfg_.max_slots; }
    int ring_frames() const { return ring_frames_; }
    const NetDims &dims() const { return L_.dims; }
    hipStream_t stream() const { return stream_; }
    const float *weights_device() const { return w_; }
    float *weights_mut() { return w_; }        // load time only: the constructor leaves the weights unset when it is given no blob
    void finish_weights();                     // after the weights are in place (upload, copy or RCCL broadcast): derived copies (fp16)
    int precision() const { return cfg_.precision; }
    // GM_TILE for the fp32-A row-epilogue GEMMs (embed, projection, FFN down, encoder_proj, decoder projection): by the planner's
    // occupancy rule on fp32 engines, never on fp16 engines (whose LAYER GEMMs take the fp16 tile path through lin16, always)
    int tile_ok() const { return cfg_.precision == 0 ? 1 : 0; }
    bool f16_tile() const { return f16_tile_; }
    const PackedLayout &layout() const { return L_; }

    int alloc_slot();                 // -1 when full; state zeroed (reference calloc, april_session.c:40-58)
    void free_slot(int slot);
    int live_slots() const { return live_.load(std::memory_order_relaxed); }      // (read by aas_create_session's least-loaded placement on any thread)

    // ---- batched hot path (one stepping thread).  A FLIGHT is everything enqueued between two host waits: frames are cut,
    // chunk steps (encoder + the three joiner/decision/decoder rounds, all on the device) and decoder refreshes are queued
    // back to back, and the host reads the 16-byte-per-round records once, at end_flight().
    // pcm arrives as `n_parts` windows that are gathered straight into pinned staging (total n_pcm samples)
    void fbank(int n_frames, const FbankFrameDesc *desc, const std::pair<const int16_t *, size_t> *parts, size_t n_parts, size_t n_pcm,
               HostPool *pool = nullptr);
    void begin_flight();
    bool flight_has_room(int rows, int nsteps = 1) const;   // `nsteps` more steps with `rows` rows in total fit into the index / record rings
    // one chunk for m sessions (m <= max_batch): returns the step's index inside the flight.  logits_out (tests): when
    // non-null the step runs eagerly, waits, and returns the logits of the three rounds [3][m][vocab]
    int step(int m, const int *slots, const int *ring_tails, const int *now_ms, float *logits_out = nullptr);
    // Layer-major step: T consecutive chunks of each of m sessions (T * m <= max_batch rows) in one
    // go.  Everything that does not depend on the recurrence -- conv front end, the INPUT half of every layer's gate GEMM, the
    // feed-forward blocks, encoder_proj -- runs once over all T * m rows; only the recurrent half of the gates and the
    // projection run per time step.  Same chains, same order: logits and state are bit-identical to T one-chunk steps.
    // ring_tails / now_ms are [T][m].  Records: [T][3][m]; logits_out (tests) [T][3][m][vocab].
    // mode 1: the T chunk steps of one feed (T >= 2) as a wavefront over the layers -- the one-launch gates GEMM per chunk as in
    // step(), the same launch of the active layers z-batched (run_sw_chain); mode 0: layer-major (long feeds).
    int lm_step(int m, int T, const int *slots, const int *ring_tails, const int *now_ms, float *logits_out = nullptr, int mode = 0);
    int lm_max_rows() const { return cfg_.max_batch; }
    // decoder output refresh for listed slots from the context held on the device; op 1 = end-of-flush reset first
    void decode_rows(int n, const int *slots, int op);
    // end of a flight in two halves, so that the next flight can be enqueued before this one has run: close_flight() queues
    // the record copy and an event and returns the flight's parity (0 / 1); wait_flight(parity) blocks until the GPU has
    // passed that event, flight_done() polls it.  At most TWO flights are open at any time (begin_flight() reuses the parity
    // of the flight before the previous one: its records must have been read by then).
    int close_flight();
    bool flight_done(int parity);
    void wait_flight(int parity);
    void end_flight();                          // close + wait (records -> host)
    bool profiling() const { return profiling_; }
    // the scheduler's hint for the flight being launched: another flight is in the air (or follows at once), so a feed is worth
    // splitting over the three streams; a lone flight runs on one stream (the events between the streams cost it ~50 us)
    void set_overlap_hint(bool on) { overlap_hint_ = on; }
    const StepRecord *records(int step_index) const { return rec_h_ + rec_off_h_[step_index]; }   // [3][m], valid after end_flight()
    void sync();                               // stepping thread (or under capture_mu_): waits for the three streams and clears the cross-stream dependency flags
    void sync_streams();                       // any thread: waits for the three streams, nothing else

    // ---- debug / parity entry points (state passed explicitly, like the ORT tensors)
    void debug_encoder(int n, const float *x, const float *h, const float *c, float *eout, float *h2, float *c2);
    void debug_decoder(int n, const int64_t *ctx, float *dout);
    void debug_joiner(int n, const float *eout, const float *dout, float *logits);
    // parity tests of the device's copy of the search decision: one decide_kernel round (op 0) or the end-of-flush reset
    // (op 1) on slots 0..n-1 with GIVEN logits rows and search states; returns the records and the new states
    void debug_decide(int n, int op, const float *logits, float early_emit, const int *now_ms, int round, int32_t *state_io, StepRecord *rec_out);
    void debug_fbank(int n_frames, const int16_t *pcm_frames /*[n][padded]*/, float *out /*[n][nbins]*/);
    void read_ring(int slot, int row, int n_rows, float *out);
    void read_greedy_state(int slot, GreedyState *out);

    // ---- profiling: when enabled every launch of the named classes is bracketed by hipEvents
    void set_profiling(bool on);
    // Gates clock (bench.py's roofline): while it is on, the feed-wavefront launch plans are built with a stamp slot per gates launch
    // (GemmArgs::stamp): the kernels time themselves under graph replay.  Switching it off reads the slots back.
    void set_gates_clock(bool on);
    void gates_clock(double *ms, long *launches, long *rows) const { *ms = gclk_ms_; *launches = gclk_launches_; *rows = gclk_rows_; }
    // ... and split by the number of problems sharing the launch (index min(n, 4) - 1): the three gates kernels of the trace
    // (one problem: gemm_f32_kernel, two: gemm_f32_zkernel, three at 256 sessions: gemm_f32_zkernel_walk) map onto it one to one
    void gates_clock_by_n(double *ms4, long *launches4) const { for (int i = 0; i < 4; ++i) { ms4[i] = gclk_ms_n_[i]; launches4[i] = gclk_launches_n_[i]; } }
    KernelTiming timing(int cls) const { return timing_[cls]; }
    void reset_timing();
    enum { T_GATES = 0, T_GEMM_OTHER = 1, T_ROW = 2, T_CONV = 3, T_FBANK = 4, T_DEC = 5, T_COUNT = 6 };
    long kernels_per_step() const { return kernels_per_step_.load(std::memory_order_relaxed); }    // launches of the last eagerly issued chunk chain

private:
    void upload_tables(const FbankHostTables &ft);
    void zero_slots(int n);
    void run_encoder_rows(int n, const int *d_slots, const int *d_tails, const float *x_direct);
    void lm_stage_embed(int m, int t0, int t1, hipStream_t st, bool own_ws = false);
    void lm_stage_layer(int l, int m, int t0, int t1, hipStream_t st);
    void lm_stage_proj(int m, int t0, int t1, hipStream_t st, float *ws = nullptr);
    void lm_resid_ssq(const float *a, int K, size_t w_off, int kz, const float *bias, const float *resid, size_t r0, int rows, hipStream_t st, float *ws = nullptr);
    GemmArgs lm_args_xpart(int l, int m, int t0, int t1) const;
    GemmArgs lm_args_gates(int l, int m, int t) const;
    GemmArgs sw_args_gates(int l, int m, int t) const;
    GemmArgs lm_args_whr(int l, int m, int t) const;
    GemmArgs lm_args_ff1(int l, int m, int t0, int t1) const;
    GemmArgs lm_args_ff2(int l, int m, int t0, int t1) const;
    // where one chunk's search reads its inputs and writes its records
    struct GreedyIo {
        int gen = 1;                          // generation of the chunk for the round flags / active marks (unique until they are cleared)
        const int *now = nullptr;             // [n] session times
        const float *eout = nullptr;          // null: the sessions' slot rows of eout_; else rows eout_rows[i] (or i) of this matrix
        const int *eout_rows = nullptr;
        const int *rec_off = nullptr; int rec_slot0 = 0;
        float *dump = nullptr;                // [3][n][vocab] or null
    };
    void run_greedy_rounds(int n, bool dump_logits, int chunk = 0, const float *eout_rows = nullptr);
    void run_greedy_rounds(int n, const GreedyIo &io);
    void run_lm_chain(int m, int T, bool dump_logits);
    void run_lm_wavefront(int m, int T, bool dump_logits);
    struct SwPlan {                          // argument blocks + launch list of run_sw_chain for one (m, T), and its captured graph
        struct Batch { size_t off; int n, macro, kind; size_t roff; int rn; size_t pf_off = 0; int pf_n = 0; };      // rn > 0: the GEMMs write partial planes, rn row problems finish them
        std::vector<GemmArgs> host; GemmArgs *dev = nullptr; std::vector<Batch> batches; hipGraphExec_t graph = nullptr; int uses = 0;
        hipGraphExec_t g3[3] = {nullptr, nullptr, nullptr};                      // split feed: front end / layers / search, one graph per stream
        std::vector<RowArgs> rhost; RowArgs *rdev = nullptr;
        std::vector<PrefetchItem> pf_host; PrefetchItem *pf_dev = nullptr;      // weight regions of every batch (launch_prefetch of the batch AFTER the one that runs)
        std::vector<std::pair<int, long>> stamp_slots; std::vector<int> stamp_n;  // gates clock: (slot, rows) and problem count of every gates launch of a plan built while it was on
    };
    SwPlan &sw_plan(int m, int T);
    void run_sw_chain(int m, int T, bool dump_logits, const SwPlan &p, int part, hipStream_t st);
    void run_sw_layers_chains(int m, int T);
    std::vector<hipStream_t> chain_streams_; std::vector<hipEvent_t> chain_ev_;
    struct StreamTrace { hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; int m = 0, T = 0; bool used = false; };
    std::deque<StreamTrace> trace_; hipEvent_t trace_base_ = nullptr;
    StreamTrace *trace_slot();
    void dump_stream_trace();
    void select_parity(int p);
    void join(hipStream_t waiter, hipStream_t src);
    void general_prologue();
    int next_step_index() { ++flight_steps_; return (int)(step_seq_++ & (uint64_t)(2 * step_cap_ - 1)); }
    void launch_rowepi(GemmArgs fused_form, size_t ws_row0, hipStream_t st);
    bool gates_tile_rows(long rows) const;
    bool ff1_tile_rows(long rows) const;
    void run_decproj(int n, const int *d_slots, const int *row_mask, const int *run_flag, int run_gen, float *out = nullptr);
    void build_dec_table();
    void run_chain(int m, bool dump_logits);     // advance + encoder + greedy rounds with arguments that depend on m only
    void timed_begin(int cls);
    void timed_end(int cls);
    void collect_timing();
    DecEmbedParams dec_params() const;

    EngineConfig cfg_;
    PackedLayout L_;
    ModelParams P_;
    hipStream_t stream_ = nullptr;
    float *w_ = nullptr;                       // packed weights
    uint16_t *wh_ = nullptr;                   // fp16 copies of the GEMM weight sections (same offsets, in elements), precision 1 only
    void lin(GemmArgs &g, size_t off) const { if (wh_) { g.wp = wh_ + off; g.wt = 1; } else { g.wp = w_ + off; g.wt = 0; } }
    // fp16 tile path (BASELINE configs[4]; kernels_gemm_tile.hip WT = 1): the four GEMMs of a layer read binary16 activations
    // (written by the producing epilogues) and weights re-packed for v_mfma_f32_16x16x32_f16; every batch size takes it
    bool f16_tile_ = false;
    uint16_t *wx_ = nullptr;                   // x32-order binary16 copies of the layer GEMM weights (same offsets as w_)
    uint16_t *y16_ = nullptr, *xb16_ = nullptr, *u16_ = nullptr, *ff16_ = nullptr;   // [max_batch][d_model | d_model | hidden | ffn]
    uint16_t *h16_ = nullptr;                  // [L][slots][d_model] binary16 copy of h
    int kzx_hr_ = 1, kzx_ff2_ = 1;             // K slabs of the projection / FFN-down GEMMs on 32-k blocks
    void lin16(GemmArgs &g, size_t off) const { g.wp = wx_ + off; g.wt = 1; g.tile_ok = 2; }
    int kz_hr() const { return f16_tile_ ? kzx_hr_ : kz_hr_; }
    int kz_ff2() const { return f16_tile_ ? kzx_ff2_ : kz_ff2_; }
    float *h_ = nullptr, *c_ = nullptr, *ring_ = nullptr, *eout_ = nullptr, *dout_ = nullptr;
    GreedyState *gstate_ = nullptr;            // [slots]
    float *dec_table_ = nullptr;               // [vocab * vocab][joiner] decoder output of every context, or null (build_dec_table)
    uint8_t *cls_ = nullptr;                   // [vocab] token classes
    int ring_frames_ = 0;
    // work buffers
    float *xin_ = nullptr, *a3_ = nullptr, *y_ = nullptr, *ssq_ = nullptr, *xb_ = nullptr, *u_ = nullptr, *ff_ = nullptr, *ws_ = nullptr, *ws_g_ = nullptr, *de_ = nullptr;
    float *logits_ = nullptr;                  // [3][max_batch][vocab] (traced steps, debug_joiner)
    float *p_lm_ = nullptr, *eout_lm_ = nullptr;   // layer-major: input half of the gates [rows][4 hidden], encoder outputs [rows][joiner] (allocated on first use)
    // step bookkeeping: pinned host rings (read by the advance kernel) + device mirrors
    int *ring_h_ = nullptr; size_t ring_cap_ = 0, ring_pos_ = 0;      // index blocks (capacities are per flight parity)
    int *step_off_h_ = nullptr, *rec_off_h_ = nullptr; int step_cap_ = 0;
    StepRecord *rec_d_ = nullptr, *rec_h_ = nullptr; size_t rec_cap_ = 0, rec_pos_ = 0;
    int flight_parity_ = 0, next_parity_ = 0; size_t ring_base_ = 0, rec_base_ = 0; int flight_steps_ = 0;
    uint64_t step_seq_ = 0;                    // steps enqueued since the engine started = the device's step counter (advance_kernel)
    hipEvent_t flight_done_[2] = {nullptr, nullptr};
    bool flight_open_[2] = {false, false};     // flight_done_[p] has been recorded and not yet waited for by begin_flight (wait_flight leaves it set: waiting twice is free)
    // streams (engine.cc "streams"): front end / search beside the layer chain, the per-parity buffers that make it safe
    hipStream_t f_stream_ = nullptr, s_stream_ = nullptr, search_stream_ = nullptr;
    hipStream_t pf_stream_ = nullptr; bool prefetch_ = false; hipEvent_t pf_ev_[8] = {}; unsigned pf_pos_ = 0;      // weight prefetch beside the layer launches (kernels.h launch_prefetch)
    std::vector<hipEvent_t> join_ev_; size_t join_pos_ = 0;
    bool f_unseen_by_m_ = false, s_unseen_by_m_ = false, m_unseen_by_f_ = false, m_unseen_by_s_ = false, flight_tail_s_ = false;
    bool overlap_hint_ = false;
    int split_streams_ = 2;                    // APRIL_SPLIT_STREAMS: 0 = one stream, 1 = search on S, 2 = search on S + front end on F
    float *y_buf_[2] = {nullptr, nullptr}, *ssq_buf_[2] = {nullptr, nullptr}, *eout_lm_buf_[2] = {nullptr, nullptr}, *ws_fe_ = nullptr, *ws_sr_ = nullptr;
    uint16_t *y16_buf_[2] = {nullptr, nullptr};
    int *step_buf_[2] = {nullptr, nullptr}, *flags_buf_[2] = {nullptr, nullptr}, *rec_off_buf_[2] = {nullptr, nullptr};
    std::map<std::pair<int, int>, int> sw_uses_;
    int *counter_d_ = nullptr, *step_d_ = nullptr, *active_d_ = nullptr, *dirty_d_ = nullptr, *rec_off_d_ = nullptr, *flags_d_ = nullptr;
    int *dec_slots_d_ = nullptr;
    float *logits_h_ = nullptr;
    // fbank staging is double-buffered so the next call can fill one pair while the previous copy is in flight
    FbankFrameDesc *hs_desc_[2] = {nullptr, nullptr}, *ds_desc_[2] = {nullptr, nullptr}; int desc_cap_ = 0;
    int16_t *hs_pcm_[2] = {nullptr, nullptr}, *ds_pcm_[2] = {nullptr, nullptr}; size_t pcm_cap_ = 0;
    int fb_flip_ = 0;
    hipEvent_t fb_done_[2] = {nullptr, nullptr};
    std::vector<size_t> part_off_;             // staging offsets of the PCM windows of one fbank call
    // fbank tables on device
    FbankTables ft_;
    std::vector<void *> table_allocs_;
    float pad_value_ = 0;
    // slots
    std::vector<int> free_, zero_pending_;     // free slots (reset), freed slots waiting for their reset launch
    void zero_pending_slots();
    std::atomic<int> live_{0};                  // written under slot_mu_, read without it (live_slots)
    std::mutex slot_mu_;
    std::mutex capture_mu_;                    // held while stream_ is being captured into a graph, and by other threads' enqueues
    // gemm split factors (fixed per shape => batch-invariant numerics)
    int kz_embed_ = 1, kz_hr_ = 1, kz_ff2_ = 1, kz_proj_ = 1, kz_out_ = 1;
    int ws_mstride_ = 0;
    float *conv_wt_ = nullptr;      // transposed weights of the first two convolutions: [9][ch0] then [ch0 * 9][ch1] (finish_weights)
    // workspace of the K-cut stream kernels at <= 16 rows (kernels.h GemmArgs::ks_ws / ks_cnt): per layer, projection and FFN down apart
    float *ks_ws_ = nullptr; unsigned *ks_cnt_ = nullptr; size_t ks_ws_stride_ = 0, ks_cnt_stride_ = 0;
    void attach_ksplit(GemmArgs &g, int l, int which) const;
    // chunk-step launch chains captured per batch size
    bool use_graphs_ = true;
    std::map<int, hipGraphExec_t> step_graphs_;
    std::map<int, int> step_seen_;             // batch size -> times seen (a chain is captured at its second use)
    std::map<std::pair<int, int>, hipGraphExec_t> lm_graphs_;      // (m, T)
    // wavefront form of the layer-major step: its stream, events, per-launch argument blocks (pinned + device), the
    // block-independent search graphs (m, block length) and their bookkeeping words
    std::map<std::pair<int, int>, hipGraphExec_t> lm_search_graphs_;
    std::map<std::pair<int, int>, SwPlan> sw_plans_;
    hipStream_t lm_stream_ = nullptr;
    std::vector<hipEvent_t> lm_events_;
    GemmArgs *zargs_h_ = nullptr, *zargs_d_ = nullptr; size_t zargs_region_ = 0, zargs_pos_ = 0;     // three regions, round robin
    hipEvent_t zargs_done_[3] = {nullptr, nullptr, nullptr}; bool zargs_busy_[3] = {false, false, false}; int zargs_next_ = 0;
    int *lm_now_d_ = nullptr, *lm_rows_d_ = nullptr, *lm_rec_off_d_ = nullptr;
    std::atomic<long> kernels_per_step_{0};    // written by the stepping thread, read by aprilx_model_stats on any thread
    long launch_count_ = 0;
    // profiling
    bool profiling_ = false;
    bool gclk_ = false; unsigned long long *gclk_slots_ = nullptr; int gclk_used_ = 0;      // gates clock (set_gates_clock): 8-word device slots
    static constexpr int GCLK_SLOTS = 2048;       // launch sites x 1152 B (STAMP_WORDS, device_utils.h)
    double gclk_ms_ = 0; long gclk_launches_ = 0, gclk_rows_ = 0; double gclk_ms_n_[4] = {0, 0, 0, 0}; long gclk_launches_n_[4] = {0, 0, 0, 0};
    struct Ev { hipEvent_t a, b; int cls; };
    std::vector<Ev> ev_pool_; size_t ev_used_ = 0;
    KernelTiming timing_[T_COUNT];
};

}  // namespace aprilx
