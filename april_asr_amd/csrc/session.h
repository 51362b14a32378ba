// Host-side session runtime for the GPU engine.
//
//   Greedy      the per-session transducer search + partial/final/silence state machine
//               (reference src/april_session.c:199-429), driven by the 16-byte per-round records
//               the device writes (arg-max, its logit, the blank logit) instead of 500-float logit
//               rows.  The decisions that steer the NEXT network call (blank or not, context push,
//               silence reset) are also taken on the device (kernels_misc.hip decide_kernel); the host
//               replays them from the same numbers when it builds the callbacks.
//   FrameBook   bookkeeping twin of the reference's OnlineFBank ring (src/fbank.c:98-127,
//               174-349): which frames exist, where they live in the HBM ring, flush padding.
//               The samples themselves only pass through (PCM16 FIFO -> pinned staging).
//   Scheduler   one stepping thread per GPU: gathers every session that has work, cuts new
//               frames (one fbank launch for all sessions), and advances all sessions with a
//               ready chunk in lock-step: one batched encoder pass, then three masked
//               joiner/decision/decoder rounds (reference src/april_session.c:431-476, batched),
//               chunk after chunk without waiting for the GPU; ONE wait per flight, then the
//               callbacks are replayed from the records.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/april_api.h"
#include "engine.h"
#include "model_loader.h"

namespace aprilx {

enum TokClass : uint8_t {
    TK_WORD_START = 1,      // text[0] == ' '
    TK_SENT_END = 2,        // single char . ! ?
    TK_COMMA = 4,           // single char ,
    TK_DOT = 8,             // text[0] == '.'
    TK_DIGIT_START = 16     // text[0] in '0'..'9'
};
std::vector<uint8_t> classify_tokens(const ModelParams &p);

struct JointResult { int32_t idx; float max_val; float blank_val; };     // what one joiner round hands to the search

struct Event {
    int type;                         // AprilResultType
    std::vector<AprilToken> tokens;
};

class Greedy {
public:
    static constexpr int kMaxActive = 72;     // reference src/april_session.h:30
    void init(const ModelParams *p, const std::vector<uint8_t> *cls);
    // consume one joiner result; returns true when the round resolved to blank (chunk done)
    bool on_joint(const JointResult &r, float early_emit, size_t now_ms, std::vector<Event> &out);
    // end of flush: FINAL, clear context, SILENCE (reference src/april_session.c:561-563)
    void finish_flush(std::vector<Event> &out);
    void reset_context_to_blank();            // first use: context = [blank, blank]
    int ctx[2] = {0, 0};
    bool ctx_dirty = false;                   // decoder must be re-run for this session

private:
    void push_ctx(int tok);
    void clear_context();
    void finalize_all(std::vector<Event> &out);
    void finalize_before_word(const AprilToken &incoming, std::vector<Event> &out);
    void emit_silence(std::vector<Event> &out);
    bool emit_partial(const AprilToken *tok, int tok_id, bool force, std::vector<Event> &out);
    void call(int type, size_t count, std::vector<Event> &out);

    const ModelParams *P_ = nullptr;
    const std::vector<uint8_t> *cls_ = nullptr;
    AprilToken active_[kMaxActive];
    int active_id_[kMaxActive];
    size_t head_ = 0, last_call_head_ = 0;
    bool emitted_silence_ = true;
    size_t last_emit_ms_ = 0;
};

struct FrameBook {
    int shift = 0, padded = 0, seg_count = 0, seg_step = 0, ring_frames = 0;
    int head = 0, tail = 0;
    long avail = 0, avail_shadow = 0;
    uint64_t rows_written = 0;      // ring rows written since the session started (real frames + flush padding)
    std::vector<int16_t> fifo;      // samples not yet fully consumed by framing
    size_t fifo_pos = 0;            // start of the next frame inside the stream  fifo ++ ext  (may point into ext)
    // A caller that blocks until its feed is processed LENDS its buffer: the samples are staged for the GPU straight from
    // there (one copy: caller -> pinned staging); only the unconsumed tail (less than a frame, normally) moves into the
    // fifo when the tick ends (settle()).
    const int16_t *ext = nullptr; size_t ext_cnt = 0;
    size_t stream_end() const { return fifo.size() + ext_cnt; }
    bool chunk_ready() const { return avail >= seg_count; }
    bool can_cut() const { return stream_end() - fifo_pos >= (size_t)padded && avail + 1 <= ring_frames; }
    void absorb_ext();              // ext -> fifo (keeps stream positions valid)
    void settle();                  // end of tick: drop consumed samples, keep the tail, forget the lent buffer
    bool flush_allowed() const { return avail_shadow >= -(long)(seg_count * 3); }
    void compact();
};

class Scheduler;
struct Model;

struct Session {
    Model *model = nullptr;
    Scheduler *sched = nullptr;
    Engine *eng = nullptr;
    int slot = -1;
    AprilRecognitionResultHandler handler = nullptr;
    void *userdata = nullptr;
    bool sync_mode = true, realtime_flag = false;

    // ---- guarded by Scheduler::mu_
    std::vector<int16_t> inbox;               // queued PCM (appended by callers; capacity is reused across feeds)
    const short *borrow_ptr = nullptr;        // PCM lent by a caller that blocks until the work is done (sync feed, feed_many):
    size_t borrow_cnt = 0;                    //   copied once, by the stepping thread, outside the lock
    bool fed = false;                         // a feed arrived since the last collection (even an empty one)
    bool flush_requested = false;
    bool busy = false;                        // owned by the stepping thread right now (inflight > 0)
    int inflight = 0;                         // ticks of this session that have been collected and not completed yet (<= 2)
    bool closing = false;
    uint64_t submitted = 0, completed = 0;    // work tickets
    std::chrono::steady_clock::time_point oldest_submit;   // when the oldest work not yet collected by the stepping thread was handed over
    bool has_oldest = false;
    std::vector<Event> done_events;           // sync sessions: events waiting for the caller thread

    // ---- owned by the stepping thread while busy
    FrameBook fb;
    Greedy greedy;
    bool dout_ready = false;
    bool compact_pending = false;
    struct Replay { int step; int row; int rows; uint32_t now_ms; int kind; int chunk; };   // kind 0: chunk `chunk` of step (3 rounds of records), 1: end of flush
    std::vector<Replay> replay;               // what the open flights did for this session, in order (a flight consumes its own items from the front)
    std::atomic<double> speed_needed{1.0};    // reference src/april_session.c:79,456-462 (EMA of processing time / audio time x 1.1);
                                              // written by the stepping thread, read by aas_realtime_get_speedup from any thread
    bool was_flushed = false;
    int flush_phase = 0;                      // 0 none, 1 pad-drain, 2 zeros, 3 pad-drain, 4 finish
    size_t now_ms = 0;
    uint64_t chunks = 0;
    std::vector<Event> events;                // produced during the current tick
    // tracing (tests): every joiner call appends `vocab` floats
    float *trace_buf = nullptr; size_t trace_cap = 0; size_t *trace_used = nullptr;
};

struct SchedStats {
    uint64_t ticks = 0, steps = 0, chunks = 0, rounds = 0, frames = 0, max_batch_seen = 0, flights = 0, replay_mismatch = 0, lm_steps = 0, lm_chunks = 0, wave_steps = 0, wave_chunks = 0;
    // host wall time of the stepping thread by phase (ms): 0 collect, 1 cut frames (host), 2 fbank call, 3 step enqueue,
    // 4 end of flight (wait for the GPU), 5 replay (decisions + events), 6 decoder refresh enqueue, 7 deliver/complete
    double host_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void add(const SchedStats &o) {
        ticks += o.ticks; steps += o.steps; chunks += o.chunks; rounds += o.rounds; frames += o.frames; flights += o.flights; replay_mismatch += o.replay_mismatch;
        lm_steps += o.lm_steps; lm_chunks += o.lm_chunks; wave_steps += o.wave_steps; wave_chunks += o.wave_chunks;
        if (o.max_batch_seen > max_batch_seen) max_batch_seen = o.max_batch_seen;
        for (int i = 0; i < 8; ++i) host_ms[i] += o.host_ms[i];
    }
};

class Scheduler {
public:
    Scheduler(Model *m, Engine *e);
    ~Scheduler();
    void attach(Session *s);
    bool detach(Session *s);                       // waits until the session is idle; false = refused (called from the session's own handler)
    // queue work for n sessions at once and (for sync sessions / wait=true) block until it is done
    // `borrow`: the caller keeps the PCM buffers alive and unchanged until the sessions are idle again (it blocks in this
    // call, or drains afterwards), so they are read in place by the stepping thread instead of being copied under the lock
    void submit(int n, Session *const *ss, const short *const *pcm, const size_t *counts, bool flush, bool wait, bool borrow = false);
    void deliver_sync_events(Session *s);          // caller-thread delivery for sync sessions
    void wait_idle(Session *s);                    // everything queued so far has been processed
    void wait_idle_many(Session *const *ss, int n);
    // until every listed session has at most `max_open` feeds that were submitted and not completed yet (pipelined group feeds)
    void wait_backlog(Session *const *ss, int n, uint64_t max_open);
    SchedStats stats();
    // hand-over -> delivery latencies (ms) of the last completed ticks, oldest first: from the submit() that queued the oldest work a
    // flight served to the moment its results were delivered (asynchronous handlers have run; synchronous callers have been released)
    size_t latencies(double *out, size_t cap, bool reset);
    Engine *engine() { return eng_; }
    bool on_loop_thread() const { return std::this_thread::get_id() == loop_tid_; }

private:
    // one launched flight: the sessions it serves, their tickets, and how many replay items / chunks it added per session
    struct Flight {
        int parity = -1;                                     // the engine's flight id (Engine::close_flight)
        bool final = true;                                   // false: the rings filled up, the same tick continues in the next flight
        std::vector<Session *> work;
        std::vector<uint64_t> taken;
        std::vector<uint32_t> mark;
        std::vector<uint64_t> chunks0, chunks1;
        std::chrono::steady_clock::time_point t0;
        std::chrono::steady_clock::time_point t_sub;          // hand-over of the oldest work in the flight (submit() of any of its sessions)
        bool has_sub = false;
    };
    void loop();
    bool collect(std::vector<Session *> &work, std::vector<uint64_t> &taken, bool block, uint64_t &work_seen);
    Flight launch_flight(const std::vector<Session *> &work, const std::vector<uint64_t> &taken);
    void complete_flight(Flight &f);
    void cut_frames(std::vector<Session *> &work, bool &progressed);
    bool step_chunks(std::vector<Session *> &ready);     // false: the flight's rings are full, (some) work is left for the next flight
    bool step_layer_major(std::vector<Session *> &group, int T, int mode = 0);
    void replay(Flight &f);
    int split_sticky_ = 0;
    int pipeline_depth_ = 2;                             // APRIL_PIPELINE: 2 = launch the next flight before completing the current one, 1 = one flight at a time

    Model *model_;
    Engine *eng_;
    HostPool pool_;                                // helpers for the per-session host copies (APRIL_HOST_THREADS, default 3)
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    // spin-then-block on both sides of the hand-over (a condition-variable wake-up costs 30..60 us each way, which is 5 %
    // of a 2 ms step): submit() bumps work_seq_, the end of a tick bumps done_seq_
    std::atomic<uint64_t> work_seq_{0}, done_seq_{0};
    int spin_step_us_ = 1000, spin_wait_us_ = 3000;   // APRIL_SPIN_STEP_US / APRIL_SPIN_WAIT_US
    std::chrono::steady_clock::time_point prev_done_;   // when the previous flight completed (stepping thread only): start of the next flight's own span
    bool have_prev_done_ = false;
    int wave_min_chunks_ = 2, wave_max_chunks_ = 7;  // APRIL_WAVE_MIN_CHUNKS (0 = chunk steps one by one) / APRIL_WAVE_MAX_CHUNKS: chunk steps of one feed as a wavefront
    int lm_min_chunks_ = 8;                          // APRIL_LM_MIN_CHUNKS: sessions with at least this many chunks waiting take the layer-major path (0 = never)
    void spin_for_done(uint64_t seen);
    std::vector<Session *> sessions_;
    bool stop_ = false;
    std::thread thread_;
    SchedStats stats_;                             // guarded by mu_
    SchedStats tick_;                              // the stepping thread's own; merged into stats_ under mu_ at the end of a tick
    std::vector<float> lat_ms_;                    // guarded by mu_: ring of the last kLatRing hand-over -> delivery latencies
    uint64_t lat_n_ = 0;
    static constexpr size_t kLatRing = 8192;
    std::chrono::steady_clock::time_point collect_t_sub_; bool collect_has_sub_ = false;   // stepping thread: oldest hand-over among the work of the last collect()
    std::thread::id loop_tid_;
    // scratch reused across ticks
    std::vector<FbankFrameDesc> desc_;
    std::vector<std::pair<const int16_t *, size_t>> pcm_parts_;   // windows to stage, in order
    std::vector<int> slots_, tails_, now_;
    std::vector<float> logit_stage_;
};

struct LoadInfo { double broadcast_ms = 0, comm_init_ms = 0; size_t broadcast_bytes = 0; int ranks = 1, used_rccl = 0; };

struct Model {
    LoadInfo load;                            // how the weights reached the engines
    HostModel host;                           // params, tokens, names (weights freed after upload)
    PackedLayout layout;
    FbankHostTables ftab;
    std::vector<uint8_t> tok_class;
    std::vector<Engine *> engines;
    std::vector<Scheduler *> scheds;
    std::vector<float> host_blob;             // only for host-only models (no engine): packed weights
    std::mutex mu;
    ~Model();
};

}  // namespace aprilx
