// See session.h.
#include "session.h"
#include <algorithm>
#include <tuple>
#include <cstring>
#include <chrono>
#include "common.h"

namespace aprilx {

namespace {
struct Lap {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double operator()() { auto n = std::chrono::steady_clock::now(); double ms = std::chrono::duration<double, std::milli>(n - t).count(); t = n; return ms; }
};
}  // namespace

// ---------------------------------------------------------------- token classes
std::vector<uint8_t> classify_tokens(const ModelParams &p)
{
    std::vector<uint8_t> c((size_t)p.token_count, 0);
    for (int i = 0; i < p.token_count; ++i) {
        const char *t = p.token((size_t)i);
        uint8_t f = 0;
        if (t[0] == ' ') f |= TK_WORD_START;                                   // april_session.c:338
        const bool single = t[0] != 0 && t[1] == 0;
        if (single && (t[0] == '.' || t[0] == '!' || t[0] == '?')) f |= TK_SENT_END;   // :341
        if (single && t[0] == ',') f |= TK_COMMA;                              // :342
        if (t[0] == '.') f |= TK_DOT;
        if (t[0] >= '0' && t[0] <= '9') f |= TK_DIGIT_START;                   // :347
        c[(size_t)i] = f;
    }
    return c;
}

// ---------------------------------------------------------------- greedy search + result state machine
void Greedy::init(const ModelParams *p, const std::vector<uint8_t> *cls)
{
    P_ = p; cls_ = cls;
    memset(active_, 0, sizeof active_);
    for (int &i : active_id_) i = -1;
    head_ = last_call_head_ = 0;
    emitted_silence_ = true;                       // april_session.c:64
    last_emit_ms_ = 0;
    ctx[0] = ctx[1] = 0;
    ctx_dirty = false;
}

void Greedy::call(int type, size_t count, std::vector<Event> &out)
{
    Event e; e.type = type;
    e.tokens.assign(active_, active_ + count);
    out.push_back(std::move(e));
}

void Greedy::push_ctx(int tok) { ctx[0] = ctx[1]; ctx[1] = tok; ctx_dirty = true; }   // :181-196 (context_size == 2)

void Greedy::reset_context_to_blank() { push_ctx(P_->blank_id); push_ctx(P_->blank_id); }   // :432-438

void Greedy::clear_context()
{
    if (ctx[0] == P_->blank_id) return;            // :297 (tests element 0, quirk kept)
    push_ctx(P_->blank_id); push_ctx(P_->blank_id);
}

void Greedy::finalize_all(std::vector<Event> &out)
{
    if (head_ == 0) return;                         // :199-211
    call(APRIL_RESULT_RECOGNITION_FINAL, head_, out);
    last_call_head_ = head_;
    head_ = 0;
}

void Greedy::finalize_before_word(const AprilToken &incoming, std::vector<Event> &out)
{
    if (head_ == 0) return;                         // :213-255
    if (incoming.flags & APRIL_TOKEN_FLAG_WORD_BOUNDARY_BIT) { finalize_all(out); return; }
    size_t start = kMaxActive;
    for (size_t i = head_ - 1; i > 2; --i)
        if (active_[i].flags & APRIL_TOKEN_FLAG_WORD_BOUNDARY_BIT) { start = i; break; }
    if (start == (size_t)kMaxActive) { finalize_all(out); return; }
    call(APRIL_RESULT_RECOGNITION_FINAL, start, out);
    memmove(active_, active_ + start, sizeof(AprilToken) * (head_ - start));
    memmove(active_id_, active_id_ + start, sizeof(int) * (head_ - start));
    head_ -= start;
}

void Greedy::emit_silence(std::vector<Event> &out)
{
    if (emitted_silence_) return;                   // :257-268
    emitted_silence_ = true;
    Event e; e.type = APRIL_RESULT_SILENCE;
    out.push_back(std::move(e));
}

bool Greedy::emit_partial(const AprilToken *tok, int tok_id, bool force, std::vector<Event> &out)
{
    if (tok) {                                      // :270-294
        if (!force && last_call_head_ == head_ + 1 && active_id_[head_] == tok_id) return false;
        active_[head_] = *tok;
        active_id_[head_] = tok_id;
        ++head_;
    } else if (!force && last_call_head_ == head_) {
        return false;
    }
    call(APRIL_RESULT_RECOGNITION_PARTIAL, head_, out);
    last_call_head_ = head_;
    return true;
}

bool Greedy::on_joint(const JointResult &r, float early_emit, size_t now_ms, std::vector<Event> &out)
{
    const int blank = P_->blank_id;
    int best = r.idx;
    float best_v = r.max_val;
    if (best < 0) { best = blank == 0 ? 1 : 0; best_v = -9999999999.0f; }   // no logit beat the initial value (NaNs)
    const float blank_v = r.blank_val;

    const bool cleared = ctx[1] == blank;           // :322
    const bool same = ctx[1] == best;               // :326
    if (same) early_emit = 0.0f;
    bool is_blank = (blank_v - early_emit) > best_v;    // :329-330

    const uint8_t tc = (*cls_)[(size_t)best];
    AprilToken tok;
    tok.token = P_->token((size_t)best);
    tok.logprob = best_v;
    tok.flags = (AprilTokenFlagBits)0;
    tok.time_ms = now_ms;
    tok.reserved = nullptr;
    int flags = 0;
    if (tc & TK_WORD_START) flags |= APRIL_TOKEN_FLAG_WORD_BOUNDARY_BIT;
    bool eos = (tc & TK_SENT_END) != 0;
    bool punct = eos || (tc & TK_COMMA);
    if (punct && head_ > 0) {                       // :345-351 "10.0" is not a sentence end
        const uint8_t lc = (*cls_)[(size_t)active_id_[head_ - 1]];
        if ((lc & TK_DIGIT_START) && (tc & TK_DOT)) { eos = false; punct = false; }
    }
    if (eos) flags |= APRIL_TOKEN_FLAG_SENTENCE_END_BIT;
    tok.flags = (AprilTokenFlagBits)flags;
    if (!cleared && punct && !same && best_v > (blank_v - 3.5f)) is_blank = false;   // :356-358

    if (!is_blank) {                                // :361-400
        last_emit_ms_ = now_ms;
        push_ctx(best);
        bool fin = head_ >= (size_t)(kMaxActive - 1);
        if (head_ > 0 && (flags & APRIL_TOKEN_FLAG_WORD_BOUNDARY_BIT)) {
            AprilToken &prev = active_[head_ - 1];
            const bool prev_eos = ((*cls_)[(size_t)active_id_[head_ - 1]] & TK_SENT_END) != 0;
            if (prev_eos && !(prev.flags & APRIL_TOKEN_FLAG_SENTENCE_END_BIT))
                prev.flags = (AprilTokenFlagBits)(prev.flags | APRIL_TOKEN_FLAG_SENTENCE_END_BIT);
            if (prev_eos) fin = true;
        }
        if (fin) finalize_before_word(tok, out);
        if (head_ >= (size_t)(kMaxActive - 1)) { LOGE("No room left even after finalizing previous words"); head_ = 0; }
        emit_partial(&tok, best, true, out);
        emitted_silence_ = false;
    } else {                                        // :401-426
        const size_t gap = now_ms - last_emit_ms_;
        const float decayed = best_v - (float)gap / 3000.0f;
        const bool confident = !same && decayed > (blank_v - 4.0f);
        if (gap >= 2200) {
            finalize_all(out);
            clear_context();
            emit_silence(out);
        } else if (confident) {
            tok.logprob -= 8.0f;
            if (emit_partial(&tok, best, false, out)) --head_;
        } else {
            emit_partial(nullptr, -1, false, out);
        }
    }
    return is_blank;
}

void Greedy::finish_flush(std::vector<Event> &out)
{
    finalize_all(out);
    clear_context();
    emit_silence(out);
}

void FrameBook::compact()
{
    if (ext) return;                                   // positions may point into the lent buffer: settle() does it
    if (fifo_pos > 0 && (fifo_pos >= 8192 || fifo_pos == fifo.size())) {
        fifo.erase(fifo.begin(), fifo.begin() + (long)fifo_pos);
        fifo_pos = 0;
    }
}

void FrameBook::absorb_ext()
{
    if (!ext) return;
    fifo.insert(fifo.end(), ext, ext + ext_cnt);
    ext = nullptr; ext_cnt = 0;
}

void FrameBook::settle()
{
    if (ext) {
        if (fifo_pos >= fifo.size()) {                 // everything older is consumed: the fifo becomes the lent buffer's tail
            const size_t skip = fifo_pos - fifo.size();
            fifo.assign(ext + skip, ext + ext_cnt);
            fifo_pos = 0;
            ext = nullptr; ext_cnt = 0;
        } else {
            absorb_ext();
        }
    }
    compact();
}

// ---------------------------------------------------------------- model
Model::~Model()
{
    for (auto *s : scheds) delete s;
    for (auto *e : engines) delete e;
}

// ---------------------------------------------------------------- scheduler
static constexpr size_t kAsyncRingSamples = 48000;     // reference src/audio_provider.c:31 (3 s at 16 kHz)

static int host_helpers()
{
    const char *v = getenv("APRIL_HOST_THREADS");
    // default: one helper per 16 hardware threads, 3..8 (measured at 2048 sessions: 8 helpers 10.07 ms per step, 3 helpers 10.59)
    const int hw = (int)std::thread::hardware_concurrency();
    const int n = v && *v ? atoi(v) : std::max(3, std::min(8, hw / 16));
    return n < 0 ? 0 : (n > 32 ? 32 : n);
}

static int env_us(const char *name, int def)
{
    const char *v = getenv(name);
    const int n = v && *v ? atoi(v) : def;
    return n < 0 ? 0 : (n > 1000000 ? 1000000 : n);
}

Scheduler::Scheduler(Model *m, Engine *e) : model_(m), eng_(e), pool_(host_helpers())
{
    spin_step_us_ = env_us("APRIL_SPIN_STEP_US", 1000);     // (1 ms: a client that pauses for a barrier or a sync between two feeds finds the thread awake; 100 us until round 3)
    spin_wait_us_ = env_us("APRIL_SPIN_WAIT_US", 3000);
    lm_min_chunks_ = env_us("APRIL_LM_MIN_CHUNKS", 8);
    wave_min_chunks_ = env_us("APRIL_WAVE_MIN_CHUNKS", 2);
    wave_max_chunks_ = std::max(1, env_us("APRIL_WAVE_MAX_CHUNKS", 7));
    pipeline_depth_ = std::max(1, std::min(2, env_us("APRIL_PIPELINE", 2)));
    thread_ = std::thread([this] { loop(); });
}

static inline void cpu_relax() { __builtin_ia32_pause(); }

// caller side: poll the tick counter for a bounded time before sleeping on the condition variable
void Scheduler::spin_for_done(uint64_t seen)
{
    if (spin_wait_us_ <= 0) return;
    const auto t0 = std::chrono::steady_clock::now();
    while (done_seq_.load(std::memory_order_acquire) == seen) {
        for (int i = 0; i < 64; ++i) cpu_relax();
        if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_wait_us_) break;
    }
}

Scheduler::~Scheduler()
{
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_work_.notify_all();
    if (thread_.joinable()) thread_.join();
}

void Scheduler::attach(Session *s)
{
    std::lock_guard<std::mutex> g(mu_);
    sessions_.push_back(s);
}

bool Scheduler::detach(Session *s)
{
    std::unique_lock<std::mutex> lk(mu_);
    if (on_loop_thread() && s->busy) {
        // aas_free from inside this session's own asynchronous result handler: the stepping thread would wait for itself
        // (the reference joins its own thread there, src/proc_thread.c:101-116).  Refuse loudly instead of hanging.
        LOGE("aas_free called from inside the session's result handler: not supported, the session is left alive");
        return false;
    }
    s->closing = true;
    cv_done_.wait(lk, [&] { return !s->busy; });
    sessions_.erase(std::remove(sessions_.begin(), sessions_.end(), s), sessions_.end());
    s->inbox.clear();
    s->borrow_ptr = nullptr; s->borrow_cnt = 0;
    return true;
}

SchedStats Scheduler::stats() { std::lock_guard<std::mutex> g(mu_); return stats_; }

size_t Scheduler::latencies(double *out, size_t cap, bool reset)
{
    std::lock_guard<std::mutex> g(mu_);
    const size_t have = lat_ms_.size(), first = lat_n_ > kLatRing ? (size_t)(lat_n_ % kLatRing) : 0;
    const size_t n = out ? std::min(have, cap) : 0;
    for (size_t i = 0; i < n; ++i) out[i] = lat_ms_[(first + (have - n) + i) % have];      // the newest n, oldest first
    const size_t ret = out ? n : have;
    if (reset) { lat_ms_.clear(); lat_n_ = 0; }
    return ret;
}

void Scheduler::submit(int n, Session *const *ss, const short *const *pcm, const size_t *counts, bool flush, bool wait, bool borrow)
{
    std::vector<Session *> overflowed;
    std::vector<uint64_t> tickets((size_t)n, 0);
    uint64_t done_seen = 0;
    const auto t_sub = std::chrono::steady_clock::now();
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (int i = 0; i < n; ++i) {
            Session *s = ss[i];
            if (s->closing) continue;
            if (!s->has_oldest) { s->oldest_submit = t_sub; s->has_oldest = true; }
            if (flush) s->flush_requested = true;
            else {
                const size_t cnt = counts[i];
                // the reference's ring refuses a push that would make it hold MAX_AUDIO samples or more (src/audio_provider.c:61)
                if (!s->sync_mode && s->inbox.size() + s->borrow_cnt + cnt >= kAsyncRingSamples) { overflowed.push_back(s); continue; }   // april_session.c:482-492
                if (cnt) {
                    if (borrow && !s->borrow_cnt && s->inbox.empty()) { s->borrow_ptr = pcm[i]; s->borrow_cnt = cnt; }
                    else s->inbox.insert(s->inbox.end(), pcm[i], pcm[i] + cnt);
                }
                s->fed = true;
            }
            tickets[(size_t)i] = ++s->submitted;
        }
        // read while mu_ is still held: the tick that takes this work cannot have ended yet (it needs mu_ to collect it), so a
        // later change of done_seq_ is never missed and the spin below never waits on a counter that has already moved
        done_seen = done_seq_.load(std::memory_order_acquire);
    }
    work_seq_.fetch_add(1, std::memory_order_release);
    cv_work_.notify_one();
    for (Session *s : overflowed) s->handler(s->userdata, APRIL_RESULT_ERROR_CANT_KEEP_UP, 0, nullptr);
    if (!wait) return;
    spin_for_done(done_seen);
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] {
        for (int i = 0; i < n; ++i) if (tickets[(size_t)i] && ss[i]->completed < tickets[(size_t)i] && !ss[i]->closing) return false;
        return true;
    });
}

void Scheduler::wait_idle(Session *s)
{
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return s->closing || (s->completed >= s->submitted && !s->busy && !s->fed && !s->flush_requested); });
}

void Scheduler::wait_idle_many(Session *const *ss, int n)
{
    {   // called right after a submit: poll for the end of the tick that took the work before sleeping
        uint64_t seen;
        bool idle;
        {
            std::lock_guard<std::mutex> g(mu_);
            seen = done_seq_.load(std::memory_order_acquire);      // under the lock, as in submit()
            idle = true;
            for (int i = 0; i < n && idle; ++i) { const Session *s = ss[i]; idle = s->closing || (s->completed >= s->submitted && !s->busy && !s->fed && !s->flush_requested); }
        }
        if (!idle) spin_for_done(seen);
    }
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] {
        for (int i = 0; i < n; ++i) {
            const Session *s = ss[i];
            if (!(s->closing || (s->completed >= s->submitted && !s->busy && !s->fed && !s->flush_requested))) return false;
        }
        return true;
    });
}

void Scheduler::wait_backlog(Session *const *ss, int n, uint64_t max_open)
{
    auto ok = [&] {
        for (int i = 0; i < n; ++i) { const Session *s = ss[i]; if (!s->closing && s->submitted - s->completed > max_open) return false; }
        return true;
    };
    for (;;) {
        uint64_t seen;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (ok()) return;
            seen = done_seq_.load(std::memory_order_acquire);      // under the lock, as in submit()
        }
        spin_for_done(seen);
        if (done_seq_.load(std::memory_order_acquire) != seen) continue;
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return ok() || done_seq_.load(std::memory_order_acquire) != seen; });
    }
}

void Scheduler::deliver_sync_events(Session *s)
{
    std::vector<Event> ev;
    { std::lock_guard<std::mutex> g(mu_); ev.swap(s->done_events); }
    for (auto &e : ev) s->handler(s->userdata, (AprilResultType)e.type, e.tokens.size(), e.tokens.empty() ? nullptr : e.tokens.data());
}

// One stepping thread, TWO flights in the air.  A flight is launched (frames cut, chunk steps enqueued, record copy + event
// queued: launch_flight) without waiting for the GPU; it is completed (wait for its event, replay the records through the
// search state machine, deliver the callbacks, release the callers: complete_flight) AFTER the next flight has been launched
// whenever work for that next flight is already queued -- asynchronous sessions, pipelined group feeds
// (aprilx_feed_many_pipelined), or simply other clients' sessions.  The host part of flight k + 1 (framing, PCM staging, index
// blocks, launch) then runs under the GPU time of flight k instead of between two flights (measured at 256 sessions: 170 us of
// GPU idle time per 100 ms feed, profiles/r04a_b256_timeline.txt).  A caller that blocks on its own feed (aas_feed_pcm16 of a
// synchronous session, aprilx_feed_many) sees the same thing as before: its tickets complete when its flight does.
bool Scheduler::collect(std::vector<Session *> &work, std::vector<uint64_t> &taken, bool block, uint64_t &work_seen)
{
    work.clear(); taken.clear();
    if (block && spin_step_us_ > 0) {           // a feed usually follows the previous one within microseconds: poll before sleeping
        const auto t0 = std::chrono::steady_clock::now();
        while (work_seq_.load(std::memory_order_acquire) == work_seen) {
            for (int i = 0; i < 64; ++i) cpu_relax();
            if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_step_us_) break;
        }
    }
    Lap lap;
    std::unique_lock<std::mutex> lk(mu_);
    auto have = [&] {
        for (Session *s : sessions_) if (!s->closing && (s->fed || s->flush_requested)) return true;
        return false;
    };
    if (block) cv_work_.wait(lk, [&] { return stop_ || have(); });
    work_seen = work_seq_.load(std::memory_order_acquire);
    if (stop_) return false;
    lap();
    collect_has_sub_ = false;
    for (Session *s : sessions_) {
        if (s->closing || (!s->fed && !s->flush_requested)) continue;
        if (s->has_oldest) { if (!collect_has_sub_ || s->oldest_submit < collect_t_sub_) collect_t_sub_ = s->oldest_submit; collect_has_sub_ = true; s->has_oldest = false; }
        s->busy = true;
        s->inflight += 1;
        if (s->borrow_cnt) {
            if (s->inbox.empty() && !s->fb.ext) { s->fb.ext = s->borrow_ptr; s->fb.ext_cnt = s->borrow_cnt; }   // read in place during this tick
            else { s->fb.absorb_ext(); s->fb.fifo.insert(s->fb.fifo.end(), s->borrow_ptr, s->borrow_ptr + s->borrow_cnt); }   // something queued behind it: keep the order
            s->borrow_ptr = nullptr; s->borrow_cnt = 0;
        }
        if (!s->inbox.empty()) { s->fb.absorb_ext(); s->fb.fifo.insert(s->fb.fifo.end(), s->inbox.begin(), s->inbox.end()); s->inbox.clear(); }
        if (s->fed) s->was_flushed = false;                               // april_session.c:510
        s->fed = false;
        if (s->flush_requested) {                                           // :547-552
            s->flush_requested = false;
            if (!s->was_flushed && s->flush_phase == 0) { s->was_flushed = true; s->flush_phase = 1; }
        }
        work.push_back(s);
        taken.push_back(s->submitted);
    }
    // lent PCM: the caller is blocked until `completed` moves, so its buffer is read in place while the flight is launched (a
    // lent buffer is only accepted when nothing is queued in front of it, so the order of samples is kept)
    tick_.host_ms[0] += lap();
    return !work.empty();
}

// Everything of a tick that does not need the GPU's answers.  Returns the flight (parity -1: nothing reached the GPU).
// final == false: the flight's rings filled up; the same sessions continue in the next flight (loop() keeps `work`).
Scheduler::Flight Scheduler::launch_flight(const std::vector<Session *> &work_in, const std::vector<uint64_t> &taken)
{
    Flight f;
    f.work = work_in; f.taken = taken; f.t0 = std::chrono::steady_clock::now();
    f.t_sub = collect_t_sub_; f.has_sub = collect_has_sub_;      // (a follow-up flight of the same tick keeps the tick's hand-over time)
    std::vector<Session *> &work = f.work;
    f.mark.resize(work.size()); f.chunks0.resize(work.size());
    for (size_t i = 0; i < work.size(); ++i) { f.mark[i] = (uint32_t)work[i]->replay.size(); f.chunks0[i] = work[i]->chunks; }
    std::vector<Session *> ready;
    eng_->begin_flight();
    bool more = false, any = false;
    const uint64_t steps0 = tick_.steps;
    for (;;) {
        bool progressed = false;
        cut_frames(work, progressed);
        ready.clear();
        for (Session *s : work) if (s->fb.chunk_ready()) ready.push_back(s);
        if (!ready.empty()) {
            if (!step_chunks(ready)) { more = true; break; }          // rings full: land this flight, continue in the next
            progressed = true;
        }
        if (!progressed) break;
        any = true;
    }
    if (more && !any && tick_.steps == steps0) {
        // a step that does not fit an EMPTY flight will never fit: the engine's index / record rings must hold max_batch rows
        // (engine.cc sizes them so); without this the stepping thread would open flight after flight forever
        LOGE("scheduler: a chunk step does not fit an empty flight (index / record rings smaller than one step of max_batch rows)");
        abort();
    }
    Lap lap;
    f.parity = eng_->close_flight();
    f.final = !more;
    for (size_t i = 0; i < work.size(); ++i) f.mark[i] = (uint32_t)work[i]->replay.size() - f.mark[i];      // items this flight appended
    // lent buffers go back to their callers when the tick completes; they are not read after this point (the samples are in
    // pinned staging): keep what framing has not consumed yet
    if (f.final) pool_.run(work.size(), 64, [&](size_t i) { work[i]->fb.settle(); });
    f.chunks1.resize(work.size());
    for (size_t i = 0; i < work.size(); ++i) f.chunks1[i] = work[i]->chunks;
    tick_.host_ms[3] += lap();
    tick_.flights++;
    return f;
}

void Scheduler::complete_flight(Flight &f)
{
    Lap lap;
    eng_->wait_flight(f.parity);
    tick_.host_ms[4] += lap();
    replay(f);
    tick_.host_ms[5] += lap();
    std::vector<Session *> &work = f.work;
    {   // reference src/april_session.c:456-462: EMA of (processing time x 1.1) / audio time per chunk.  All sessions of a
        // flight are stepped together, so a chunk's processing time is the flight's wall time over the chunks the session advanced.
        // With two flights in the air a flight's wall time since its launch includes the time it queued behind the one before it:
        // what it cost is the span since that one completed (the EMA read up to 2 x too high under steady pipelined load)
        const auto now = std::chrono::steady_clock::now();
        const auto from = (have_prev_done_ && prev_done_ > f.t0) ? prev_done_ : f.t0;
        const double tick_ms = std::chrono::duration<double, std::milli>(now - from).count();
        prev_done_ = now; have_prev_done_ = true;
        const double stride_ms = (double)(model_->host.params.segment_step * model_->host.params.frame_shift_ms);
        for (size_t i = 0; i < work.size(); ++i) {
            Session *s = work[i];
            const uint64_t n = f.chunks1[i] - f.chunks0[i];
            if (!n) continue;
            double v = s->speed_needed.load(std::memory_order_relaxed);
            for (uint64_t k = 0; k < n; ++k) v = (v * 9.0 + (tick_ms / (double)n) * 1.1 / stride_ms) / 10.0;
            s->speed_needed.store(v, std::memory_order_relaxed);
        }
    }
    // async sessions: deliver on this (library) thread, outside the lock
    for (Session *s : work) if (!s->sync_mode) {
        for (auto &e : s->events) s->handler(s->userdata, (AprilResultType)e.type, e.tokens.size(), e.tokens.empty() ? nullptr : e.tokens.data());
        s->events.clear();
    }
    {
        std::lock_guard<std::mutex> g(mu_);
        for (size_t i = 0; i < work.size(); ++i) {
            Session *s = work[i];
            if (s->sync_mode) { for (auto &e : s->events) s->done_events.push_back(std::move(e)); s->events.clear(); }
            if (f.final) {
                if (f.taken[i] > s->completed) s->completed = f.taken[i];
                s->inflight -= 1;
                s->busy = s->inflight > 0;
            }
        }
        if (f.final) tick_.ticks++;
        if (f.final && f.has_sub) {
            const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - f.t_sub).count();
            if (lat_ms_.size() < kLatRing) lat_ms_.push_back(ms); else lat_ms_[(size_t)(lat_n_ % kLatRing)] = ms;
            ++lat_n_;
        }
        tick_.host_ms[7] += lap();
        stats_.add(tick_);
        tick_ = SchedStats();
    }
    done_seq_.fetch_add(1, std::memory_order_release);
    cv_done_.notify_all();
}

void Scheduler::loop()
{
    HIP_CHECK(hipSetDevice(eng_->device()));
    { std::lock_guard<std::mutex> g(mu_); loop_tid_ = std::this_thread::get_id(); }
    std::vector<Session *> work;
    std::vector<uint64_t> taken;
    uint64_t work_seen = 0;
    bool have_pending = false, have_next = false, cont = false;
    Flight pending, next;
    for (;;) {
        // ---- launch: the continuation of a tick whose flight filled the rings first, else whatever has been queued
        if (!have_next) {
            bool go = cont;
            if (!go) {
                go = collect(work, taken, /*block=*/!have_pending, work_seen);
                if (!go && !have_pending) { std::lock_guard<std::mutex> g(mu_); if (stop_) return; }
            }
            if (go) {
                // a flight launched behind one that is still in the air overlaps with it (and, with steady pipelined feeding, the
                // one after it will be launched behind this one): worth its three streams.  `split_sticky_` keeps the choice for
                // a few flights so that a flight that happened to find the GPU idle does not flip the launch path back and forth
                if (have_pending) split_sticky_ = 8; else if (split_sticky_ > 0) --split_sticky_;
                eng_->set_overlap_hint(pipeline_depth_ >= 2 && split_sticky_ > 0);
                next = launch_flight(work, taken);
                have_next = true;
                cont = !next.final;
            }
        }
        // ---- complete the older flight: at once when a younger one is already behind it on the stream (or nothing can be
        // overlapped: profiling runs account every launch to the flight that issued it), else as soon as the GPU is through
        // with it -- while watching for new work that could still be launched behind it
        if (have_pending) {
            bool done = have_next || pipeline_depth_ < 2 || eng_->profiling();
            if (!done) {
                const auto t0 = std::chrono::steady_clock::now();
                for (;;) {
                    if (eng_->flight_done(pending.parity)) { done = true; break; }
                    if (work_seq_.load(std::memory_order_acquire) != work_seen) break;                 // something was queued: try to launch it first
                    for (int i = 0; i < 32; ++i) cpu_relax();
                    if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_wait_us_) { done = true; break; }   // long flight: sleep on it
                }
            }
            if (done) { complete_flight(pending); have_pending = false; }
        }
        if (!have_pending && have_next) {
            pending = std::move(next); next = Flight(); have_next = false; have_pending = true;
            // nothing to overlap with: finish it right away when pipelining is off
            if (pipeline_depth_ < 2 || eng_->profiling()) { complete_flight(pending); have_pending = false; }
        }
    }
}

void Scheduler::cut_frames(std::vector<Session *> &work, bool &progressed)
{
    Lap lap;
    desc_.clear(); pcm_parts_.clear();
    size_t staged = 0;
    std::vector<Session *> finishers;
    // FbankFrameDesc::pcm_off is a 32-bit sample offset into ONE staging buffer: a pass stages at most `stage_limit` samples
    // (default 2^30; 1640 sessions x a full 8192-frame ring of backlog would pass 2^31) and carries the rest to the next pass
    // of launch_flight()'s loop.  APRIL_STAGE_LIMIT_SAMPLES exists for the test that crosses the limit with small numbers.
    static const size_t stage_limit = [] {
        const char *v = getenv("APRIL_STAGE_LIMIT_SAMPLES");
        const long n = v && *v ? atol(v) : (1L << 30);
        return (size_t)std::min(1L << 30, std::max(1L << 12, n));
    }();
    for (Session *s : work) {
        FrameBook &fb = s->fb;
        // new real frames: frame k covers stream samples [k*shift, k*shift + padded)  (fbank.c:195-236)
        if (fb.can_cut()) {
            if (staged + (size_t)fb.padded > stage_limit) { progressed = true; continue; }      // next pass
            const size_t base = staged;
            const size_t first = fb.fifo_pos;
            int cut = 0;
            // (up to a ring's worth per pass: a long feed then yields ~70 chunks per session at once for the layer-major step)
            while (fb.can_cut() && cut < fb.ring_frames && base + (size_t)cut * fb.shift + (size_t)fb.padded <= stage_limit) {
                FbankFrameDesc d; d.slot = s->slot; d.ring_row = fb.head; d.pcm_off = (int)(base + (size_t)cut * fb.shift);
                desc_.push_back(d);
                fb.head = (fb.head + 1) % fb.ring_frames;
                fb.rows_written += 1;
                fb.avail += 1;
                fb.avail_shadow = fb.avail;                       // fbank.c:300
                fb.fifo_pos += (size_t)fb.shift;
                ++cut;
            }
            const size_t last_end = first + (size_t)(cut - 1) * fb.shift + (size_t)fb.padded;
            const size_t fsz = fb.fifo.size();
            if (first < fsz) pcm_parts_.emplace_back(fb.fifo.data() + first, std::min(last_end, fsz) - first);          // (contiguous in staging)
            if (last_end > fsz) pcm_parts_.emplace_back(fb.ext + (std::max(first, fsz) - fsz), last_end - std::max(first, fsz));
            staged += last_end - first;
            s->compact_pending = true;                 // the fifo must not move until the window has been staged
            progressed = true;
            continue;
        }
        if (fb.chunk_ready() || s->flush_phase == 0) continue;
        // drained: flush state machine (april_session.c:552-563, fbank.c:308-325)
        switch (s->flush_phase) {
        case 1: case 3:
            if (fb.flush_allowed()) {
                // The reference drains with "pad the ring up to one segment, pull it, ask again" (src/fbank.c:308-325 under
                // src/april_session.c:552-560) until the shadow counter says stop.  Every round pads exactly what the pull before it
                // consumed and lowers the shadow counter by one segment step, so the number of rounds is known now: all their
                // padding rows go out in this pass and the rounds' chunks are stepped TOGETHER (one wavefront instead of one
                // single-chunk step per round: the 28 flush chunks of a session cost 2 steps instead of 28).  Same rows in the same
                // ring places, same counters afterwards.
                long av = fb.avail, sh = fb.avail_shadow;
                const long floor_sh = -(long)(fb.seg_count * 3);
                const long room = (long)fb.ring_frames - fb.avail;
                long pad = 0;
                while (sh >= floor_sh) {
                    const long need = av < fb.seg_count ? fb.seg_count - av : 0;
                    if (pad + need > room) break;                 // (never with the default ring; a tiny test ring takes the rest next pass)
                    pad += need; av += need;
                    av -= fb.seg_step; sh -= fb.seg_step;
                }
                for (long i = 0; i < pad; ++i) {
                    FbankFrameDesc d; d.slot = s->slot; d.ring_row = fb.head; d.pcm_off = -1;
                    desc_.push_back(d);
                    fb.head = (fb.head + 1) % fb.ring_frames;
                    fb.rows_written += 1;
                    fb.avail += 1;                                // padding does not touch the shadow counter
                }
            } else {
                s->flush_phase += 1;
            }
            progressed = true;
            break;
        case 2:
            fb.absorb_ext();
            fb.fifo.insert(fb.fifo.end(), (size_t)2 * 3200, (int16_t)0);      // april_session.c:555-556
            s->flush_phase = 3;
            progressed = true;
            break;
        case 4:
            // FINAL, clear context, SILENCE (april_session.c:561-563): the host part is replayed in order with the chunk
            // records of this flight; the device part (context reset + decoder refresh) is queued here
            s->replay.push_back(Session::Replay{-1, 0, 0, (uint32_t)s->now_ms, 1, 0});
            finishers.push_back(s);
            s->flush_phase = 0;
            progressed = true;
            break;
        default: break;
        }
    }
    tick_.host_ms[1] += lap();
    if (!desc_.empty()) {
        eng_->fbank((int)desc_.size(), desc_.data(), pcm_parts_.data(), pcm_parts_.size(), staged, &pool_);
        pool_.run(work.size(), 64, [&](size_t i) { Session *s = work[i]; if (s->compact_pending) { s->fb.compact(); s->compact_pending = false; } });
        tick_.frames += desc_.size();
        tick_.host_ms[2] += lap();
    }
    if (!finishers.empty()) {
        slots_.clear();
        for (Session *s : finishers) slots_.push_back(s->slot);
        eng_->decode_rows((int)slots_.size(), slots_.data(), 1);
        tick_.host_ms[6] += lap();
    }
}

// Layer-major step for a group of sessions that all have at least T chunks waiting (a long feed: "whole file at once",
// reference use case example.cpp:157-216 -> src/april_session.c:431-476).  The encoder does not depend on emitted tokens, so
// the T chunks of a session go through each layer together; see Engine::lm_step.
bool Scheduler::step_layer_major(std::vector<Session *> &group, int T, int mode)
{
    const int m = (int)group.size();
    const int rows = m * T;
    const NetDims &d = eng_->dims();
    if (!eng_->flight_has_room(rows + m, 1)) return false;
    const int stride_ms = model_->host.params.segment_step * model_->host.params.frame_shift_ms;
    slots_.clear(); tails_.assign((size_t)rows, 0); now_.assign((size_t)rows, 0);
    bool traced = false;
    for (int i = 0; i < m; ++i) {
        Session *s = group[(size_t)i];
        FrameBook &fb = s->fb;
        slots_.push_back(s->slot);
        for (int t = 0; t < T; ++t) {
            tails_[(size_t)t * m + i] = (fb.tail + t * fb.seg_step) % fb.ring_frames;          // fbank.c:327-349, T pulls
            now_[(size_t)t * m + i] = (int)(s->now_ms + (size_t)(t + 1) * stride_ms);         // april_session.c:442-443
        }
        fb.tail = (fb.tail + T * fb.seg_step) % fb.ring_frames;
        fb.avail -= (long)T * fb.seg_step;
        fb.avail_shadow -= (long)T * fb.seg_step;
        if (s->trace_buf) traced = true;
    }
    if (traced) logit_stage_.resize((size_t)3 * rows * d.vocab);
    const int k = eng_->lm_step(m, T, slots_.data(), tails_.data(), now_.data(), traced ? logit_stage_.data() : nullptr, mode);
    for (int i = 0; i < m; ++i) {
        Session *s = group[(size_t)i];
        for (int t = 0; t < T; ++t) {
            s->now_ms += (size_t)stride_ms;
            s->chunks++;
            s->replay.push_back(Session::Replay{k, i, m, (uint32_t)s->now_ms, 0, t});
            if (traced && s->trace_buf) {
                const StepRecord *recs = eng_->records(k);
                for (int r = 0; r < 3; ++r) {
                    const size_t at = ((size_t)t * 3 + r) * m + i;
                    const StepRecord &rec = recs[at];
                    if (!(rec.flags & REC_VALID)) break;
                    if (*s->trace_used + (size_t)d.vocab <= s->trace_cap) {
                        memcpy(s->trace_buf + *s->trace_used, logit_stage_.data() + at * d.vocab, (size_t)d.vocab * 4);
                        *s->trace_used += (size_t)d.vocab;
                    }
                    if (rec.flags & REC_BLANK) break;
                }
            }
        }
    }
    tick_.steps++; tick_.chunks += (uint64_t)rows;
    if (mode == 1) { tick_.wave_steps++; tick_.wave_chunks += (uint64_t)rows; }
    else { tick_.lm_steps++; tick_.lm_chunks += (uint64_t)rows; }
    if ((uint64_t)m > tick_.max_batch_seen) tick_.max_batch_seen = (uint64_t)m;
    return true;
}

bool Scheduler::step_chunks(std::vector<Session *> &ready)
{
    Lap lap;
    const NetDims &d = eng_->dims();
    const int MB = eng_->max_batch();
    // first use of a session: context = [blank, blank] (already in the slot's device state), run the decoder (april_session.c:432-438)
    slots_.clear();
    for (Session *s : ready) if (!s->dout_ready) {
        s->greedy.reset_context_to_blank();
        s->greedy.ctx_dirty = false;
        s->dout_ready = true;
        slots_.push_back(s->slot);
    }
    if (!slots_.empty()) { eng_->decode_rows((int)slots_.size(), slots_.data(), 0); tick_.host_ms[6] += lap(); }

    // sessions with a long backlog (a whole file fed at once) go layer-major, in groups that fit the work buffers
    std::vector<Session *> one, lm;
    auto waiting = [](const Session *s) { return (int)((s->fb.avail - s->fb.seg_count) / s->fb.seg_step + 1); };
    // (fp16 tile engines: layer-major since round 4 -- the tile kernels have the two halves of the gate GEMM; APRIL_F16_LM=0 falls back
    // to successive feed wavefronts of up to wave_max_chunks_ chunks)
    static const bool f16_lm = !(getenv("APRIL_F16_LM") && atoi(getenv("APRIL_F16_LM")) == 0);
    const bool lm_ok = lm_min_chunks_ > 0 && lm_min_chunks_ <= MB && (!eng_->f16_tile() || f16_lm);
    for (Session *s : ready) ((lm_ok && waiting(s) >= lm_min_chunks_) ? lm : one).push_back(s);
    // ... a FEW sessions: layer-major pays while the rows of a time step are a handful (the recurrent pair of a step as weight streams,
    // everything else batched over time); from ~48 sessions with a backlog the successive feed wavefronts of up to wave_max_chunks_
    // chunks are faster (aprilv0 dims, 2 .. 10 s handed over at once, ms per 100 ms of all sessions, layer-major vs wavefronts: 1 session
    // 0.13 vs 0.33, 32: 0.33 vs 0.49, 64: 0.73 vs 0.60, 256: 1.74 vs 1.31 -- round 6; an asynchronous client that runs ahead of the GPU
    // is the case: bench.py `reference_api_async` 1.93 -> 1.45 ms per step)
    static const size_t lm_max_sessions = (size_t)env_us("APRIL_LM_MAX_SESSIONS", 48);
    if (lm.size() > lm_max_sessions && wave_min_chunks_ > 0) { one.insert(one.end(), lm.begin(), lm.end()); lm.clear(); }
    if (!lm.empty()) {
        const int per = std::max(1, MB / lm_min_chunks_);
        std::vector<Session *> group;
        for (size_t o = 0; o < lm.size(); o += (size_t)per) {
            group.assign(lm.begin() + (long)o, lm.begin() + (long)std::min(lm.size(), o + (size_t)per));
            int T = MB / (int)group.size();
            for (Session *s : group) T = std::min(T, waiting(s));
            if (!step_layer_major(group, T)) { tick_.host_ms[3] += lap(); return false; }
        }
    }

    const int n = (int)one.size();
    // A feed usually carries 2..3 chunks for every session (100 ms = 2.5 chunks): those chunk steps run as ONE wavefront over
    // the layers (Engine::lm_step mode 1) when the rows fit the work buffers; what is left over goes chunk by chunk.
    if (n > 0 && wave_min_chunks_ > 0) {
        int T = MB / n;
        for (Session *s : one) T = std::min(T, waiting(s));
        T = std::min(T, wave_max_chunks_);
        if (T >= wave_min_chunks_) {
            const bool ok = step_layer_major(one, T, 1);
            tick_.host_ms[3] += lap();
            return ok;
        }
    }
    if (n > 0 && !eng_->flight_has_room(n, (n + MB - 1) / MB)) { tick_.host_ms[3] += lap(); return false; }
    const int stride_ms = model_->host.params.segment_step * model_->host.params.frame_shift_ms;
    for (int o = 0; o < n; o += MB) {
        const int m = std::min(MB, n - o);
        slots_.clear(); tails_.clear(); now_.clear();
        bool traced = false;
        for (int i = 0; i < m; ++i) {
            Session *s = one[(size_t)(o + i)];
            FrameBook &fb = s->fb;
            slots_.push_back(s->slot);
            tails_.push_back(fb.tail);                                  // fbank.c:327-349
            fb.tail = (fb.tail + fb.seg_step) % fb.ring_frames;
            fb.avail -= fb.seg_step;
            fb.avail_shadow -= fb.seg_step;
            s->now_ms += (size_t)stride_ms;                             // april_session.c:442-443
            s->chunks++;
            now_.push_back((int)s->now_ms);
            if (s->trace_buf) traced = true;
        }
        if (traced) logit_stage_.resize((size_t)3 * m * d.vocab);
        const int k = eng_->step(m, slots_.data(), tails_.data(), now_.data(), traced ? logit_stage_.data() : nullptr);
        for (int i = 0; i < m; ++i) {
            Session *s = one[(size_t)(o + i)];
            s->replay.push_back(Session::Replay{k, i, m, (uint32_t)s->now_ms, 0, 0});
            if (traced && s->trace_buf) {                               // tests: the logits of every round that ran, in order
                const StepRecord *recs = eng_->records(k);
                for (int r = 0; r < 3; ++r) {
                    const StepRecord &rec = recs[(size_t)r * m + i];
                    if (!(rec.flags & REC_VALID)) break;
                    if (*s->trace_used + (size_t)d.vocab <= s->trace_cap) {
                        memcpy(s->trace_buf + *s->trace_used, logit_stage_.data() + ((size_t)r * m + i) * d.vocab, (size_t)d.vocab * 4);
                        *s->trace_used += (size_t)d.vocab;
                    }
                    if (rec.flags & REC_BLANK) break;
                }
            }
        }
        tick_.steps++; tick_.chunks += (uint64_t)m;
        if ((uint64_t)m > tick_.max_batch_seen) tick_.max_batch_seen = (uint64_t)m;
    }
    tick_.host_ms[3] += lap();
    return true;
}

// After the flight: per session, in the order things happened, feed the device's per-round records to the search state
// machine (which builds the callbacks) and check that it takes the same decisions the device took.
void Scheduler::replay(Flight &f)
{
    for (size_t w = 0; w < f.work.size(); ++w) {
        Session *s = f.work[w];
        // the first mark[w] items of the session's list are this flight's (flights complete in the order they were launched)
        const size_t n = f.mark[w];
        for (size_t q = 0; q < n; ++q) {
            const Session::Replay &it = s->replay[q];
            if (it.kind == 1) { s->greedy.finish_flush(s->events); s->greedy.ctx_dirty = false; continue; }
            const StepRecord *recs = eng_->records(it.step);
            for (int r = 0; r < 3; ++r) {                               // april_session.c:449-454
                const StepRecord &rec = recs[((size_t)it.chunk * 3 + r) * it.rows + it.row];
                if (!(rec.flags & REC_VALID)) { tick_.replay_mismatch++; LOGE("replay: device skipped a round the host expected (slot %d)", s->slot); break; }
                const JointResult jr{rec.idx, rec.max_val, rec.blank_val};
                const bool blank = s->greedy.on_joint(jr, r == 0 ? 1.0f : 0.0f, (size_t)it.now_ms, s->events);
                const bool ctx = s->greedy.ctx_dirty;
                s->greedy.ctx_dirty = false;
                tick_.rounds++;
                if (blank != ((rec.flags & REC_BLANK) != 0) || ctx != ((rec.flags & REC_CTX) != 0)) {
                    tick_.replay_mismatch++;
                    LOGE("replay: host and device decisions differ (slot %d, round %d: host blank=%d ctx=%d, device flags=%u)", s->slot, r, (int)blank, (int)ctx, rec.flags);
                }
                if (blank) break;
            }
        }
        s->replay.erase(s->replay.begin(), s->replay.begin() + (long)n);
    }
}

}  // namespace aprilx
