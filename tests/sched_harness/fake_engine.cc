// Host-only stand-in for the GPU engine (csrc/engine.cc) and for the few HIP / RCCL entry points the host runtime calls, so that
// the REAL scheduler and C ABI -- csrc/session.cc, csrc/april_api.cc, csrc/host_pool.h, the loader -- can run under
// -fsanitize=thread and -fsanitize=address,undefined on a machine without a GPU (reference threading
// contract src/april_session.c:479-493,567-585, src/audio_provider.c:25-40).  Test infrastructure: nothing here ships.
//
// What the fake keeps of the engine's contract, because the scheduler's correctness depends on it:
//   * flights: begin / step / lm_step / decode_rows / close / flight_done / wait; at most two open; a step's records become
//     VISIBLE (records()) only when its flight has completed -- until then the visible ring holds poison (flags 0), so a
//     scheduler that reads records early, or reuses a parity too soon, shows up as a replay mismatch;
//   * completion is asynchronous: a flight is "done" a random 20..400 us after it was closed (FAKE_DELAY_US), polled by
//     flight_done() or waited for by wait_flight();
//   * the index / record rings are small (FAKE_STEP_CAP steps per flight) so that flights fill up and continue in follow-up flights;
//   * the search decisions are made by a per-slot copy of the product's own host state machine (Greedy) fed with pseudo-random
//     joiner results that depend only on the session's own time line -- so the device flags the scheduler replays against are
//     exactly consistent (replay_mismatch must stay 0) and the callbacks of a session do not depend on the ingest mode;
//   * fbank() reads every staged PCM window (a lent caller buffer that was freed or changed too early is an ASan / TSan report)
//     and checks the frame descriptors against the staged range.
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <thread>
#include "engine.h"
#include "session.h"
#include "common.h"
#include <rccl/rccl.h>

namespace aprilx {

namespace {

struct FakeState {
    std::vector<Greedy> mirror;                 // per slot: the decisions the "device" takes
    std::vector<uint8_t> mirror_init;
    std::vector<StepRecord> stage;              // records as the "device" has written them (both parities)
    std::chrono::steady_clock::time_point done_at[2];
    bool closed[2] = {false, false}, published[2] = {true, true};
    size_t rec_first[2] = {0, 0}, rec_last[2] = {0, 0};
    std::vector<uint8_t> tok_class;
    std::mt19937 rng{12345};
    int delay_lo = 20, delay_hi = 400;
    std::atomic<uint64_t> pcm_sum{0};
    uint64_t frames = 0, steps = 0;
};
std::mutex g_mu;
std::map<const Engine *, std::unique_ptr<FakeState>> g_state;
FakeState &st(const Engine *e) { std::lock_guard<std::mutex> g(g_mu); return *g_state[e]; }

int env_i(const char *n, int d) { const char *v = getenv(n); return v && *v ? atoi(v) : d; }

uint32_t mix(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t h = a * 2654435761u ^ (b + 0x9e3779b9u + (a << 6) + (a >> 2));
    h ^= c * 40503u + 0x7f4a7c15u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}

}  // namespace

std::recursive_mutex &hip_legacy_mutex() { static std::recursive_mutex m; return m; }

void plan_layout(const NetDims &d, bool has_dec_conv_b, PackedLayout &L)
{
    L = PackedLayout();
    L.dims = d; L.has_dec_conv_b = has_dec_conv_b; L.layers.resize((size_t)d.n_layers); L.total = 64; L.vocab_pad = (d.vocab + 15) / 16 * 16;
    L.norm_eps.assign((size_t)d.n_layers, 0.25f);
}
std::vector<std::pair<size_t, size_t>> gemm_sections(const PackedLayout &) { return {}; }
void pack_weights(const HostModel &, PackedLayout &L, std::vector<float> &blob) { blob.assign(L.total, 0.0f); }

Engine::Engine(const EngineConfig &cfg, const PackedLayout &layout, const float *, const float *, const ModelParams &params, const FbankHostTables &,
               const std::vector<uint8_t> &tok_class)
    : cfg_(cfg), L_(layout), P_(params)
{
    auto fs = std::make_unique<FakeState>();
    fs->tok_class = tok_class;
    fs->mirror.resize((size_t)cfg.max_slots); fs->mirror_init.assign((size_t)cfg.max_slots, 0);
    step_cap_ = env_i("FAKE_STEP_CAP", 4);                        // steps per flight (power of two)
    rec_cap_ = (size_t)env_i("FAKE_REC_CAP", 1 << 13);            // records per flight
    fs->stage.assign(2 * rec_cap_, StepRecord{0, 0.f, 0.f, 0u});
    rec_h_ = new StepRecord[2 * rec_cap_]();
    rec_off_h_ = new int[(size_t)2 * step_cap_]();
    fs->delay_lo = env_i("FAKE_DELAY_US", 20); fs->delay_hi = std::max(fs->delay_lo, env_i("FAKE_DELAY_US_MAX", 400));
    ring_frames_ = env_i("APRIL_RING_FRAMES", 8192);
    for (int s = cfg.max_slots - 1; s >= 0; --s) free_.push_back(s);
    w_ = new float[L_.total]();
    std::lock_guard<std::mutex> g(g_mu);
    g_state[this] = std::move(fs);
}

Engine::~Engine()
{
    delete[] rec_h_; delete[] rec_off_h_; delete[] w_;
    std::lock_guard<std::mutex> g(g_mu);
    g_state.erase(this);
}

void Engine::finish_weights() {}

int Engine::alloc_slot()
{
    std::lock_guard<std::mutex> g(slot_mu_);
    if (free_.empty()) return -1;
    const int s = free_.back(); free_.pop_back(); ++live_;
    FakeState &f = st(this);
    f.mirror_init[(size_t)s] = 0;
    return s;
}

void Engine::free_slot(int slot)
{
    std::lock_guard<std::mutex> g(slot_mu_);
    free_.push_back(slot); --live_;
}

void Engine::fbank(int n_frames, const FbankFrameDesc *desc, const std::pair<const int16_t *, size_t> *parts, size_t n_parts, size_t n_pcm, HostPool *)
{
    FakeState &f = st(this);
    uint64_t sum = 0; size_t total = 0;
    for (size_t p = 0; p < n_parts; ++p) {                      // every staged sample is read, as the pinned-staging gather does
        const int16_t *q = parts[p].first;
        for (size_t i = 0; i < parts[p].second; ++i) sum += (uint16_t)q[i];
        total += parts[p].second;
    }
    if (total != n_pcm) { LOGE("fake engine: fbank: windows hold %zu samples, caller says %zu", total, n_pcm); abort(); }
    for (int i = 0; i < n_frames; ++i) {
        if (desc[i].slot < 0 || desc[i].slot >= cfg_.max_slots || desc[i].ring_row < 0 || desc[i].ring_row >= ring_frames_) { LOGE("fake engine: fbank: bad descriptor"); abort(); }
        if (desc[i].pcm_off >= 0 && (size_t)desc[i].pcm_off + (size_t)1 > n_pcm) { LOGE("fake engine: fbank: frame window outside the staged samples"); abort(); }
    }
    f.pcm_sum.fetch_add(sum, std::memory_order_relaxed);
    f.frames += (uint64_t)n_frames;
}

void Engine::begin_flight()
{
    FakeState &f = st(this);
    const int p = flight_parity_ = next_parity_; next_parity_ ^= 1;
    if (f.closed[p] && !f.published[p]) { LOGE("fake engine: begin_flight reuses parity %d before its flight completed", p); abort(); }
    f.closed[p] = false; f.published[p] = false;
    rec_base_ = (size_t)p * rec_cap_; rec_pos_ = rec_base_; flight_steps_ = 0;
    f.rec_first[p] = f.rec_last[p] = rec_base_;
    for (size_t i = 0; i < rec_cap_; ++i) rec_h_[rec_base_ + i] = StepRecord{-1, 0.f, 0.f, 0u};      // poison until the flight completes
}

bool Engine::flight_has_room(int rows, int nsteps) const
{
    return flight_steps_ + nsteps <= step_cap_ && rec_pos_ + (size_t)3 * (size_t)rows <= rec_base_ + rec_cap_;
}

// the three joiner rounds of one chunk of one row: results from a hash of the session's own time line, decisions by the mirror
static void fake_rounds(FakeState &f, const ModelParams &P, int slot, int now_ms, StepRecord *recs, size_t stride, size_t at)
{
    Greedy &g = f.mirror[(size_t)slot];
    std::vector<Event> sink;
    const int vocab = P.token_count;
    for (int r = 0; r < 3; ++r) {
        const uint32_t h = mix((uint32_t)now_ms, (uint32_t)r, 77u);
        JointResult jr;
        const bool want_blank = (h % 100u) < 72u;
        jr.idx = want_blank ? P.blank_id : (int32_t)(1 + (h >> 8) % (uint32_t)std::max(1, vocab - 1));
        if (jr.idx == P.blank_id && !want_blank) jr.idx = (P.blank_id + 1) % vocab;
        jr.max_val = (float)((h >> 4) % 1000u) * 0.01f - 2.0f;
        jr.blank_val = jr.max_val - (float)((h >> 14) % 700u) * 0.01f;
        const bool blank = g.on_joint(jr, r == 0 ? 1.0f : 0.0f, (size_t)now_ms, sink);
        const bool ctx = g.ctx_dirty;
        g.ctx_dirty = false;
        recs[at + (size_t)r * stride] = StepRecord{jr.idx, jr.max_val, jr.blank_val, (uint32_t)(REC_VALID | (blank ? REC_BLANK : 0) | (ctx ? REC_CTX : 0))};
        if (blank) break;
    }
}

int Engine::step(int m, const int *slots, const int *ring_tails, const int *now_ms, float *logits_out)
{
    return lm_step(m, 1, slots, ring_tails, now_ms, logits_out, 0);
}

int Engine::lm_step(int m, int T, const int *slots, const int *ring_tails, const int *now_ms, float *logits_out, int)
{
    FakeState &f = st(this);
    if (!flight_has_room(m * T, 1)) { LOGE("fake engine: step without room in the flight's rings"); abort(); }
    if (m * T > cfg_.max_batch) { LOGE("fake engine: %d rows > max_batch", m * T); abort(); }
    const int k = next_step_index();
    rec_off_h_[k] = (int)rec_pos_;
    StepRecord *recs = f.stage.data() + rec_pos_;
    for (size_t i = 0; i < (size_t)3 * m * T; ++i) recs[i] = StepRecord{0, 0.f, 0.f, 0u};
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < m; ++i) {
            if (ring_tails[(size_t)t * m + i] < 0 || ring_tails[(size_t)t * m + i] >= ring_frames_) { LOGE("fake engine: bad ring tail"); abort(); }
            fake_rounds(f, P_, slots[i], now_ms[(size_t)t * m + i], recs, (size_t)m, (size_t)t * 3 * m + i);      // [T][3][m]
        }
    if (logits_out) memset(logits_out, 0, (size_t)3 * m * T * L_.dims.vocab * sizeof(float));
    rec_pos_ += (size_t)3 * m * T;
    f.rec_last[flight_parity_] = rec_pos_;
    ++f.steps;
    return k;
}

void Engine::decode_rows(int n, const int *slots, int op)
{
    FakeState &f = st(this);
    std::vector<Event> sink;
    for (int i = 0; i < n; ++i) {
        Greedy &g = f.mirror[(size_t)slots[i]];
        if (op == 0) {                                   // first use of a session in this slot
            g = Greedy();
            g.init(&P_, &f.tok_class);
            g.reset_context_to_blank(); g.ctx_dirty = false;
            f.mirror_init[(size_t)slots[i]] = 1;
        } else { g.finish_flush(sink); g.ctx_dirty = false; }
    }
}

int Engine::close_flight()
{
    FakeState &f = st(this);
    const int p = flight_parity_;
    std::uniform_int_distribution<int> d(f.delay_lo, f.delay_hi);
    f.done_at[p] = std::chrono::steady_clock::now() + std::chrono::microseconds(d(f.rng));
    f.closed[p] = true;
    return p;
}

static void publish(FakeState &f, StepRecord *visible, int p)
{
    if (f.published[p]) return;
    for (size_t i = f.rec_first[p]; i < f.rec_last[p]; ++i) visible[i] = f.stage[i];
    f.published[p] = true;
}

bool Engine::flight_done(int parity)
{
    FakeState &f = st(this);
    if (!f.closed[parity]) { LOGE("fake engine: flight_done on a flight that was never closed"); abort(); }
    if (std::chrono::steady_clock::now() < f.done_at[parity]) return false;
    publish(f, rec_h_, parity);
    return true;
}

void Engine::wait_flight(int parity)
{
    FakeState &f = st(this);
    if (!f.closed[parity]) { LOGE("fake engine: wait_flight on a flight that was never closed"); abort(); }
    std::this_thread::sleep_until(f.done_at[parity]);
    publish(f, rec_h_, parity);
}

void Engine::end_flight() { wait_flight(close_flight()); }
void Engine::sync() {}
void Engine::sync_streams() {}
void Engine::set_profiling(bool on) { profiling_ = on; }
void Engine::set_gates_clock(bool on) { gclk_ = on; }
void Engine::reset_timing() {}
void Engine::read_ring(int, int, int n_rows, float *out) { memset(out, 0, (size_t)n_rows * L_.dims.mel * sizeof(float)); }
void Engine::read_greedy_state(int slot, GreedyState *out)
{
    FakeState &f = st(this);
    const Greedy &g = f.mirror[(size_t)slot];
    *out = GreedyState{g.ctx[0], g.ctx[1], 0, 0u};
}
void Engine::debug_encoder(int, const float *, const float *, const float *, float *, float *, float *) { abort(); }
void Engine::debug_decoder(int, const int64_t *, float *) { abort(); }
void Engine::debug_joiner(int, const float *, const float *, float *) { abort(); }
void Engine::debug_decide(int, int, const float *, float, const int *, int, int32_t *, StepRecord *) { abort(); }
void Engine::debug_fbank(int, const int16_t *, float *) { abort(); }

// planner entry points of the kernel files that april_api.cc exposes for CPU-side tests: not part of this harness
bool gemm_fullk(int, int, int, bool, int, int) { return true; }
int gemm_partials(int, int, int, int, int) { return 1; }
bool gemm_tile_planned(int, int, int, int) { return false; }
int recur_form(const GemmArgs &) { return 0; }
int gemm_kw_waves(const GemmArgs &) { return 0; }

}  // namespace aprilx

// ---- the HIP / RCCL entry points the host runtime calls (no device behind them)
extern "C" {
hipError_t hipGetDeviceCount(int *count) { *count = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return hipSuccess; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { if (dst && src) memcpy(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyPeer(void *dst, int, const void *src, int, size_t n) { if (dst && src) memcpy(dst, src, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "fake HIP"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInternalError; }
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int) { return ncclInternalError; }
ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t) { return ncclSuccess; }
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
ncclResult_t ncclBroadcast(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { return ncclInternalError; }
const char *ncclGetErrorString(ncclResult_t) { return "fake RCCL"; }
}
