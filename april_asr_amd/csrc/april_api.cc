// The exported C ABI: the reference's eleven aam_* / aas_* entry points
// (include/april_api.h, reference april_api.h:58-196) and the engine-level
// aprilx_* entry points (include/aprilx_engine.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <cmath>
#include "../../include/april_api.h"
#include "../../include/aprilx_engine.h"
#include "common.h"
#include "session.h"
#include "rccl_group.h"

using namespace aprilx;

namespace aprilx { int g_loglevel = LOG_WARNING; }

struct AprilASRModel_i { Model m; };
struct AprilASRSession_i { Session s; };

namespace {
bool g_inited = false;
int g_client_version = 0;
std::vector<int> g_devices;

int env_int(const char *name, int def)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : def;
}

#define RCCL_TRY(expr)                                                                     \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess) { LOGE("RCCL: %s failed: %s", #expr, ncclGetErrorString(r_)); return false; } \
    } while (0)

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
double aprilx_now_ms() { return now_ms(); }

// One process, several GPUs (APRIL_GPU_DEVICES=0,1,...): the packed weights are uploaded ONCE, to the first device, and
// broadcast from there to the other devices' engines with RCCL over xGMI (grouped ncclBroadcast on a communicator made by
// ncclCommInitAll) -- the one collective of this system, at model load (reference load site src/april_model.c:57-61).
bool broadcast_local(Model &m)
{
    // The process-wide legacy-stream lock (engine.h) is taken around this function's own device copies only, NEVER across RCCL
    // bootstrap or collective waits: those block on other ranks for an unbounded time, and a stepping thread of another live model
    // that needs a new graph shape would wait behind them holding its capture lock.  RCCL's own set-up work is safe beside
    // a relaxed-mode capture of another thread -- measured: tools/rccl_capture_probe (profiles/r06_rccl_capture_probe.txt), 27 720
    // captures beside communicator init + broadcast + destroy, none failed; the same probe's legacy-stream hipMemcpy leg aborts.
    std::vector<Engine *> peers;            // one engine per distinct device, root first
    std::vector<int> devs;
    // APRIL_FAULT_RCCL (fault injection, tests): 1 = every engine counts as a broadcast peer even when it shares a device with
    // another one, so that on a ONE-GPU box (APRIL_GPU_DEVICES=0,0) RCCL is asked for a communicator over duplicate devices
    // and genuinely fails -> the failure / fallback / clean-up paths below run; 2 = the broadcast to peer 1 reports an error
    // inside the open group (needs two real devices) -> the abort path
    const int fault = env_int("APRIL_FAULT_RCCL", 0);
    for (Engine *e : m.engines) if (fault == 1 || std::find(devs.begin(), devs.end(), e->device()) == devs.end()) { devs.push_back(e->device()); peers.push_back(e); }
    const size_t count = m.layout.total;
    if (devs.size() > 1) {
        // Every exit path destroys the communicators; an RCCL failure is loud, fails the load under APRIL_STRICT_RCCL=1 and
        // otherwise falls back to a peer copy from the root device (the weights are already there), so a node whose RCCL
        // cannot initialise still serves.
        const double t0 = now_ms();
        std::vector<ncclComm_t> comms(devs.size(), nullptr);
        double t1 = t0;
        // RCCL + HIP behind the policy of rccl_group.h (the same function runs against a failing stub in tests/cpp/rccl_group_test.cc)
        struct RealRccl {
            typedef ncclComm_t Comm;
            std::vector<Engine *> &peers; const std::vector<int> &devs; size_t count; int fault;
            bool comm_init_all(Comm *c, int n, const int *d) { RCCL_TRY(ncclCommInitAll(c, n, d)); return true; }
            bool group_start() { RCCL_TRY(ncclGroupStart()); return true; }
            bool group_end() { const ncclResult_t r = ncclGroupEnd(); if (r != ncclSuccess) { LOGE("RCCL: ncclGroupEnd failed: %s", ncclGetErrorString(r)); return false; } return true; }
            bool broadcast(int i, Comm comm) {
                // (fault 2: the call itself is made with null buffers, so that RCCL's own argument check fails INSIDE the group and
                // records the group error, as a real failure would)
                const bool inject = fault == 2 && i == 1;
                const ncclResult_t r = ncclBroadcast(inject ? nullptr : peers[0]->weights_device(), inject ? nullptr : peers[(size_t)i]->weights_mut(), count, ncclFloat, 0, comm, peers[(size_t)i]->stream());
                if (r != ncclSuccess) { LOGE("RCCL: ncclBroadcast (device %d) failed: %s", devs[(size_t)i], ncclGetErrorString(r)); return false; }
                return true;
            }
            void comm_abort(Comm c) { (void)ncclCommAbort(c); }
            bool set_device(int d) { return hipSetDevice(d) == hipSuccess; }
            bool stream_sync(int i) { return hipStreamSynchronize(peers[(size_t)i]->stream()) == hipSuccess; }
            double now_ms() { return aprilx_now_ms(); }
        } api{peers, devs, count, fault};
        auto rccl_path = [&]() -> bool { return rccl_group_broadcast(api, devs, comms, &t1); };
        const bool used = rccl_path();
        for (ncclComm_t c : comms) if (c) (void)ncclCommDestroy(c);
        if (used) {
            m.load.broadcast_ms = now_ms() - t1; m.load.comm_init_ms = t1 - t0;
            m.load.broadcast_bytes = count * 4; m.load.ranks = (int)devs.size(); m.load.used_rccl = 1;
        } else {
            if (env_int("APRIL_STRICT_RCCL", 0)) { LOGE("aam: RCCL weight broadcast failed and APRIL_STRICT_RCCL=1: giving up"); return false; }
            LOGE("aam: RCCL weight broadcast failed: falling back to peer copies from device %d (used_rccl = 0)", devs[0]);
            const double t2 = now_ms();
            HipLegacyLock legacy;           // (legacy-stream copies: not while another engine captures a graph)
            for (size_t i = 1; i < peers.size(); ++i) {
                HIP_CHECK(hipSetDevice(devs[i]));
                if (devs[i] == devs[0]) HIP_CHECK(hipMemcpy(peers[i]->weights_mut(), peers[0]->weights_device(), count * 4, hipMemcpyDeviceToDevice));
                else HIP_CHECK(hipMemcpyPeer(peers[i]->weights_mut(), devs[i], peers[0]->weights_device(), devs[0], count * 4));
            }
            m.load.broadcast_ms = now_ms() - t2; m.load.comm_init_ms = 0;
            m.load.broadcast_bytes = count * 4; m.load.ranks = (int)devs.size(); m.load.used_rccl = 0;
        }
    }
    // further engines on a device that already holds the weights ("lanes"): a device-to-device copy
    HipLegacyLock legacy;
    for (Engine *e : m.engines) {
        if (std::find(peers.begin(), peers.end(), e) != peers.end()) continue;
        Engine *src = peers[(size_t)(std::find(devs.begin(), devs.end(), e->device()) - devs.begin())];
        HIP_CHECK(hipSetDevice(e->device()));
        HIP_CHECK(hipMemcpy(e->weights_mut(), src->weights_device(), count * 4, hipMemcpyDeviceToDevice));
    }
    return true;
}

// blob_host / blob_device (on g_devices[0]) may both be null: the caller fills engine 0's weights (RCCL receive) and then
// calls distribute_weights() itself
bool create_engines(Model &m, const float *blob_host, const float *blob_device)
{
    const ModelParams &P = m.host.params;
    if (!build_fbank_tables(P.sample_rate, P.frame_shift_ms, P.frame_length_ms, P.mel_features, P.round_pow2 != 0, P.mel_low, P.mel_high, m.ftab)) {
        LOGE("aam: unsupported frame length (an FFT size of 8 .. 8192 that pocketfft runs through its radix passes; a large prime factor means Bluestein, which is not built)");
        return false;
    }
    if (m.layout.dims.embed_in % 64 || m.layout.dims.d_model % 64 || m.layout.dims.hidden % 64 || m.layout.dims.ffn % 64 || m.layout.dims.joiner % 64 || m.layout.dims.conv_ch[2] % 16) {
        LOGE("aam: layer widths must be multiples of 64 for the MFMA kernels (pad_host_model rounds a file's widths up at load: this model did not come through it)");
        return false;
    }
    if (m.layout.dims.d_model > 2048) { LOGE("aam: d_model > 2048 unsupported (row scales are staged for at most 64 column groups)"); return false; }
    m.tok_class = classify_tokens(P);
    EngineConfig cfg;
    cfg.max_slots = env_int("APRIL_MAX_SESSIONS", 4096);
    // rows of the work buffers = sessions x chunks stepped together (a 100 ms feed of 2048 sessions is 3 x 2048 rows); < 1 GB at 8192
    cfg.max_batch = std::max(1, env_int("APRIL_MAX_BATCH", 8192));
    if (const char *pv = getenv("APRIL_PRECISION")) {
        const std::string v(pv);
        if (v == "f16" || v == "fp16" || v == "half") cfg.precision = 1;
        else if (!(v.empty() || v == "f32" || v == "fp32")) { LOGE("aam: APRIL_PRECISION must be f32 or f16 (got '%s')", pv); return false; }
    }
    for (size_t i = 0; i < g_devices.size(); ++i) {
        cfg.device = g_devices[i];
        Engine *e = new Engine(cfg, m.layout, i == 0 ? blob_host : nullptr, i == 0 ? blob_device : nullptr, P, m.ftab, m.tok_class);
        m.engines.push_back(e);
    }
    return true;
}

bool distribute_weights(Model &m)
{
    if (!broadcast_local(m)) return false;
    for (Engine *e : m.engines) { e->finish_weights(); m.scheds.push_back(new Scheduler(&m, e)); }
    return true;
}

bool build_runtime(Model &m, const float *blob_host, const float *blob_device)
{
    return create_engines(m, blob_host, blob_device) && distribute_weights(m);
}

// ---- blob (de)serialisation: [magic][meta_bytes][weight_floats][meta][pad to 256][weights]
struct BlobHeader { char magic[8]; uint64_t meta_bytes, weight_floats, weights_offset; };

void put(std::string &b, const void *p, size_t n) { b.append((const char *)p, n); }
template <typename T> void putv(std::string &b, const T &v) { put(b, &v, sizeof v); }
void puts_(std::string &b, const std::string &s) { uint64_t n = s.size(); putv(b, n); put(b, s.data(), n); }

std::string make_meta(const Model &m)
{
    std::string b;
    puts_(b, m.host.language); puts_(b, m.host.name); puts_(b, m.host.description);
    const ModelParams &P = m.host.params;
    int32_t ints[13] = {P.batch_size, P.segment_size, P.segment_step, P.mel_features, P.sample_rate, P.frame_shift_ms, P.frame_length_ms,
                        P.round_pow2, P.mel_low, P.mel_high, P.snip_edges, P.token_count, P.blank_id};
    put(b, ints, sizeof ints);
    uint64_t stride = P.token_stride; putv(b, stride);
    put(b, P.tokens.data(), P.tokens.size());
    putv(b, m.layout.dims);
    uint8_t hb = m.layout.has_dec_conv_b; putv(b, hb);
    putv(b, m.layout.embed_eps);
    put(b, m.layout.norm_eps.data(), m.layout.norm_eps.size() * 4);
    return b;
}

struct MetaRd {
    const char *p; size_t n, pos = 0; bool bad = false;
    void get(void *dst, size_t k) { if (pos + k > n) { bad = true; return; } memcpy(dst, p + pos, k); pos += k; }
    template <typename T> T val() { T v{}; get(&v, sizeof v); return v; }
    std::string str() { uint64_t k = val<uint64_t>(); if (bad || pos + k > n) { bad = true; return ""; } std::string s(p + pos, (size_t)k); pos += (size_t)k; return s; }
};

bool parse_meta(const char *p, size_t n, Model &m)
{
    MetaRd r{p, n};
    m.host.language = r.str(); m.host.name = r.str(); m.host.description = r.str();
    int32_t ints[13]; r.get(ints, sizeof ints);
    ModelParams &P = m.host.params;
    P.batch_size = ints[0]; P.segment_size = ints[1]; P.segment_step = ints[2]; P.mel_features = ints[3]; P.sample_rate = ints[4];
    P.frame_shift_ms = ints[5]; P.frame_length_ms = ints[6]; P.round_pow2 = ints[7]; P.mel_low = ints[8]; P.mel_high = ints[9];
    P.snip_edges = ints[10]; P.token_count = ints[11]; P.blank_id = ints[12];
    P.token_stride = (size_t)r.val<uint64_t>();
    std::string perr;
    if (r.bad || !validate_params(P, perr) || P.token_stride == 0 || P.token_stride > 4096) { if (!perr.empty()) LOGE("aprilx: blob %s", perr.c_str()); return false; }
    P.tokens.resize((size_t)P.token_count * P.token_stride);
    r.get(P.tokens.data(), P.tokens.size());
    NetDims d = r.val<NetDims>();
    const bool hb = r.val<uint8_t>() != 0;
    if (r.bad || d.n_layers <= 0 || d.n_layers > 256) return false;
    for (size_t t = 0; t < (size_t)P.token_count; ++t) P.tokens[(t + 1) * P.token_stride - 1] = 0;    // token text is NUL-terminated within its slot
    auto in = [](long v, long lo, long hi) { return v >= lo && v <= hi; };
    if (!in(d.d_model, 16, 1 << 16) || !in(d.hidden, 16, 1 << 16) || !in(d.ffn, 16, 1 << 18) || !in(d.joiner, 16, 1 << 16) ||
        !in(d.vocab, 2, 1 << 20) || d.vocab != P.token_count || !in(d.mel, 8, 1024) || !in(d.seg, 3, 64) || d.context != 2 ||
        !in(d.dec_groups, 1, d.d_model) || d.d_model % d.dec_groups != 0 || !in(d.f_out, 1, 4096) || !in(d.embed_in, 16, 1 << 22) ||
        !in(d.conv_ch[0], 1, 4096) || !in(d.conv_ch[1], 1, 4096) || !in(d.conv_ch[2], 1, 4096) ||
        d.seg != P.segment_size || d.mel != P.mel_features ||                      // the network input is the PARAMS chunk (april_model.c:65-72)
        !in(d.conv_stride[0], 1, 8) || !in(d.conv_stride[1], 1, 8) || !in(d.conv_stride[2], 1, 8) ||
        d.embed_in != d.conv_ch[2] * d.f_out || !in(d.d_norm, 0, d.d_model))
        return false;
    plan_layout(d, hb, m.layout);
    m.layout.embed_eps = r.val<float>();
    m.layout.norm_eps.resize((size_t)d.n_layers);
    r.get(m.layout.norm_eps.data(), (size_t)d.n_layers * 4);
    m.host.dims = d;
    auto eps_ok = [](float e) { return std::isfinite(e) && e > 0.0f; };
    if (r.bad || !eps_ok(m.layout.embed_eps)) return false;
    for (float e : m.layout.norm_eps) if (!eps_ok(e)) return false;
    return true;
}

void free_host_weights(HostModel &h)
{
    for (int i = 0; i < 3; ++i) { std::vector<float>().swap(h.conv_w[i]); std::vector<float>().swap(h.conv_b[i]); }
    std::vector<float>().swap(h.w_embed); std::vector<LayerWeights>().swap(h.layers);
    std::vector<float>().swap(h.w_encproj); std::vector<float>().swap(h.emb); std::vector<float>().swap(h.dec_conv);
    std::vector<float>().swap(h.w_decproj); std::vector<float>().swap(h.w_out);
}
}  // namespace

namespace {
void crash_backtrace(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "libapril(mi355x): fatal signal, native backtrace:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
}  // namespace

extern "C" {

// ------------------------------------------------------------------ reference ABI
void aam_api_init(int version)
{
    if (env_int("APRIL_BACKTRACE", 0)) { signal(SIGSEGV, crash_backtrace); signal(SIGABRT, crash_backtrace); }   // debugging aid
    g_client_version = version;                       // stored, never checked (reference src/init.c:34)
    if (const char *lv = getenv("APRIL_LOG_LEVEL")) {
        static const char *names[5] = {"DEBUG", "INFO", "WARNING", "ERROR", "NONE"};
        for (int i = 0; i < 5; ++i) if (strcmp(lv, names[i]) == 0) g_loglevel = i;
    }
    g_devices.clear();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        LOGE("aam_api_init: no HIP device visible; aam_create_model will fail (this build has no CPU path)");
        g_inited = false;
        return;
    }
    if (const char *dv = getenv("APRIL_GPU_DEVICES")) {
        std::stringstream ss(dv); std::string item;
        while (std::getline(ss, item, ',')) if (!item.empty()) { int d = atoi(item.c_str()); if (d >= 0 && d < count) g_devices.push_back(d); }
    }
    if (g_devices.empty()) g_devices.push_back(0);
    g_inited = true;
}

AprilASRModel aam_create_model(const char *model_path)
{
    if (!g_inited) { LOGE("aam: not initialised (call aam_api_init; a HIP device is required)"); return nullptr; }
    AprilASRModel_i *h = new AprilASRModel_i();
    std::string err;
    if (!load_april_file(model_path, h->m.host, err) || !pad_host_model(h->m.host, err)) { LOGE("aam: failed to load %s: %s", model_path ? model_path : "(null)", err.c_str()); delete h; return nullptr; }
    plan_layout(h->m.host.dims, !h->m.host.dec_conv_b.empty(), h->m.layout);
    std::vector<float> blob;
    pack_weights(h->m.host, h->m.layout, blob);
    if (!build_runtime(h->m, blob.data(), nullptr)) { delete h; return nullptr; }
    free_host_weights(h->m.host);
    LOGI("aam: loaded model %s", h->m.host.name.c_str());
    return h;
}

// Host-only load: parse + extract + pack, no GPU objects.  For loader / state-machine tests on
// machines without a GPU; sessions cannot be created on such a model.
AprilASRModel aprilx_model_load_host(const char *model_path)
{
    AprilASRModel_i *h = new AprilASRModel_i();
    std::string err;
    if (!load_april_file(model_path, h->m.host, err) || !pad_host_model(h->m.host, err)) { LOGE("aam: failed to load %s: %s", model_path ? model_path : "(null)", err.c_str()); delete h; return nullptr; }
    plan_layout(h->m.host.dims, !h->m.host.dec_conv_b.empty(), h->m.layout);
    pack_weights(h->m.host, h->m.layout, h->m.host_blob);
    const ModelParams &P = h->m.host.params;
    if (!build_fbank_tables(P.sample_rate, P.frame_shift_ms, P.frame_length_ms, P.mel_features, P.round_pow2 != 0, P.mel_low, P.mel_high, h->m.ftab)) { delete h; return nullptr; }
    h->m.tok_class = classify_tokens(P);
    free_host_weights(h->m.host);
    return h;
}

const char *aam_get_name(AprilASRModel model) { return model->m.host.name.c_str(); }
const char *aam_get_description(AprilASRModel model) { return model->m.host.description.c_str(); }
const char *aam_get_language(AprilASRModel model) { return model->m.host.language.c_str(); }
size_t aam_get_sample_rate(AprilASRModel model) { return (size_t)model->m.host.params.sample_rate; }
void aam_free(AprilASRModel model) { if (model) delete model; }

AprilASRSession aas_create_session(AprilASRModel model, AprilConfig config)
{
    if (!model) return nullptr;
    if (config.handler == nullptr) {                   // reference src/april_session.c:81-85
        LOGE("No handler provided! A handler is required, please provide a handler");
        return nullptr;
    }
    Model &m = model->m;
    if (m.engines.empty()) { LOGE("aas: model was loaded host-only (no GPU engine); cannot create sessions"); return nullptr; }
    size_t best = 0;
    for (size_t i = 1; i < m.engines.size(); ++i) if (m.engines[i]->live_slots() < m.engines[best]->live_slots()) best = i;
    const int slot = m.engines[best]->alloc_slot();
    if (slot < 0) { LOGE("aas: no free session slot (APRIL_MAX_SESSIONS)"); return nullptr; }
    AprilASRSession_i *h = new AprilASRSession_i();
    Session &s = h->s;
    s.model = &m; s.eng = m.engines[best]; s.sched = m.scheds[best]; s.slot = slot;
    s.handler = config.handler; s.userdata = config.userdata;
    s.sync_mode = (config.flags & (APRIL_CONFIG_FLAG_ASYNC_RT_BIT | APRIL_CONFIG_FLAG_ASYNC_NO_RT_BIT)) == 0;   // :28
    s.realtime_flag = (config.flags & APRIL_CONFIG_FLAG_ASYNC_RT_BIT) != 0;
    s.fb.shift = m.ftab.shift; s.fb.padded = m.ftab.padded;
    s.fb.seg_count = m.host.params.segment_size; s.fb.seg_step = m.host.params.segment_step;
    s.fb.ring_frames = s.eng->ring_frames();
    s.greedy.init(&m.host.params, &m.tok_class);
    s.sched->attach(&s);
    return h;
}

void aas_feed_pcm16(AprilASRSession session, short *pcm16, size_t short_count)
{
    Session *s = &session->s;
    const short *p = pcm16;
    s->sched->submit(1, &s, &p, &short_count, false, s->sync_mode, /*borrow=*/s->sync_mode);
    if (s->sync_mode) s->sched->deliver_sync_events(s);
}

void aas_flush(AprilASRSession session)
{
    Session *s = &session->s;
    s->sched->submit(1, &s, nullptr, nullptr, true, s->sync_mode);
    if (s->sync_mode) s->sched->deliver_sync_events(s);
}

// reference src/april_session.c:95-97: the EMA of (processing time x 1.1 / audio time) for ASYNC_RT sessions, 1.0 otherwise.
// The reference also feeds it to sonic to compress audio when it exceeds 1; this engine never time-compresses (it batches
// instead), so the value only reports how close the session's GPU is to falling behind real time.
float aas_realtime_get_speedup(AprilASRSession session)
{
    Session *s = &session->s;
    if (!s->realtime_flag) return 1.0f;
    // like the reference (april_session.c:95-97) this only reads the field: no wait, so it is safe inside a result handler
    // (the stepping thread) and never blocks behind a continuously fed stream
    return (float)s->speed_needed.load(std::memory_order_relaxed);
}

void aas_free(AprilASRSession session)
{
    if (!session) return;
    Session *s = &session->s;
    // refused (with an error message) only from inside the session's OWN handler; freeing another, idle session from a
    // handler is a normal free
    if (!s->sched->detach(s)) return;
    s->eng->free_slot(s->slot);
    delete session;
}

// ------------------------------------------------------------------ engine-level ABI
int aprilx_model_dims(AprilASRModel model, AprilxDims *o)
{
    if (!model || !o) return -1;
    const NetDims &d = model->m.layout.dims;
    const ModelParams &P = model->m.host.params;
    o->n_layers = d.n_layers; o->d_model = d.d_model; o->hidden = d.hidden; o->ffn = d.ffn; o->joiner = d.joiner; o->vocab = d.vocab;
    o->mel = d.mel; o->seg = d.seg; o->seg_step = P.segment_step; o->context = d.context;
    o->fft_size = model->m.ftab.padded; o->frame_shift = model->m.ftab.shift; o->sample_rate = P.sample_rate; o->blank_id = P.blank_id;
    o->n_devices = (int)model->m.engines.size();
    o->precision = model->m.engines.empty() ? 0 : model->m.engines[0]->precision();
    o->d_model_file = d.d_norm;
    int64_t n = 0; int cin = 1;
    for (int i = 0; i < 3; ++i) { n += (int64_t)d.conv_ch[i] * cin * 9 + d.conv_ch[i]; cin = d.conv_ch[i]; }
    n += (int64_t)d.embed_in * d.d_model + d.d_model;
    n += (int64_t)d.n_layers * ((int64_t)2 * d.d_model * 4 * d.hidden + 8 * d.hidden + (int64_t)d.hidden * d.d_model + (int64_t)2 * d.d_model * d.ffn + d.ffn + d.d_model + 1);
    n += (int64_t)d.d_model * d.joiner + d.joiner + (int64_t)d.vocab * d.d_model + (int64_t)d.d_model * (d.d_model / d.dec_groups) * d.context;
    n += (int64_t)d.d_model * d.joiner + d.joiner + (int64_t)d.joiner * d.vocab + d.vocab;
    o->param_count = n;
    return 0;
}

const char *aprilx_model_token(AprilASRModel model, int32_t id)
{
    if (!model || id < 0 || id >= model->m.host.params.token_count) return nullptr;
    return model->m.host.params.token((size_t)id);
}

size_t aprilx_model_blob_size(AprilASRModel model)
{
    const std::string meta = make_meta(model->m);
    const size_t woff = (sizeof(BlobHeader) + meta.size() + 255) & ~(size_t)255;
    return woff + model->m.layout.total * 4;
}

int aprilx_model_export_blob(AprilASRModel model, void *dst, size_t dst_size)
{
    HipLegacyLock legacy;
    const std::string meta = make_meta(model->m);
    BlobHeader hd;
    memcpy(hd.magic, "APXBLOB2", 8);
    hd.meta_bytes = meta.size(); hd.weight_floats = model->m.layout.total;
    hd.weights_offset = (sizeof(BlobHeader) + meta.size() + 255) & ~(size_t)255;
    if (dst_size < hd.weights_offset + hd.weight_floats * 4) return -1;
    memset(dst, 0, (size_t)hd.weights_offset);
    memcpy(dst, &hd, sizeof hd);
    memcpy((char *)dst + sizeof hd, meta.data(), meta.size());
    if (model->m.engines.empty()) { memcpy((char *)dst + hd.weights_offset, model->m.host_blob.data(), hd.weight_floats * 4); return 0; }
    Engine *e = model->m.engines[0];
    HIP_CHECK(hipSetDevice(e->device()));
    HIP_CHECK(hipMemcpy((char *)dst + hd.weights_offset, e->weights_device(), hd.weight_floats * 4, hipMemcpyDeviceToHost));
    return 0;
}

int aprilx_model_save_blob(AprilASRModel model, const char *path)
{
    if (!model || !path) return -1;
    std::vector<char> buf(aprilx_model_blob_size(model));
    if (aprilx_model_export_blob(model, buf.data(), buf.size()) != 0) return -1;
    FILE *f = fopen(path, "wb");
    if (!f) { LOGE("aprilx: cannot write '%s'", path); return -1; }
    const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    return (fclose(f) == 0 && ok) ? 0 : -1;
}

// The fp16 cache file (BASELINE configs[4]): same header and metadata, magic "APXBL16B"; the payload walks the blob's float
// index space in order -- MFMA-packed Linear / LSTM matrices as binary16 (round to nearest even, what the engine's fp16
// copies hold), everything else (convolutions, biases, embeddings) as fp32.  Half the size of the fp32 file; loading expands
// it to the fp32 blob whose fp16 copies are bit-identical to those of the original model, so it serves fp16-operand mode only.
int aprilx_model_save_blob_f16(AprilASRModel model, const char *path)
{
    if (!model || !path) return -1;
    std::vector<char> full(aprilx_model_blob_size(model));
    if (aprilx_model_export_blob(model, full.data(), full.size()) != 0) return -1;
    BlobHeader hd; memcpy(&hd, full.data(), sizeof hd);
    const float *w = (const float *)(full.data() + hd.weights_offset);
    std::string out(full.data(), (size_t)hd.weights_offset);
    memcpy(&out[0], "APXBL16B", 8);
    size_t pos = 0;
    auto raw = [&](size_t upto) { if (upto > pos) out.append((const char *)(w + pos), (upto - pos) * 4); pos = upto; };
    for (const auto &sec : gemm_sections(model->m.layout)) {
        raw(sec.first);
        std::vector<_Float16> h(sec.second);
        for (size_t i = 0; i < sec.second; ++i) h[i] = (_Float16)w[sec.first + i];
        out.append((const char *)h.data(), h.size() * 2);
        pos = sec.first + sec.second;
    }
    raw((size_t)hd.weight_floats);
    FILE *f = fopen(path, "wb");
    if (!f) { LOGE("aprilx: cannot write '%s'", path); return -1; }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return (fclose(f) == 0 && ok) ? 0 : -1;
}

AprilASRModel aprilx_model_load_blob(const char *path)
{
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) { LOGE("aprilx: cannot read '%s'", path ? path : "(null)"); return nullptr; }
    std::vector<char> buf;
    if (fseek(f, 0, SEEK_END) == 0) { const long n = ftell(f); if (n > 0) { buf.resize((size_t)n); rewind(f); if (fread(buf.data(), 1, buf.size(), f) != buf.size()) buf.clear(); } }
    fclose(f);
    if (buf.size() >= sizeof(BlobHeader) && memcmp(buf.data(), "APXBL16B", 8) == 0) {
        // expand to the fp32 blob; only the fp16-operand engine may use the rounded matrices
        if (g_inited) {
            const char *pv = getenv("APRIL_PRECISION");
            if (!pv || !(std::string(pv) == "f16" || std::string(pv) == "fp16" || std::string(pv) == "half")) { LOGE("aprilx: '%s' is an fp16 cache file: set APRIL_PRECISION=f16 (the fp32 engine needs the fp32 file)", path); return nullptr; }
        }
        BlobHeader hd; memcpy(&hd, buf.data(), sizeof hd);
        if (hd.weights_offset > buf.size() || hd.meta_bytes > buf.size() || sizeof hd + hd.meta_bytes > hd.weights_offset || hd.weight_floats > ((uint64_t)1 << 34)) { LOGE("aprilx: bad blob"); return nullptr; }
        AprilASRModel_i probe;
        if (!parse_meta(buf.data() + sizeof hd, (size_t)hd.meta_bytes, probe.m) || probe.m.layout.total != hd.weight_floats) { LOGE("aprilx: blob metadata invalid"); return nullptr; }
        std::vector<char> full((size_t)hd.weights_offset + (size_t)hd.weight_floats * 4);
        memcpy(full.data(), buf.data(), (size_t)hd.weights_offset);
        memcpy(full.data(), "APXBLOB2", 8);
        float *w = (float *)(full.data() + hd.weights_offset);
        const char *src = buf.data() + hd.weights_offset, *end = buf.data() + buf.size();
        size_t pos = 0;
        bool ok = true;
        auto raw = [&](size_t upto) { if (upto > pos) { const size_t n = (upto - pos) * 4; if ((size_t)(end - src) < n) { ok = false; return; } memcpy(w + pos, src, n); src += n; } pos = upto; };
        for (const auto &sec : gemm_sections(probe.m.layout)) {
            raw(sec.first);
            if (!ok || (size_t)(end - src) < sec.second * 2) { ok = false; break; }
            const _Float16 *h = (const _Float16 *)src;
            for (size_t i = 0; i < sec.second; ++i) w[sec.first + i] = (float)h[i];
            src += sec.second * 2; pos = sec.first + sec.second;
        }
        if (ok) raw((size_t)hd.weight_floats);
        if (!ok || src != end) { LOGE("aprilx: fp16 blob payload size mismatch"); return nullptr; }
        return aprilx_model_from_blob(full.data(), full.size(), 0);
    }
    return buf.empty() ? nullptr : aprilx_model_from_blob(buf.data(), buf.size(), 0);
}

AprilASRModel aprilx_model_from_blob(const void *blob, size_t size, int blob_is_device_ptr)
{
    // (the legacy-stream lock is taken around this function's own copies and, recursively, by the engines' constructors -- not
    // around build_runtime as a whole: its weight broadcast may wait on RCCL, see broadcast_local)
    // Without an initialised GPU runtime (aam_api_init not called / no device) a HOST blob still
    // yields a host-only model: metadata + packed weights, no engine, no sessions (loader tests,
    // and the gloo leg of the broadcast path on CPU-only machines).
    const bool host_only = !g_inited;
    if (host_only && blob_is_device_ptr) { LOGE("aprilx: device blob without an initialised GPU runtime"); return nullptr; }
    BlobHeader hd;
    if (size < sizeof hd) return nullptr;
    if (!host_only && g_devices.empty()) { LOGE("aprilx: no device selected"); return nullptr; }
    if (blob_is_device_ptr) { HipLegacyLock legacy; HIP_CHECK(hipSetDevice(g_devices[0])); HIP_CHECK(hipMemcpy(&hd, blob, sizeof hd, hipMemcpyDeviceToHost)); }
    else memcpy(&hd, blob, sizeof hd);
    if (memcmp(hd.magic, "APXBLOB2", 8) != 0 || hd.weights_offset > size || hd.weight_floats > (size - hd.weights_offset) / 4 ||
        hd.meta_bytes > size || sizeof hd + hd.meta_bytes > hd.weights_offset) { LOGE("aprilx: bad blob"); return nullptr; }
    std::string meta((size_t)hd.meta_bytes, '\0');
    if (blob_is_device_ptr) { HipLegacyLock legacy; HIP_CHECK(hipMemcpy(&meta[0], (const char *)blob + sizeof hd, meta.size(), hipMemcpyDeviceToHost)); }
    else memcpy(&meta[0], (const char *)blob + sizeof hd, meta.size());
    AprilASRModel_i *h = new AprilASRModel_i();
    if (!parse_meta(meta.data(), meta.size(), h->m) || h->m.layout.total != hd.weight_floats) { LOGE("aprilx: blob metadata invalid"); delete h; return nullptr; }
    const float *w = (const float *)((const char *)blob + hd.weights_offset);
    if (host_only) {
        const ModelParams &P = h->m.host.params;
        h->m.host_blob.assign(w, w + hd.weight_floats);
        if (!build_fbank_tables(P.sample_rate, P.frame_shift_ms, P.frame_length_ms, P.mel_features, P.round_pow2 != 0, P.mel_low, P.mel_high, h->m.ftab)) { delete h; return nullptr; }
        h->m.tok_class = classify_tokens(P);
        return h;
    }
    // a device blob lives on g_devices[0]: engine 0 copies it device-to-device, further devices receive it over RCCL
    if (!build_runtime(h->m, blob_is_device_ptr ? nullptr : w, blob_is_device_ptr ? w : nullptr)) { delete h; return nullptr; }
    return h;
}

// ---- one process per GPU: the model travels from rank 0 to every other rank over RCCL (xGMI inside a node)
int aprilx_broadcast_get_id(void *id_out, size_t cap)
{
    if (!id_out || cap < sizeof(ncclUniqueId)) return -1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) { LOGE("RCCL: ncclGetUniqueId failed"); return -1; }
    memcpy(id_out, &id, sizeof id);
    return (int)sizeof id;
}

AprilASRModel aprilx_model_broadcast(AprilASRModel root_model, int rank, int world, const void *id_bytes)
{
    // The legacy-stream lock (engine.h) is held around this function's allocations and host <-> device copies only; the RCCL
    // bootstrap and the three collectives run WITHOUT it (they wait on the other ranks; two ranks driven from threads of one process
    // would otherwise deadlock on it, and another model's stepping thread would stall behind it: ADVICE r5; broadcast_local has the
    // measurement that makes this safe).
    if (!g_inited || world < 1 || rank < 0 || rank >= world || !id_bytes) { LOGE("aprilx_model_broadcast: bad arguments or library not initialised"); return nullptr; }
    if (rank == 0 && (!root_model || root_model->m.engines.empty())) { LOGE("aprilx_model_broadcast: rank 0 must pass a model that lives on a GPU"); return nullptr; }
    auto fail = [&](const char *what, ncclResult_t r) { LOGE("RCCL: %s failed: %s", what, ncclGetErrorString(r)); return (AprilASRModel) nullptr; };
    ncclUniqueId id; memcpy(&id, id_bytes, sizeof id);
    const int dev = rank == 0 ? root_model->m.engines[0]->device() : g_devices[0];
    HIP_CHECK(hipSetDevice(dev));
    const double t0 = now_ms();
    ncclComm_t comm = nullptr;
    hipStream_t st = nullptr;
    uint64_t *hdr_d = nullptr; char *meta_d = nullptr; float *scratch = nullptr;
    // everything this call allocates is released on every exit path (communicator, stream, staging buffers)
    struct Cleanup {
        ncclComm_t &comm; hipStream_t &st; uint64_t *&hdr_d; char *&meta_d; float *&scratch;
        ~Cleanup() {
            {
                HipLegacyLock legacy;
                if (hdr_d) (void)hipFree(hdr_d);
                if (meta_d) (void)hipFree(meta_d);
                if (scratch) (void)hipFree(scratch);
            }
            if (comm) (void)ncclCommDestroy(comm);
            if (st) (void)hipStreamDestroy(st);
        }
    } cleanup{comm, st, hdr_d, meta_d, scratch};
    ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) { comm = nullptr; return fail("ncclCommInitRank", r); }
    const double t1 = now_ms();
    HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // 1. sizes, 2. metadata (names, PARAMS, token table, dimensions), 3. the packed weights straight into the engine
    std::string meta;
    uint64_t hdr[2] = {0, 0};
    {
        HipLegacyLock legacy;
        HIP_CHECK(hipMalloc((void **)&hdr_d, 16));
        if (rank == 0) { meta = make_meta(root_model->m); hdr[0] = meta.size(); hdr[1] = root_model->m.layout.total; HIP_CHECK(hipMemcpy(hdr_d, hdr, 16, hipMemcpyHostToDevice)); }
    }
    if ((r = ncclBroadcast(hdr_d, hdr_d, 16, ncclUint8, 0, comm, st)) != ncclSuccess) return fail("ncclBroadcast(sizes)", r);
    HIP_CHECK(hipStreamSynchronize(st));
    {
        HipLegacyLock legacy;
        HIP_CHECK(hipMemcpy(hdr, hdr_d, 16, hipMemcpyDeviceToHost));
        if (hdr[0] == 0 || hdr[0] > ((uint64_t)1 << 30) || hdr[1] == 0) { LOGE("aprilx_model_broadcast: implausible sizes"); return nullptr; }
        HIP_CHECK(hipMalloc((void **)&meta_d, (size_t)hdr[0]));
        if (rank == 0) HIP_CHECK(hipMemcpy(meta_d, meta.data(), meta.size(), hipMemcpyHostToDevice));
    }
    if ((r = ncclBroadcast(meta_d, meta_d, (size_t)hdr[0], ncclUint8, 0, comm, st)) != ncclSuccess) return fail("ncclBroadcast(metadata)", r);
    HIP_CHECK(hipStreamSynchronize(st));
    AprilASRModel_i *h = nullptr;
    if (rank == 0) h = root_model;
    else {
        meta.resize((size_t)hdr[0]);
        { HipLegacyLock legacy; HIP_CHECK(hipMemcpy(&meta[0], meta_d, meta.size(), hipMemcpyDeviceToHost)); }
        h = new AprilASRModel_i();
        if (!parse_meta(meta.data(), meta.size(), h->m) || h->m.layout.total != hdr[1] || !create_engines(h->m, nullptr, nullptr)) {
            LOGE("aprilx_model_broadcast: received metadata is invalid"); delete h; h = nullptr;
        }
    }
    // every rank takes part in the weight broadcast even if its model could not be built (a missing rank would hang the others)
    float *buf = h ? h->m.engines[0]->weights_mut() : nullptr;
    if (!buf) { HipLegacyLock legacy; HIP_CHECK(hipMalloc((void **)&scratch, (size_t)hdr[1] * 4)); buf = scratch; }
    const double t2 = now_ms();
    r = ncclBroadcast(buf, buf, (size_t)hdr[1], ncclFloat, 0, comm, st);
    if (r == ncclSuccess) HIP_CHECK(hipStreamSynchronize(st));
    const double t3 = now_ms();
    if (r != ncclSuccess) { if (rank != 0 && h) delete h; return fail("ncclBroadcast(weights)", r); }
    if (!h) return nullptr;
    if (rank != 0 && !distribute_weights(h->m)) { delete h; return nullptr; }
    h->m.load.broadcast_ms = t3 - t2; h->m.load.comm_init_ms = t1 - t0; h->m.load.broadcast_bytes = (size_t)hdr[1] * 4; h->m.load.ranks = world; h->m.load.used_rccl = 1;
    return h;
}

int aprilx_model_load_info(AprilASRModel model, AprilxLoadInfo *out)
{
    if (!model || !out) return -1;
    out->broadcast_ms = model->m.load.broadcast_ms; out->comm_init_ms = model->m.load.comm_init_ms;
    out->broadcast_bytes = (uint64_t)model->m.load.broadcast_bytes; out->ranks = model->m.load.ranks; out->used_rccl = model->m.load.used_rccl;
    return 0;
}

void aprilx_feed_many(size_t n, AprilASRSession *sessions, const short *const *pcm16, const size_t *short_counts)
{
    if (n == 0) return;
    // group by scheduler (GPU); each group is submitted at once so the sessions step together
    std::vector<Scheduler *> scheds;
    for (size_t i = 0; i < n; ++i) { Scheduler *sc = sessions[i]->s.sched; if (std::find(scheds.begin(), scheds.end(), sc) == scheds.end()) scheds.push_back(sc); }
    std::vector<std::vector<Session *>> groups(scheds.size());
    std::vector<std::vector<const short *>> gp(scheds.size());
    std::vector<std::vector<size_t>> gc(scheds.size());
    for (size_t i = 0; i < n; ++i) {
        size_t k = (size_t)(std::find(scheds.begin(), scheds.end(), sessions[i]->s.sched) - scheds.begin());
        groups[k].push_back(&sessions[i]->s); gp[k].push_back(pcm16[i]); gc[k].push_back(short_counts[i]);
    }
    // queue on every GPU first (no wait), then wait, so GPUs run concurrently
    for (size_t k = 0; k < scheds.size(); ++k) scheds[k]->submit((int)groups[k].size(), groups[k].data(), gp[k].data(), gc[k].data(), false, false, /*borrow=*/true);
    for (size_t k = 0; k < scheds.size(); ++k) scheds[k]->wait_idle_many(groups[k].data(), (int)groups[k].size());
    for (size_t i = 0; i < n; ++i) if (sessions[i]->s.sync_mode) sessions[i]->s.sched->deliver_sync_events(&sessions[i]->s);
}

void aprilx_feed_many_pipelined(size_t n, AprilASRSession *sessions, const short *const *pcm16, const size_t *short_counts, int depth)
{
    if (n == 0) return;
    if (depth < 1) depth = 1;
    std::vector<Scheduler *> scheds;
    for (size_t i = 0; i < n; ++i) { Scheduler *sc = sessions[i]->s.sched; if (std::find(scheds.begin(), scheds.end(), sc) == scheds.end()) scheds.push_back(sc); }
    std::vector<std::vector<Session *>> groups(scheds.size());
    std::vector<std::vector<const short *>> gp(scheds.size());
    std::vector<std::vector<size_t>> gc(scheds.size());
    for (size_t i = 0; i < n; ++i) {
        size_t k = (size_t)(std::find(scheds.begin(), scheds.end(), sessions[i]->s.sched) - scheds.begin());
        groups[k].push_back(&sessions[i]->s); gp[k].push_back(pcm16[i]); gc[k].push_back(short_counts[i]);
    }
    // the samples are copied into the sessions' queues (the caller may reuse its buffers at once), every GPU first, then the wait
    for (size_t k = 0; k < scheds.size(); ++k) scheds[k]->submit((int)groups[k].size(), groups[k].data(), gp[k].data(), gc[k].data(), false, false, /*borrow=*/false);
    for (size_t k = 0; k < scheds.size(); ++k) scheds[k]->wait_backlog(groups[k].data(), (int)groups[k].size(), (uint64_t)(depth - 1));
    for (size_t i = 0; i < n; ++i) if (sessions[i]->s.sync_mode) sessions[i]->s.sched->deliver_sync_events(&sessions[i]->s);
}

void aprilx_drain_many(size_t n, AprilASRSession *sessions)
{
    if (n == 0) return;
    std::vector<Scheduler *> scheds;
    for (size_t i = 0; i < n; ++i) { Scheduler *sc = sessions[i]->s.sched; if (std::find(scheds.begin(), scheds.end(), sc) == scheds.end()) scheds.push_back(sc); }
    for (Scheduler *sc : scheds) {
        std::vector<Session *> g;
        for (size_t i = 0; i < n; ++i) if (sessions[i]->s.sched == sc) g.push_back(&sessions[i]->s);
        sc->wait_backlog(g.data(), (int)g.size(), 0);
    }
    for (size_t i = 0; i < n; ++i) if (sessions[i]->s.sync_mode) sessions[i]->s.sched->deliver_sync_events(&sessions[i]->s);
}

void aprilx_flush_many(size_t n, AprilASRSession *sessions)
{
    for (size_t i = 0; i < n; ++i) { Session *s = &sessions[i]->s; s->sched->submit(1, &s, nullptr, nullptr, true, false); }
    for (size_t i = 0; i < n; ++i) aprilx_session_drain(sessions[i]);
    for (size_t i = 0; i < n; ++i) if (sessions[i]->s.sync_mode) sessions[i]->s.sched->deliver_sync_events(&sessions[i]->s);
}

void aprilx_session_drain(AprilASRSession session)
{
    Session *s = &session->s;
    s->sched->submit(0, nullptr, nullptr, nullptr, false, false);   // wake-up only
    s->sched->wait_idle(s);
}

int aprilx_run_encoder(AprilASRModel model, int n, const float *x, const float *h, const float *c, float *eout, float *h2, float *c2)
{
    if (!model || n <= 0 || model->m.engines.empty()) return -1;
    if (n > model->m.engines[0]->max_batch()) { LOGE("aprilx_run_encoder: n = %d exceeds APRIL_MAX_BATCH = %d", n, model->m.engines[0]->max_batch()); return -1; }
    model->m.engines[0]->debug_encoder(n, x, h, c, eout, h2, c2);
    return 0;
}
int aprilx_run_decoder(AprilASRModel model, int n, const int64_t *context, float *dout)
{
    if (!model || n <= 0 || model->m.engines.empty() || n > model->m.engines[0]->max_slots()) return -1;
    model->m.engines[0]->debug_decoder(n, context, dout);
    return 0;
}
int aprilx_run_joiner(AprilASRModel model, int n, const float *eout, const float *dout, float *logits)
{
    if (!model || n <= 0 || model->m.engines.empty() || n > model->m.engines[0]->max_slots()) return -1;
    model->m.engines[0]->debug_joiner(n, eout, dout, logits);
    return 0;
}
int aprilx_run_decide(AprilASRModel model, int n, int op, const float *logits, float early_emit, const int32_t *now_ms, int round, int32_t *state_io, void *records_out)
{
    if (!model || n <= 0 || model->m.engines.empty() || n > model->m.engines[0]->max_slots() || !state_io) return -1;
    if (op == 0 && (!logits || !now_ms || !records_out || round < 0 || round > 2)) return -1;
    model->m.engines[0]->debug_decide(n, op, logits, early_emit, (const int *)now_ms, round, state_io, (StepRecord *)records_out);
    return 0;
}
int aprilx_plan_gemm(int M, int N, int kz, int zcount, int tile_ok, int force, int32_t *out)
{
    if (!out || M <= 0 || N <= 0 || kz <= 0 || zcount <= 0) return -1;
    out[0] = gemm_fullk(M, N, kz, force != 0, zcount, tile_ok) ? 1 : 0;
    out[1] = gemm_partials(M, N, kz, zcount, tile_ok);
    out[2] = gemm_tile_planned(M, N, kz, zcount) ? 1 : 0;
    return 0;
}
int aprilx_stream_form(int kind, int M, int N, int K, int kz, int groups)
{
    if (kind < 0 || kind > 5 || M <= 0 || N <= 0 || K <= 0 || kz <= 0) return -1;
    static float dummy[4];
    static int idummy[4];
    GemmArgs g;                                       // only which pointers are set matters to recur_form
    g.M = M; g.N = N; g.K = K; g.kz = kz; g.wp = dummy; g.out = dummy; g.bias = dummy; g.a0 = dummy; g.K0 = K;
    RowScale rs; rs.ssq = dummy; rs.groups = groups; rs.inv_n = 1.0f;
    switch (kind) {
    case 0: g.epi = EPI_LSTM; g.K0 = K / 2; g.K1 = K / 2; g.a1 = dummy; g.x_scale = rs; g.c_state = dummy; g.slot_idx = idummy; break;
    case 1: g.epi = EPI_LSTM; g.K0 = K / 2; g.K1 = K / 2; g.a1 = dummy; g.wave_mask = 0xC; g.p_add = dummy; g.c_state = dummy; g.slot_idx = idummy; break;
    case 2: g.epi = EPI_XPART; g.K0 = K / 2; g.K1 = K / 2; g.a1 = dummy; g.wave_mask = 0x3; g.x_scale = rs; break;
    case 3: g.epi = EPI_BIAS_DSWISH; break;
    case 4: g.epi = EPI_HR; g.r_scale = rs; g.state = dummy; g.resid = dummy; g.slot_idx = idummy; break;
    default: g.epi = EPI_RESID_SSQ; g.resid = dummy; g.ssq_out = dummy; break;
    }
    return recur_form(g);
}
int aprilx_run_fbank(AprilASRModel model, int n_frames, const int16_t *pcm_frames, float *out)
{
    if (!model || model->m.engines.empty() || n_frames <= 0 || n_frames > model->m.engines[0]->ring_frames()) return -1;
    model->m.engines[0]->debug_fbank(n_frames, pcm_frames, out);
    return 0;
}

void aprilx_session_trace_logits(AprilASRSession session, float *buf, size_t cap_floats, size_t *used_floats)
{
    Session *s = &session->s;
    s->sched->wait_idle(s);
    s->trace_buf = buf; s->trace_cap = cap_floats; s->trace_used = used_floats;
}

uint64_t aprilx_session_chunks(AprilASRSession session) { return session->s.chunks; }

uint64_t aprilx_session_read_frames(AprilASRSession session, uint64_t first, int n, float *out)
{
    Session *s = &session->s;
    s->sched->wait_idle(s);
    const uint64_t total = s->fb.rows_written;
    if (out && n > 0) {
        const int R = s->eng->ring_frames();
        if (first + (uint64_t)n > total) return total;                      // (part of) the range has not been written yet: nothing copied, the count as before
        if (total - first > (uint64_t)R) return UINT64_MAX;                 // the first rows of the range have been overwritten: nothing copied
        int done = 0;
        while (done < n) {          // (the ring may wrap inside the range)
            const int row = (int)((first + (uint64_t)done) % (uint64_t)R), cnt = std::min(n - done, R - row);
            s->eng->read_ring(s->slot, row, cnt, out + (size_t)done * s->eng->dims().mel);
            done += cnt;
        }
    }
    return total;
}

void aprilx_session_context(AprilASRSession session, int32_t *host_ctx, int32_t *device_state)
{
    Session *s = &session->s;
    s->sched->wait_idle(s);
    if (host_ctx) { host_ctx[0] = s->greedy.ctx[0]; host_ctx[1] = s->greedy.ctx[1]; }
    if (device_state) {
        GreedyState g;
        s->eng->read_greedy_state(s->slot, &g);
        device_state[0] = g.ctx0; device_state[1] = g.ctx1; device_state[2] = g.last_tok; device_state[3] = (int32_t)g.last_emit_ms;
    }
}

void aprilx_model_stats(AprilASRModel model, int device_index, AprilxStats *out)
{
    memset(out, 0, sizeof *out);
    if (!model || device_index < 0 || device_index >= (int)model->m.scheds.size()) return;
    const SchedStats st = model->m.scheds[(size_t)device_index]->stats();
    out->ticks = st.ticks; out->steps = st.steps; out->chunks = st.chunks; out->rounds = st.rounds; out->frames = st.frames; out->max_batch_seen = st.max_batch_seen;
    for (int i = 0; i < 8; ++i) out->host_ms[i] = st.host_ms[i];
    out->flights = st.flights; out->replay_mismatch = st.replay_mismatch; out->lm_steps = st.lm_steps; out->lm_chunks = st.lm_chunks; out->wave_steps = st.wave_steps; out->wave_chunks = st.wave_chunks;
    Engine *e = model->m.engines[(size_t)device_index];
    out->kernels_per_step = (uint64_t)e->kernels_per_step();
    for (int i = 0; i < 6; ++i) { out->kernel_ms[i] = e->timing(i).ms; out->kernel_launches[i] = (uint64_t)e->timing(i).launches; }
    double gms = 0; long gl = 0, gr = 0;
    e->gates_clock(&gms, &gl, &gr);
    out->gates_clock_ms = gms; out->gates_clock_launches = (uint64_t)gl; out->gates_clock_rows = (uint64_t)gr;
    double ms4[4]; long l4[4];
    e->gates_clock_by_n(ms4, l4);
    for (int i = 0; i < 4; ++i) { out->gates_clock_ms_by_n[i] = ms4[i]; out->gates_clock_launches_by_n[i] = (uint64_t)l4[i]; }
}

int aprilx_model_feed_latency(AprilASRModel model, int device_index, double *out_ms, int cap, int reset)
{
    if (!model || device_index < 0 || device_index >= (int)model->m.scheds.size()) return 0;
    return (int)model->m.scheds[(size_t)device_index]->latencies(out_ms, cap > 0 ? (size_t)cap : 0, reset != 0);
}

void aprilx_model_profile(AprilASRModel model, int enable)
{
    for (Engine *e : model->m.engines) {
        e->set_profiling(enable == 1); if (enable == 1) e->reset_timing();
        e->set_gates_clock(enable == 2);
    }
}

struct AprilxGreedy_i {
    Greedy g; AprilRecognitionResultHandler handler; void *ud; std::vector<Event> ev;
    void flush_events() { for (auto &e : ev) handler(ud, (AprilResultType)e.type, e.tokens.size(), e.tokens.empty() ? nullptr : e.tokens.data()); ev.clear(); }
};

AprilxGreedy aprilx_greedy_create(AprilASRModel model, AprilRecognitionResultHandler handler, void *userdata)
{
    if (!model || !handler) return nullptr;
    AprilxGreedy_i *g = new AprilxGreedy_i();
    g->g.init(&model->m.host.params, &model->m.tok_class);
    g->g.reset_context_to_blank();
    g->handler = handler; g->ud = userdata;
    return g;
}
int aprilx_greedy_step(AprilxGreedy g, int32_t idx, float max_val, float blank_val, float early_emit, size_t now_ms, int32_t *ctx_out)
{
    JointResult r{idx, max_val, blank_val};
    const bool blank = g->g.on_joint(r, early_emit, now_ms, g->ev);
    g->flush_events();
    if (ctx_out) { ctx_out[0] = g->g.ctx[0]; ctx_out[1] = g->g.ctx[1]; }
    return blank ? 1 : 0;
}
void aprilx_greedy_finish(AprilxGreedy g) { g->g.finish_flush(g->ev); g->flush_events(); }
void aprilx_greedy_free(AprilxGreedy g) { delete g; }

// A handler implemented in C for load generators: userdata -> uint64_t[6] {calls, partial, final, cant_keep_up, silence, tokens}
void aprilx_counting_handler(void *userdata, AprilResultType type, size_t count, const AprilToken *tokens)
{
    (void)tokens;
    uint64_t *c = (uint64_t *)userdata;
    if (!c) return;
    __atomic_fetch_add(&c[0], 1, __ATOMIC_RELAXED);
    if ((int)type >= 1 && (int)type <= 4) __atomic_fetch_add(&c[(int)type], 1, __ATOMIC_RELAXED);
    __atomic_fetch_add(&c[5], (uint64_t)count, __ATOMIC_RELAXED);
}

int aprilx_model_fbank_tables(AprilASRModel model, float *window, float *mel)
{
    if (!model) return -1;
    const FbankHostTables &t = model->m.ftab;
    if (window) memcpy(window, t.window.data(), t.window.size() * 4);
    if (mel) memcpy(mel, t.mel.data(), t.mel.size() * 4);
    return t.padded;
}

int aprilx_probe_file(const char *path, char *err, size_t err_cap)
{
    FILE *fd = fopen(path, "rb");
    std::string e;
    if (!fd) e = "cannot open file";
    else {
        fseek(fd, 0, SEEK_END); long sz = ftell(fd); fseek(fd, 0, SEEK_SET);
        std::vector<uint8_t> blob(sz > 0 ? (size_t)sz : 0);
        size_t got = blob.empty() ? 0 : fread(blob.data(), 1, blob.size(), fd);
        fclose(fd);
        ContainerInfo info;
        if (got != blob.size()) e = "short read";
        else if (parse_container(blob, info, e)) { if (err && err_cap) err[0] = 0; return 0; }
    }
    if (err && err_cap) { strncpy(err, e.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
    return -1;
}

}  // extern "C"
