// Drives the REAL host runtime (csrc/april_api.cc + session.cc + host_pool.h + loader) against tests/sched_harness/fake_engine.cc
// through the C ABI only, in the scenarios of tests/test_gpu_pipeline.py / test_gpu_concurrency.py: lock-step and pipelined group
// feeds, asynchronous sessions, many client threads with their own synchronous sessions, sessions created and freed while others
// stream, frees from inside result handlers, queue overflow, irregular feed sizes, a long feed beside short ones.  Built with
// -fsanitize=thread and with -fsanitize=address,undefined by tests/test_sched_sanitizers.py; exits non-zero on any inconsistency
// (the sanitizers abort on their own findings).
// usage: driver model.april
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/april_api.h"
#include "../../include/aprilx_engine.h"

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "CHECK failed at %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); ++g_fail; } } while (0)

struct Sess {
    AprilASRSession h = nullptr;
    uint64_t digest = 1469598103934665603ull;
    int calls = 0, tokens = 0, cant_keep_up = 0;
    std::vector<short> pcm;
    void mix(const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; ++i) { digest ^= b[i]; digest *= 1099511628211ull; } }
};

static void handler(void *ud, AprilResultType type, size_t count, const AprilToken *toks)
{
    Sess *s = (Sess *)ud;
    if (type == APRIL_RESULT_ERROR_CANT_KEEP_UP) { s->cant_keep_up++; return; }
    s->calls++;
    int t = (int)type; s->mix(&t, sizeof t);
    for (size_t i = 0; i < count; ++i) {
        s->mix(toks[i].token, strlen(toks[i].token));
        s->mix(&toks[i].logprob, sizeof(float));
        int fl = (int)toks[i].flags; s->mix(&fl, sizeof fl);
        uint64_t ms = toks[i].time_ms; s->mix(&ms, sizeof ms);
        s->tokens++;
    }
}

static std::vector<short> lcg_pcm(size_t n, unsigned seed)
{
    std::vector<short> v(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (auto &x : v) { s = s * 1664525u + 1013904223u; x = (short)((int)((s >> 16) & 0x3fff) - 8192); }
    return v;
}

static Sess *make(AprilASRModel m, unsigned seed, size_t samples, int flags)
{
    Sess *s = new Sess();
    s->pcm = lcg_pcm(samples, seed);
    AprilConfig cfg; memset(&cfg, 0, sizeof cfg);
    cfg.handler = handler; cfg.userdata = s; cfg.flags = (AprilConfigFlagBits)flags;
    s->h = aas_create_session(m, cfg);
    if (!s->h) { fprintf(stderr, "aas_create_session failed\n"); exit(2); }
    return s;
}

// one pass over nsess sessions x steps feeds of `feed` samples in an ingest mode; returns the per-session digests
enum Mode { LOCKSTEP, PIPE2, ASYNC_PIPE2, THREADS };
static std::vector<uint64_t> stream(AprilASRModel m, int nsess, int steps, size_t feed, Mode mode, int *calls_out = nullptr)
{
    std::vector<Sess *> ss;
    for (int i = 0; i < nsess; ++i) ss.push_back(make(m, 4242u + (unsigned)i, feed * (size_t)steps, mode == ASYNC_PIPE2 ? APRIL_CONFIG_FLAG_ASYNC_NO_RT_BIT : 0));
    std::vector<AprilASRSession> hs; for (Sess *s : ss) hs.push_back(s->h);
    std::vector<const short *> ptr((size_t)nsess); std::vector<size_t> cnt((size_t)nsess, feed);
    if (mode == THREADS) {
        const int nthreads = 8;
        std::vector<std::thread> th;
        // a monitoring thread beside the clients: statistics, latencies, the speed-up estimate, idle-session readers
        std::atomic<bool> mon_stop{false};
        Sess *idle = make(m, 99u, 3200, 0);
        aas_feed_pcm16(idle->h, idle->pcm.data(), 3200);
        std::thread mon([&] {
            double lat[32]; float rows[80 * 4];
            while (!mon_stop.load()) {
                AprilxStats st; aprilx_model_stats(m, 0, &st);
                (void)aprilx_model_feed_latency(m, 0, lat, 32, 0);
                for (int i = 0; i < nsess; i += 7) (void)aas_realtime_get_speedup(ss[(size_t)i]->h);
                int32_t hc[2], dc[4];
                aprilx_session_context(idle->h, hc, dc);                     // (an idle session's state, while others are stepped)
                (void)aprilx_session_read_frames(idle->h, 0, 4, rows);
                (void)aprilx_session_chunks(idle->h);
            }
        });
        for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t] {
            for (int k = 0; k < steps; ++k)
                for (int i = t; i < nsess; i += nthreads) aas_feed_pcm16(ss[(size_t)i]->h, ss[(size_t)i]->pcm.data() + (size_t)k * feed, feed);
            for (int i = t; i < nsess; i += nthreads) aas_flush(ss[(size_t)i]->h);
        });
        for (auto &t : th) t.join();
        mon_stop.store(true); mon.join();
        aas_free(idle->h); delete idle;
    } else {
        for (int k = 0; k < steps; ++k) {
            for (int i = 0; i < nsess; ++i) ptr[(size_t)i] = ss[(size_t)i]->pcm.data() + (size_t)k * feed;
            if (mode == LOCKSTEP) aprilx_feed_many((size_t)nsess, hs.data(), ptr.data(), cnt.data());
            else aprilx_feed_many_pipelined((size_t)nsess, hs.data(), ptr.data(), cnt.data(), 2);
        }
        aprilx_drain_many((size_t)nsess, hs.data());
        aprilx_flush_many((size_t)nsess, hs.data());
        aprilx_drain_many((size_t)nsess, hs.data());
    }
    std::vector<uint64_t> d;
    int calls = 0;
    for (Sess *s : ss) { d.push_back(s->digest); calls += s->calls; CHECK(s->cant_keep_up == 0, "unexpected CANT_KEEP_UP"); }
    if (calls_out) *calls_out = calls;
    for (Sess *s : ss) { aas_free(s->h); delete s; }
    return d;
}

// ---- frees from inside handlers: a session's own handler may not free it (refused, logged); it may free another, idle session
struct FreeCtx { AprilASRSession self = nullptr, other = nullptr; std::atomic<int> tried{0}; };
static void freeing_handler(void *ud, AprilResultType, size_t, const AprilToken *)
{
    FreeCtx *c = (FreeCtx *)ud;
    if (c->tried.exchange(1) == 0) {
        aas_free(c->self);                       // refused: must return without freeing (and without deadlock)
        if (c->other) { aas_free(c->other); c->other = nullptr; }      // allowed: another, idle session
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: driver model.april\n"); return 2; }
    aam_api_init(APRIL_VERSION);
    AprilASRModel m = aam_create_model(argv[1]);
    if (!m) { fprintf(stderr, "model load failed\n"); return 2; }
    const int NS = 40, STEPS = 24;

    // 1. the same sessions through every ingest mode: identical callbacks
    int calls = 0;
    const std::vector<uint64_t> ref = stream(m, NS, STEPS, 1600, LOCKSTEP, &calls);
    CHECK(calls >= NS, "only %d callbacks in the reference pass", calls);
    const std::vector<uint64_t> a = stream(m, NS, STEPS, 1600, PIPE2), b = stream(m, NS, STEPS, 1600, ASYNC_PIPE2), c = stream(m, NS, STEPS, 1600, THREADS);
    CHECK(a == ref, "pipelined group feed: callbacks differ from the lock-step feed");
    CHECK(b == ref, "asynchronous sessions: callbacks differ from the lock-step feed");
    CHECK(c == ref, "client threads with their own sessions: callbacks differ from the lock-step feed");
    // irregular feeds (not a whole number of frames) and half-second feeds (several chunk steps / the layer-major path per flight)
    CHECK(stream(m, 9, 40, 1234, LOCKSTEP) == stream(m, 9, 40, 1234, PIPE2), "irregular feeds: modes differ");
    CHECK(stream(m, 12, 6, 8000, LOCKSTEP) == stream(m, 12, 6, 8000, ASYNC_PIPE2), "half-second feeds: modes differ");
    CHECK(stream(m, 3, 1, 80000, LOCKSTEP) == stream(m, 3, 1, 80000, PIPE2), "five seconds in one feed: modes differ");

    // 2. churn: sessions are created, fed and freed on two threads while four others stream their own sessions
    {
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        std::vector<uint64_t> got(4 * 5, 0);
        for (int t = 0; t < 4; ++t) th.emplace_back([&, t] {
            std::vector<Sess *> mine;
            for (int i = 0; i < 5; ++i) mine.push_back(make(m, 4242u + (unsigned)(t * 5 + i), 1600 * (size_t)STEPS, 0));
            for (int k = 0; k < STEPS; ++k) for (Sess *s : mine) aas_feed_pcm16(s->h, s->pcm.data() + (size_t)k * 1600, 1600);
            for (Sess *s : mine) aas_flush(s->h);
            for (int i = 0; i < 5; ++i) { got[(size_t)(t * 5 + i)] = mine[(size_t)i]->digest; aas_free(mine[(size_t)i]->h); delete mine[(size_t)i]; }
        });
        for (int t = 0; t < 2; ++t) th.emplace_back([&, t] {
            unsigned n = 0;
            while (!stop.load()) {
                ++n;
                Sess *s = make(m, 9000u + (unsigned)t * 1000u + n, 4800, (n & 1) ? APRIL_CONFIG_FLAG_ASYNC_NO_RT_BIT : 0);
                aas_feed_pcm16(s->h, s->pcm.data(), 3200);
                if (n % 3 == 0) aas_flush(s->h);
                aas_free(s->h);                     // (an asynchronous session may still have work queued: aas_free waits for it)
                delete s;
            }
        });
        for (int t = 0; t < 4; ++t) th[(size_t)t].join();
        stop.store(true);
        for (size_t t = 4; t < th.size(); ++t) th[t].join();
        for (int i = 0; i < 20; ++i) CHECK(got[(size_t)i] == ref[(size_t)i], "churn: session %d differs from the reference run", i);
    }

    // 2b. a second model is loaded, used and freed while client threads stream on the first (engine construction / destruction -- device
    // allocations, copies on the legacy stream, table builds -- beside the first model's graph captures and launches)
    {
        std::vector<std::thread> th;
        std::vector<uint64_t> got(12, 0);
        for (int t = 0; t < 3; ++t) th.emplace_back([&, t] {
            std::vector<Sess *> mine;
            for (int i = 0; i < 4; ++i) mine.push_back(make(m, 4242u + (unsigned)(t * 4 + i), 1600 * (size_t)STEPS, 0));
            for (int k = 0; k < STEPS; ++k) for (Sess *s : mine) aas_feed_pcm16(s->h, s->pcm.data() + (size_t)k * 1600, 1600);
            for (Sess *s : mine) aas_flush(s->h);
            for (int i = 0; i < 4; ++i) { got[(size_t)(t * 4 + i)] = mine[(size_t)i]->digest; aas_free(mine[(size_t)i]->h); delete mine[(size_t)i]; }
        });
        std::vector<uint64_t> second;
        th.emplace_back([&] {
            for (int rep = 0; rep < 2; ++rep) {
                AprilASRModel m2 = aam_create_model(argv[1]);
                if (!m2) { fprintf(stderr, "second model failed to load\n"); ++g_fail; return; }
                second = stream(m2, 6, 8, 1600, rep ? PIPE2 : LOCKSTEP);
                aam_free(m2);
            }
        });
        for (auto &t : th) t.join();
        for (int i = 0; i < 12; ++i) CHECK(got[(size_t)i] == ref[(size_t)i], "second model beside the first: session %d of the first model differs", i);
        const std::vector<uint64_t> alone = stream(m, 6, 8, 1600, LOCKSTEP);
        CHECK(second == alone, "the second model's sessions differ from the same sessions on the first model");
    }

    // 3. frees from inside a handler
    {
        FreeCtx ctx;
        Sess *other = make(m, 7u, 1600, 0);
        AprilConfig cfg; memset(&cfg, 0, sizeof cfg);
        cfg.handler = freeing_handler; cfg.userdata = &ctx; cfg.flags = APRIL_CONFIG_FLAG_ASYNC_NO_RT_BIT;
        AprilASRSession self = aas_create_session(m, cfg);
        ctx.self = self; ctx.other = other->h;
        std::vector<short> pcm = lcg_pcm(16000 * 3, 5u);
        for (int k = 0; k < 30 && !ctx.tried.load(); ++k) { aas_feed_pcm16(self, pcm.data() + (size_t)k * 1600, 1600); aprilx_session_drain(self); }
        aas_flush(self); aprilx_session_drain(self);
        CHECK(ctx.tried.load() == 1, "the freeing handler never ran");
        CHECK(ctx.other == nullptr, "the handler did not free the other session");
        aas_free(self);                              // outside the handler: a normal free
        delete other;                                // (its session handle was freed by the handler)
    }

    // 4. queue overflow of an asynchronous session: a push that would make the queue hold >= 48000 samples is refused on the caller thread
    {
        Sess *s = make(m, 11u, 60000, APRIL_CONFIG_FLAG_ASYNC_NO_RT_BIT);
        aas_feed_pcm16(s->h, s->pcm.data(), 48000);
        CHECK(s->cant_keep_up == 1, "48000 samples in one push: expected CANT_KEEP_UP, got %d", s->cant_keep_up);
        aas_feed_pcm16(s->h, s->pcm.data(), 47999);
        aprilx_session_drain(s->h);
        CHECK(s->cant_keep_up == 1, "47999 samples must be accepted");
        aas_free(s->h); delete s;
    }

    AprilxStats st; memset(&st, 0, sizeof st);
    for (int dev = 0; dev < 8; ++dev) {                       // every engine of the model (APRIL_GPU_DEVICES=0,0,0: three on the one fake device)
        AprilxStats one; aprilx_model_stats(m, dev, &one);
        st.replay_mismatch += one.replay_mismatch; st.chunks += one.chunks; st.flights += one.flights; st.steps += one.steps; st.lm_steps += one.lm_steps; st.wave_steps += one.wave_steps;
    }
    CHECK(st.replay_mismatch == 0, "replay_mismatch = %llu: the scheduler read records of a flight that had not completed, or replayed them out of order", (unsigned long long)st.replay_mismatch);
    CHECK(st.chunks > 0 && st.flights > 0, "nothing ran");
    double lat[64]; const int nl = aprilx_model_feed_latency(m, 0, lat, 64, 0);
    CHECK(nl > 0 && lat[0] > 0.0, "no feed latencies recorded");
    printf("HARNESS %s: %llu chunks, %llu flights, %llu steps (%llu layer-major / %llu wavefront), %d callbacks in the reference pass\n", g_fail ? "FAILED" : "ok",
           (unsigned long long)st.chunks, (unsigned long long)st.flights, (unsigned long long)st.steps, (unsigned long long)st.lm_steps, (unsigned long long)st.wave_steps, calls);
    aam_free(m);
    return g_fail ? 1 : 0;
}
