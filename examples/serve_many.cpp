// Many concurrent streams from one thread, the way a server front end would drive this library: N sessions are fed 100 ms at a
// time through the engine ABI's pipelined group feed (include/aprilx_engine.h aprilx_feed_many_pipelined, depth 2: the call for
// feed k + 1 returns when feed k is complete, so the library prepares and launches a feed while the GPU still works on the one
// before it).  The reference has no counterpart: its sessions are fed one by one (example.cpp) and each runs its own ONNX graphs.
//
//   g++ -O2 -std=c++17 examples/serve_many.cpp -I include -L april_asr_amd -laprilasr -Wl,-rpath,$PWD/april_asr_amd -o serve_many
//   ./serve_many model.april audio.raw [sessions=64] [mode=pipelined|lockstep]
//
// Every session gets the same PCM16 file, rotated by (session index x 0.37 s) so that the streams differ.  Prints one line per
// session -- "<index> <callbacks> <final results> <tokens in final results> <text of the last final result>" -- and the wall time.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "april_api.h"
#include "aprilx_engine.h"

struct Stream { size_t calls = 0, finals = 0, final_tokens = 0; std::string last_final; };

static void on_result(void *ud, AprilResultType type, size_t count, const AprilToken *tokens)
{
    Stream *s = static_cast<Stream *>(ud);
    s->calls++;
    if (type != APRIL_RESULT_RECOGNITION_FINAL) return;
    s->finals++; s->final_tokens += count;
    s->last_final.clear();
    for (size_t i = 0; i < count; ++i) s->last_final += tokens[i].token;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <model.april> <audio.raw (PCM16 mono)> [sessions=64] [pipelined|lockstep]\n", argv[0]); return 2; }
    const int n = argc > 3 ? atoi(argv[3]) : 64;
    const bool pipelined = !(argc > 4 && !strcmp(argv[4], "lockstep"));
    aam_api_init(APRIL_VERSION);
    AprilASRModel model = aam_create_model(argv[1]);
    if (!model) { fprintf(stderr, "failed to load model %s\n", argv[1]); return 1; }
    FILE *f = fopen(argv[2], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    std::vector<short> pcm;
    { short buf[4096]; size_t got; while ((got = fread(buf, sizeof(short), 4096, f)) > 0) pcm.insert(pcm.end(), buf, buf + got); }
    fclose(f);
    const size_t step = aam_get_sample_rate(model) / 10;                 // 100 ms
    const size_t steps = pcm.size() / step;
    if (!steps) { fprintf(stderr, "audio shorter than one feed\n"); return 1; }

    std::vector<Stream> streams((size_t)n);
    std::vector<AprilASRSession> sessions((size_t)n);
    std::vector<std::vector<short>> audio((size_t)n);
    for (int i = 0; i < n; ++i) {
        AprilConfig cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.handler = on_result; cfg.userdata = &streams[(size_t)i];
        cfg.flags = APRIL_CONFIG_FLAG_ZERO_BIT;                           // synchronous: handlers run on this thread, inside the feed calls
        sessions[(size_t)i] = aas_create_session(model, cfg);
        if (!sessions[(size_t)i]) { fprintf(stderr, "failed to create session %d\n", i); return 1; }
        const size_t rot = ((size_t)i * (size_t)(0.37 * aam_get_sample_rate(model))) % pcm.size();
        audio[(size_t)i].assign(pcm.begin() + (long)rot, pcm.end());
        audio[(size_t)i].insert(audio[(size_t)i].end(), pcm.begin(), pcm.begin() + (long)rot);
    }
    std::vector<const short *> ptrs((size_t)n);
    std::vector<size_t> counts((size_t)n, step);
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t k = 0; k < steps; ++k) {
        for (int i = 0; i < n; ++i) ptrs[(size_t)i] = audio[(size_t)i].data() + k * step;
        if (pipelined) aprilx_feed_many_pipelined((size_t)n, sessions.data(), ptrs.data(), counts.data(), 2);
        else aprilx_feed_many((size_t)n, sessions.data(), ptrs.data(), counts.data());
    }
    if (pipelined) aprilx_drain_many((size_t)n, sessions.data());
    aprilx_flush_many((size_t)n, sessions.data());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n; ++i) printf("%d %zu %zu %zu %s\n", i, streams[(size_t)i].calls, streams[(size_t)i].finals, streams[(size_t)i].final_tokens, streams[(size_t)i].last_final.c_str());
    fprintf(stderr, "%d streams x %.1f s of audio in %.1f ms (%s feed): %.0f audio-seconds per second\n", n, steps * 0.1, ms, pipelined ? "pipelined" : "lock-step",
            n * steps * 0.1 / (ms * 1e-3));
    for (AprilASRSession s : sessions) aas_free(s);
    aam_free(model);
    return 0;
}
