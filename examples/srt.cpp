// SubRip (.srt) writer against the C ABI in include/april_api.h -- the role of the reference's example_srt.cpp
// (`./srt file.wav model.april`): every FINAL result becomes one subtitle per token, running from the token's time to the
// next token's (the last one lasts 2 s), and showing the text up to and including that token.  Uses the reference's entry
// points only; the token times come from AprilToken.time_ms (40 ms per chunk, reference src/april_session.c:442-443).
//
//   g++ -O2 -std=c++17 examples/srt.cpp -I include -L april_asr_amd -laprilasr -Wl,-rpath,$PWD/april_asr_amd -o srt
//   ./srt audio.wav model.april > audio.srt        (16-bit mono PCM WAV or raw PCM16 at the model's sample rate)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "april_api.h"

static int g_cue = 0;

// hh:mm:ss,mmm the way the reference's example computes it: units are peeled off while the remainder EXCEEDS one unit, so
// an exact multiple keeps a full unit in the next field (60000 ms prints as 00:00:59,1000) -- kept for identical output
static void stamp(size_t ms, char *out, size_t cap)
{
    auto peel = [&](size_t unit) { const size_t n = ms > unit ? (ms - 1) / unit : 0; ms -= n * unit; return (int)n; };
    const int h = peel(3600 * 1000), m = peel(60 * 1000), s = peel(1000);
    snprintf(out, cap, "%02d:%02d:%02d,%03d", h, m, s, (int)ms);
}

static void on_result(void *, AprilResultType type, size_t count, const AprilToken *tokens)
{
    if (type != APRIL_RESULT_RECOGNITION_FINAL) return;
    std::string text;
    for (size_t t = 0; t < count; ++t) {
        const size_t start = tokens[t].time_ms, end = t + 1 < count ? tokens[t + 1].time_ms : start + 2000;
        char a[32], b[32];
        stamp(start, a, sizeof a); stamp(end, b, sizeof b);
        text += tokens[t].token;
        printf("%d\n%s --> %s\n%s\n\n", ++g_cue, a, b, text.c_str());
    }
    fflush(stdout);
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s <file.wav | file.raw> <model.april>\n", argv[0]); return 2; }
    aam_api_init(APRIL_VERSION);
    AprilASRModel model = aam_create_model(argv[2]);
    if (!model) { fprintf(stderr, "failed to load model %s\n", argv[2]); return 1; }
    AprilConfig cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.handler = on_result;
    cfg.flags = APRIL_CONFIG_FLAG_ZERO_BIT;
    AprilASRSession session = aas_create_session(model, cfg);
    if (!session) { fprintf(stderr, "failed to create session\n"); aam_free(model); return 1; }
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 1; }
    const size_t n = strlen(argv[1]);
    if (n > 4 && !strcmp(argv[1] + n - 4, ".wav")) fseek(in, 44, SEEK_SET);     // canonical 44-byte header (examples/main.cpp parses the chunks properly)
    // the whole file in one call: long feeds take the engine's layer-major schedule
    std::vector<short> pcm;
    short buf[4096];
    size_t got;
    while ((got = fread(buf, sizeof(short), 4096, in)) > 0) pcm.insert(pcm.end(), buf, buf + got);
    fclose(in);
    aas_feed_pcm16(session, pcm.data(), pcm.size());
    aas_flush(session);
    aas_free(session);
    aam_free(model);
    return 0;
}
