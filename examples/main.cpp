// Command-line transcriber against the C ABI in include/april_api.h -- the role of the reference's
// example.cpp (`./main file.wav model.april`, `./main - model.april` for raw PCM16 on stdin), written for
// this library: no GPU- or engine-specific call is needed, only the reference's twelve entry points.
//
//   g++ -O2 -std=c++17 examples/main.cpp -I include -L april_asr_amd -laprilasr -Wl,-rpath,$PWD/april_asr_amd -o main
//   ./main audio.wav model.april          (16-bit mono PCM WAV at the model's sample rate)
//   parec --format=s16 --rate=16000 --channels=1 --latency-ms=100 | ./main - model.april
//
// Output format follows the reference's: "- text" for partial results, "@ text" for final ones.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "april_api.h"

static void on_result(void *, AprilResultType type, size_t count, const AprilToken *tokens)
{
    if (type == APRIL_RESULT_ERROR_CANT_KEEP_UP) { fprintf(stderr, "can't keep up\n"); return; }
    if (type == APRIL_RESULT_SILENCE) { printf("\n"); fflush(stdout); return; }
    std::string line = type == APRIL_RESULT_RECOGNITION_FINAL ? "@ " : "- ";
    for (size_t i = 0; i < count; ++i) line += tokens[i].token;
    printf("%s\n", line.c_str());
    fflush(stdout);
}

// minimal RIFF/WAVE reader: returns the byte offset of the PCM data or -1
static long wav_data_offset(FILE *f, int *channels, int *rate, int *bits)
{
    unsigned char h[12];
    if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) return -1;
    for (;;) {
        unsigned char c[8];
        if (fread(c, 1, 8, f) != 8) return -1;
        const unsigned len = c[4] | (c[5] << 8) | (c[6] << 16) | ((unsigned)c[7] << 24);
        if (!memcmp(c, "fmt ", 4)) {
            std::vector<unsigned char> b(len);
            if (fread(b.data(), 1, len, f) != len || len < 16) return -1;
            *channels = b[2] | (b[3] << 8);
            *rate = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            *bits = b[14] | (b[15] << 8);
        } else if (!memcmp(c, "data", 4)) {
            return ftell(f);
        } else {
            fseek(f, len + (len & 1), SEEK_CUR);
        }
    }
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s <file.wav | file.raw | -> <model.april>\n", argv[0]); return 2; }
    aam_api_init(APRIL_VERSION);
    AprilASRModel model = aam_create_model(argv[2]);
    if (!model) { fprintf(stderr, "failed to load model %s\n", argv[2]); return 1; }
    fprintf(stderr, "model: %s (%s), %s, %zu Hz\n", aam_get_name(model), aam_get_language(model), aam_get_description(model), aam_get_sample_rate(model));

    AprilConfig cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.handler = on_result;
    cfg.flags = APRIL_CONFIG_FLAG_ZERO_BIT;           // synchronous: results arrive inside aas_feed_pcm16
    AprilASRSession session = aas_create_session(model, cfg);
    if (!session) { fprintf(stderr, "failed to create session\n"); aam_free(model); return 1; }

    FILE *in = strcmp(argv[1], "-") ? fopen(argv[1], "rb") : stdin;
    if (!in) { perror(argv[1]); return 1; }
    const size_t n = strlen(argv[1]);
    if (in != stdin && n > 4 && !strcmp(argv[1] + n - 4, ".wav")) {
        int ch = 0, rate = 0, bits = 0;
        if (wav_data_offset(in, &ch, &rate, &bits) < 0 || ch != 1 || bits != 16 || (size_t)rate != aam_get_sample_rate(model)) {
            fprintf(stderr, "need 16-bit mono PCM WAV at %zu Hz (got %d ch, %d bit, %d Hz)\n", aam_get_sample_rate(model), ch, bits, rate);
            return 1;
        }
    }
    std::vector<short> buf(1600);                     // 100 ms at 16 kHz, like `parec --latency-ms=100`
    size_t got;
    while ((got = fread(buf.data(), sizeof(short), buf.size(), in)) > 0) aas_feed_pcm16(session, buf.data(), got);
    aas_flush(session);
    aas_free(session);
    aam_free(model);
    return 0;
}
