// CPU-only check of the round-6 loader additions under -fsanitize=address,undefined (tests/test_loader_sanitized.py):
//   * build_fbank_tables (csrc/fbank_tables.cc) for EVERY frame length 8 .. 2100 and a few larger ones: factor lists multiply back to the
//     length, twiddle and root-of-unity tables have the sizes the device passes index, every value is a finite point of the unit circle;
//   * pad_host_model (csrc/model_loader.cc) on models with random widths that are multiples of 16: padded sizes, real entries kept, padding zero.
// The device passes (kernels_fbank.hip) read tw[k][(j-1)(ido-1) + 2i-2 / 2i-1] and tws[k][2 i / 2 i + 1]: an index beyond these tables is what this guards.
#include "fbank_tables.h"
#include "model_loader.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace aprilx;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { ++fails; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static void check_tables(int n)
{
    FbankHostTables t;
    // frame_length_ms = 25 with sample_rate = 40 n gives a window of n samples; round_pow2 = 0
    const bool ok = build_fbank_tables(40 * n, 10, 25, 80, false, 20, 0, t);
    if (!ok) return;                                   // (a Bluestein length: refused)
    CHECK(t.padded == n, "padded %d != %d", t.padded, n);
    long prod = 1;
    for (int f : t.factors) prod *= f;
    CHECK(prod == n, "factors of %d multiply to %ld", n, prod);
    CHECK(t.tw.size() == t.factors.size() && t.tws.size() == t.factors.size(), "table count for %d", n);
    size_t l1 = 1;
    for (size_t k = 0; k < t.factors.size(); ++k) {
        const size_t ip = (size_t)t.factors[k], ido = (size_t)n / (l1 * ip);
        if (k + 1 < t.factors.size()) CHECK(t.tw[k].size() == (ip - 1) * (ido - 1), "tw size n %d factor %zu", n, ip);
        if (ip > 5) CHECK(t.tws[k].size() == 2 * ip, "tws size n %d factor %zu", n, ip); else CHECK(t.tws[k].empty(), "tws for a small factor");
        if (k + 1 < t.factors.size())
            for (size_t j = 1; j < ip; ++j)
                for (size_t i = 1; i <= (ido - 1) / 2; ++i) {      // (rows of ido - 1 entries: an odd count when ido is even, the last one unused)
                    const double c = t.tw[k][(j - 1) * (ido - 1) + 2 * i - 2], s = t.tw[k][(j - 1) * (ido - 1) + 2 * i - 1];
                    const double a = 6.283185307179586 * (double)(j * l1 * i) / (double)n;
                    CHECK(std::fabs(c - std::cos(a)) < 1e-12 && std::fabs(s - std::sin(a)) < 1e-12, "twiddle n %d factor %zu j %zu i %zu: %g %g", n, ip, j, i, c, s);
                }
        for (size_t i = 0; i < t.tws[k].size(); i += 2) {
            const double c = t.tws[k][i], s = t.tws[k][i + 1], a = 6.283185307179586 * (double)(i / 2) / (double)ip;
            CHECK(std::fabs(c - std::cos(a)) < 1e-12 && std::fabs(s - std::sin(a)) < 1e-12, "root of unity n %d factor %zu entry %zu", n, ip, i / 2);
        }
        l1 *= ip;
    }
    // spot value of the first factor's first twiddle: (cos, sin)(2 pi l1 j i / n) with l1 = 1, j = 1, i = 1
    if (t.factors.size() > 1 && !t.tw[0].empty()) {
        const double a = 6.283185307179586 / (double)n;
        CHECK(std::fabs(t.tw[0][0] - std::cos(a)) < 1e-12 && std::fabs(t.tw[0][1] - std::sin(a)) < 1e-12, "first twiddle of %d", n);
    }
}

static std::vector<float> rnd(std::mt19937 &g, size_t n) { std::vector<float> v(n); for (float &x : v) x = (float)(int)(g() % 2001 - 1000) / 1000.0f + 0.0005f; return v; }

static void check_pad(std::mt19937 &g)
{
    HostModel m;
    NetDims &D = m.dims;
    const int gs = 4 << (g() % 2);                                  // decoder group size 4 or 8
    D.n_layers = 1 + (int)(g() % 2); D.d_model = 16 * (int)(1 + g() % 20); D.hidden = 16 * (int)(1 + g() % 24); D.ffn = 16 * (int)(1 + g() % 30);
    D.joiner = 16 * (int)(1 + g() % 12); D.vocab = 17 + (int)(g() % 60); D.mel = 80; D.seg = 9; D.context = 2;
    while (D.d_model % gs) D.d_model += 16;
    D.dec_groups = D.d_model / gs;
    D.conv_ch[0] = 8; D.conv_ch[1] = 16; D.conv_ch[2] = 16 * (int)(1 + g() % 9); D.f_out = 19; D.embed_in = D.conv_ch[2] * D.f_out;
    const int d = D.d_model, h = D.hidden, f = D.ffn, j = D.joiner, c2 = D.conv_ch[2], F = D.f_out, V = D.vocab;
    m.conv_w[2] = rnd(g, (size_t)c2 * D.conv_ch[1] * 9); m.conv_b[2] = rnd(g, (size_t)c2);
    m.w_embed = rnd(g, (size_t)c2 * F * d); m.b_embed = rnd(g, (size_t)d);
    m.layers.resize((size_t)D.n_layers);
    for (LayerWeights &lw : m.layers) {
        lw.w_gates = rnd(g, (size_t)2 * d * 4 * h); lw.b_gates = rnd(g, (size_t)4 * h); lw.w_hr = rnd(g, (size_t)h * d);
        lw.w_ff1 = rnd(g, (size_t)d * f); lw.b_ff1 = rnd(g, (size_t)f); lw.w_ff2 = rnd(g, (size_t)f * d); lw.b_ff2 = rnd(g, (size_t)d);
    }
    m.w_encproj = rnd(g, (size_t)d * j); m.b_encproj = rnd(g, (size_t)j); m.emb = rnd(g, (size_t)V * d);
    m.dec_conv = rnd(g, (size_t)d * gs * 2); if (g() & 1) m.dec_conv_b = rnd(g, (size_t)d);
    m.w_decproj = rnd(g, (size_t)d * j); m.b_decproj = rnd(g, (size_t)j); m.w_out = rnd(g, (size_t)j * V); m.b_out = rnd(g, (size_t)V);
    const HostModel before = m;
    std::string err;
    const bool ok = pad_host_model(m, err);
    const int d2 = (d + 63) & ~63, h2 = (h + 63) & ~63, f2 = (f + 63) & ~63, j2 = (j + 63) & ~63;
    if (!ok) { CHECK((d2 - d) % gs != 0, "refused without a reason: %s", err.c_str()); return; }
    const NetDims &P = m.dims;
    CHECK(P.d_model == d2 && P.hidden == h2 && P.ffn == f2 && P.joiner == j2 && P.embed_in == P.conv_ch[2] * F && P.embed_in % 64 == 0, "padded dims");
    CHECK(P.dec_groups * gs == d2, "decoder groups");
    CHECK((P.d_norm == 0) == (d2 == d && h2 == h && f2 == f && j2 == j && P.conv_ch[2] == c2), "d_norm set exactly when something was padded");
    CHECK(m.layers[0].w_gates.size() == (size_t)2 * d2 * 4 * h2 && m.w_embed.size() == (size_t)P.embed_in * d2 && m.w_out.size() == (size_t)j2 * V && m.emb.size() == (size_t)V * d2, "sizes");
    // every real gate weight at its padded place, everything else zero (sum check)
    const LayerWeights &a = before.layers[0], &b = m.layers[0];
    double sa = 0, sb = 0;
    for (float x : a.w_gates) sa += (double)x;
    for (float x : b.w_gates) sb += (double)x;
    CHECK(std::fabs(sa - sb) < 1e-6 * (1 + std::fabs(sa)), "gate weight sums %g %g", sa, sb);
    for (int r = 0; r < 2 * d; ++r) for (int c = 0; c < 4 * h; c += 37) {
        const int r2 = r < d ? r : d2 + (r - d), c2n = (c / h) * h2 + c % h;
        CHECK(a.w_gates[(size_t)r * 4 * h + c] == b.w_gates[(size_t)r2 * 4 * h2 + c2n], "gate weight moved wrongly");
    }
    for (int c = 0; c < 4 * h; ++c) CHECK(a.b_gates[(size_t)c] == b.b_gates[(size_t)((c / h) * h2 + c % h)], "gate bias");
    for (int r = 0; r < f; r += 5) for (int c = 0; c < d; c += 7) CHECK(a.w_ff2[(size_t)r * d + c] == b.w_ff2[(size_t)r * d2 + c], "ff2");
    for (int r = f; r < f2; ++r) for (int c = 0; c < d2; ++c) CHECK(b.w_ff2[(size_t)r * d2 + c] == 0.0f, "ff2 padding row");
}

int main()
{
    for (int n = 8; n <= 2100; ++n) check_tables(n);
    for (int n : {2187, 2401, 3125, 4096, 4410, 6615, 8192}) check_tables(n);
    std::mt19937 g(12345);
    for (int i = 0; i < 60; ++i) check_pad(g);
    printf(fails ? "%d FAILURES\n" : "loader tables / padding: all checks passed\n", fails);
    return fails ? 1 : 0;
}
