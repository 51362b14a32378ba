// Measurement + parity aid for the GM_PP schedule (csrc/kernels_gemm_pp.hip), linked against the product's objects:
//   1. bitwise comparison of every output (cell state, binary16 rows, fp32 partial rows) between GM_TILE (the round-3 fp16 tile
//      kernels, APRIL_GM_PP=0 form) and GM_PP at 256- and 128-row tiles: gates (two A segments, BasicNorm scale, LSTM cell), FFN up
//      (DoubleSwish), and the layer-major halves of the gate GEMM (EPI_XPART, EPI_LSTM + p_add), ragged row counts included;
//   2. back-to-back launch time of each form, one problem per launch and z-batched (n problems of one shape, own weights).
// build: tools/build_pp_bench.sh      usage: tools/pp_bench [iters=200] [dims: large|v0|both]
#include "kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
using namespace aprilx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class T> static T *dalloc(size_t n) { T *p; CK(hipMalloc((void **)&p, n * sizeof(T))); return p; }

static unsigned long long *g_trace = nullptr;      // PPB_TRACE=1 (pp_bench_trace build): per-workgroup phase stamps of the GM_PP kernel
static const size_t TRACE_WGS = 4096;
static unsigned g_seed = 1;
static float frand(float scale) { g_seed = g_seed * 1664525u + 1013904223u; return ((float)((g_seed >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }

static _Float16 *dev_halves(size_t n, float scale)
{
    std::vector<_Float16> h(n);
    for (auto &v : h) v = (_Float16)frand(scale);
    _Float16 *d = dalloc<_Float16>(n); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static float *dev_floats(size_t n, float scale, bool square = false)
{
    std::vector<float> h(n);
    for (auto &v : h) { v = frand(scale); if (square) v = v * v + 0.1f; }
    float *d = dalloc<float>(n); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

// one layer's gates (or FFN-up) problem on binary16 operands
struct Problem {
    int M, d, hidden, kind;      // kind 0: gates [y16 | h16(slot)] x Wg + cell; 1: FFN up (K = d, N = hidden) + DoubleSwish; 2: XPART half; 3: LSTM h half + p_add
    _Float16 *y16, *h16, *w16, *out16;
    float *ssq, *bias, *c_state, *c_init, *p_out, *p_in;
    int *slots;
};

static Problem make_problem(int M, int d, int hidden, int kind)
{
    Problem p{M, d, hidden, kind};
    const int S = M;
    const int N = kind == 1 ? hidden : 4 * hidden, K = kind == 1 ? d : 2 * d;
    p.y16 = dev_halves((size_t)M * d, 2.0f);
    p.h16 = dev_halves((size_t)S * d, 2.0f);
    p.w16 = dev_halves((size_t)N * K, 0.1f);
    p.out16 = dalloc<_Float16>((size_t)M * hidden);
    p.ssq = dev_floats((size_t)M * (d / 32), 1.0f, true);
    p.bias = dev_floats((size_t)N, 1.0f);
    p.c_init = dev_floats((size_t)S * hidden, 2.0f);
    p.c_state = dalloc<float>((size_t)S * hidden);
    p.p_out = dalloc<float>((size_t)M * N);
    p.p_in = dev_floats((size_t)M * N, 1.0f);
    std::vector<int> perm((size_t)M);
    for (int i = 0; i < M; ++i) perm[(size_t)i] = i;
    for (int i = M - 1; i > 0; --i) { g_seed = g_seed * 1664525u + 1013904223u; std::swap(perm[(size_t)i], perm[(size_t)((g_seed >> 8) % (unsigned)(i + 1))]); }
    p.slots = dalloc<int>((size_t)M); CK(hipMemcpy(p.slots, perm.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    return p;
}

static GemmArgs args_of(const Problem &p, int zcount)
{
    GemmArgs g;
    g.wt = 1; g.tile_ok = 2; g.kz = 1; g.zcount = zcount; g.M = p.M; g.trace = g_trace;
    g.wp = p.w16;
    if (p.kind == 1) {
        g.a0 = reinterpret_cast<const float *>(p.y16); g.lda0 = p.d; g.K0 = p.d; g.N = p.hidden; g.K = p.d;
        g.epi = EPI_BIAS_DSWISH; g.bias = p.bias; g.out = nullptr; g.out16 = p.out16; g.ldo = p.hidden;
        return g;
    }
    g.a0 = reinterpret_cast<const float *>(p.y16); g.lda0 = p.d; g.K0 = p.d;
    g.a1 = reinterpret_cast<const float *>(p.h16); g.lda1 = p.d; g.aidx1 = p.slots; g.K1 = p.d;
    g.x_scale.ssq = p.ssq; g.x_scale.groups = p.d / 32; g.x_scale.inv_n = 1.0f / (float)p.d; g.x_scale.eps = 0.25f;
    g.N = 4 * p.hidden; g.K = 2 * p.d; g.epi = EPI_LSTM; g.bias = p.bias; g.c_state = p.c_state; g.slot_idx = p.slots; g.hidden = p.hidden;
    g.out = nullptr; g.out16 = p.out16; g.ldo = p.hidden;
    if (p.kind == 2) { g.epi = EPI_XPART; g.wave_mask = 0x3; g.out = p.p_out; g.ldo = g.N; g.out16 = nullptr; g.bias = nullptr; g.c_state = nullptr; g.slot_idx = nullptr; }
    if (p.kind == 3) { g.wave_mask = 0xC; g.p_add = p.p_in; g.ldp = g.N; g.x_scale = RowScale(); }
    return g;
}

struct Chain {
    std::vector<GemmArgs> gh; GemmArgs *gd = nullptr; int n = 0;
    void run(hipStream_t s) const { if (n == 1) launch_gemm(gh[0], s); else launch_gemm_z(gh.data(), n, gd, s); }
};

static Chain make_chain(const std::vector<Problem> &ps)
{
    Chain c; c.n = (int)ps.size();
    std::vector<GemmArgs> items;
    for (const Problem &p : ps) items.push_back(args_of(p, c.n));
    if (c.n == 1) c.gh = items;
    else {
        c.gh.resize(items.size()); stage_gemm_z(items.data(), c.n, c.gh.data());
        c.gd = dalloc<GemmArgs>(items.size()); CK(hipMemcpy(c.gd, c.gh.data(), items.size() * sizeof(GemmArgs), hipMemcpyHostToDevice));
    }
    return c;
}

static void reset(const std::vector<Problem> &ps)
{
    for (const Problem &p : ps) {
        const int N = p.kind == 1 ? p.hidden : 4 * p.hidden;
        CK(hipMemcpy(p.c_state, p.c_init, (size_t)p.M * p.hidden * 4, hipMemcpyDeviceToDevice));
        CK(hipMemset(p.out16, 0xff, (size_t)p.M * p.hidden * 2));
        CK(hipMemset(p.p_out, 0xff, (size_t)p.M * N * 4));
    }
}

static std::vector<unsigned char> snapshot(const std::vector<Problem> &ps)
{
    std::vector<unsigned char> all;
    auto add = [&](const void *d, size_t bytes) { const size_t o = all.size(); all.resize(o + bytes); CK(hipMemcpy(all.data() + o, d, bytes, hipMemcpyDeviceToHost)); };
    for (const Problem &p : ps) {
        const int N = p.kind == 1 ? p.hidden : 4 * p.hidden;
        if (p.kind == 2) add(p.p_out, (size_t)p.M * N * 4);
        else { add(p.out16, (size_t)p.M * p.hidden * 2); if (p.kind != 1) add(p.c_state, (size_t)p.M * p.hidden * 4); }
    }
    return all;
}

static double time_chain(const Chain &c, hipStream_t s, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) c.run(s);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) c.run(s);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / iters;
}

static int g_fail = 0;

static void run_case(const char *name, int M, int d, int hidden, int kind, int n, hipStream_t s, int iters)
{
    std::vector<Problem> ps;
    for (int i = 0; i < n; ++i) ps.push_back(make_problem(M, d, hidden, kind));
    const int N = kind == 1 ? hidden : 4 * hidden, K = kind == 1 ? d : 2 * d;
    const double flops = 2.0 * M * N * ((kind == 2 || kind == 3) ? K / 2 : K) * n;
    // forms: GM_TILE (GM_PP off), GM_PP planner, GM_PP 256-row, GM_PP 128-row
    struct Form { const char *tag; int enable, mt; } forms[] = {{"tile", 0, 0}, {"pp", 1, 0}, {"pp16", 1, 16}, {"pp8", 1, 8}, {"ppw", 1, 12}};
    constexpr int NF = 5;
    std::vector<unsigned char> ref;
    double us[NF] = {0, 0, 0, 0, 0};
    bool same[NF] = {true, true, true, true, true};
    for (int f = 0; f < NF; ++f) {
        gemm_pp_pin(forms[f].enable, forms[f].mt);
        Chain c = make_chain(ps);
        reset(ps);
        c.run(s); CK(hipStreamSynchronize(s));
        const std::vector<unsigned char> got = snapshot(ps);
        if (f == 0) ref = got; else same[f] = got.size() == ref.size() && memcmp(got.data(), ref.data(), ref.size()) == 0;
        if (!same[f]) {
            ++g_fail;
            size_t bad = 0, first = (size_t)-1;
            for (size_t i = 0; i < ref.size() && i < got.size(); ++i) if (ref[i] != got[i]) { ++bad; if (first == (size_t)-1) first = i; }
            fprintf(stderr, "MISMATCH %s form %s: %zu of %zu bytes differ, first at %zu\n", name, forms[f].tag, bad, ref.size(), first);
        }
        us[f] = time_chain(c, s, iters);
        if (g_trace && f >= 2) {
            // one more launch with fresh stamps: mean over the workgroups that wrote (the trace rows of a z-batched launch overlap: the
            // kernel indexes by blockIdx, all problems of one shape -> the last writer wins, fine for a mean)
            CK(hipMemset(g_trace, 0, TRACE_WGS * 16 * 8));
            c.run(s); CK(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(TRACE_WGS * 16);
            CK(hipMemcpy(h.data(), g_trace, h.size() * 8, hipMemcpyDeviceToHost));
            double acc[2][7] = {{0}}; long cnt = 0; double kb = 0;
            for (size_t w = 0; w < TRACE_WGS; ++w) if (h[w * 16 + 7]) { ++cnt; kb = (double)h[w * 16 + 7]; for (int gsel = 0; gsel < 2; ++gsel) for (int i = 0; i < 7; ++i) acc[gsel][i] += (double)h[w * 16 + gsel * 8 + i]; }
            if (cnt) for (int gsel = 0; gsel < 2; ++gsel)
                printf("    trace %-5s wave %d: per k block (cycles, mean of %ld workgroups, %g k blocks): load phase (DMA + read issue) %.0f | barrier after load %.0f | compute phase (MFMA issue + reads land) %.0f | barrier after compute %.0f | scale (per tile) %.0f | prologue %.0f | epilogue %.0f (cycles per tile)\n",
                       forms[f].tag, gsel * 4, cnt, kb, acc[gsel][0] / cnt / kb, acc[gsel][1] / cnt / kb, acc[gsel][2] / cnt / kb, acc[gsel][3] / cnt / kb, acc[gsel][4] / cnt, acc[gsel][5] / cnt, acc[gsel][6] / cnt);
        }
        if (c.gd) CK(hipFree(c.gd));
    }
    // second round of timings, interleaved the other way round (clock / cache state)
    for (int f = NF - 1; f >= 0; --f) {
        gemm_pp_pin(forms[f].enable, forms[f].mt);
        Chain c = make_chain(ps);
        us[f] = std::min(us[f], time_chain(c, s, iters));
        if (c.gd) CK(hipFree(c.gd));
    }
    gemm_pp_pin(-1, 0);
    printf("%-28s M %5d x %d  N %5d K %5d | tile %7.2f us (%6.1f TF) | pp %7.2f (%6.1f TF) %s | pp16 %7.2f (%6.1f TF) %s | pp8 %7.2f (%6.1f TF) %s | ppw %7.2f (%6.1f TF) %s\n", name, M, n, N, K,
           us[0], flops / us[0] * 1e-6, us[1], flops / us[1] * 1e-6, same[1] ? "same" : "DIFF", us[2], flops / us[2] * 1e-6, same[2] ? "same" : "DIFF",
           us[3], flops / us[3] * 1e-6, same[3] ? "same" : "DIFF", us[4], flops / us[4] * 1e-6, same[4] ? "same" : "DIFF");
    fflush(stdout);
    for (const Problem &p : ps) {
        CK(hipFree(p.y16)); CK(hipFree(p.h16)); CK(hipFree(p.w16)); CK(hipFree(p.out16)); CK(hipFree(p.ssq)); CK(hipFree(p.bias));
        CK(hipFree(p.c_state)); CK(hipFree(p.c_init)); CK(hipFree(p.p_out)); CK(hipFree(p.p_in)); CK(hipFree(p.slots));
    }
}

// the same launch over R different problem sets in turn (R x weights > the 256 MB memory-side cache): every launch streams its
// weights from HBM, as inside the engine (16 layers = 0.5 GB of weights per step), instead of finding them where the launch before left them
static void run_cold_case(const char *name, int M, int d, int hidden, int kind, int n, int R, hipStream_t s, int iters)
{
    std::vector<std::vector<Problem>> sets((size_t)R);
    std::vector<Chain> chains;
    for (int r = 0; r < R; ++r) for (int i = 0; i < n; ++i) sets[(size_t)r].push_back(make_problem(M, d, hidden, kind));
    gemm_pp_pin(-1, 0);
    for (int r = 0; r < R; ++r) chains.push_back(make_chain(sets[(size_t)r]));
    auto time_rot = [&](int rot) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 2 * R; ++i) chains[(size_t)(i % rot)].run(s);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) chains[(size_t)(i % rot)].run(s);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / iters;
    };
    const double warm = time_rot(1), cold = time_rot(R);
    printf("%-28s M %5d x %d | planner's form, one problem set over and over %7.2f us | %d sets in turn (weights from HBM) %7.2f us\n", name, M, n, warm, R, cold);
    fflush(stdout);
    for (auto &c : chains) if (c.gd) CK(hipFree(c.gd));
    for (auto &ps : sets) for (const Problem &p : ps) {
        CK(hipFree(p.y16)); CK(hipFree(p.h16)); CK(hipFree(p.w16)); CK(hipFree(p.out16)); CK(hipFree(p.ssq)); CK(hipFree(p.bias));
        CK(hipFree(p.c_state)); CK(hipFree(p.c_init)); CK(hipFree(p.p_out)); CK(hipFree(p.p_in)); CK(hipFree(p.slots));
    }
}

// timing-race check: the same launch again and again while a second stream streams 512 MB copies through HBM (the DMA pieces of the
// ping-pong tiles then land late and in a different order from run to run); every run must give the bits of the GM_TILE form
static void run_stress(const char *name, int M, int d, int hidden, int kind, int n, int mt, hipStream_t s, int runs)
{
    std::vector<Problem> ps;
    for (int i = 0; i < n; ++i) ps.push_back(make_problem(M, d, hidden, kind));
    gemm_pp_pin(0, 0);
    Chain ref_chain = make_chain(ps);
    reset(ps); ref_chain.run(s); CK(hipStreamSynchronize(s));
    const std::vector<unsigned char> ref = snapshot(ps);
    gemm_pp_pin(1, mt);
    Chain c = make_chain(ps);
    hipStream_t hog; CK(hipStreamCreate(&hog));
    char *ha = dalloc<char>((size_t)512 << 20), *hb = dalloc<char>((size_t)512 << 20);
    int bad = 0;
    for (int r = 0; r < runs; ++r) {
        reset(ps);
        if (r & 1) CK(hipMemcpyAsync(hb, ha, (size_t)512 << 20, hipMemcpyDeviceToDevice, hog));      // every other run beside a copy
        c.run(s); CK(hipStreamSynchronize(s));
        const std::vector<unsigned char> got = snapshot(ps);
        if (got.size() != ref.size() || memcmp(got.data(), ref.data(), ref.size()) != 0) ++bad;
        CK(hipStreamSynchronize(hog));
    }
    gemm_pp_pin(-1, 0);
    if (bad) { ++g_fail; fprintf(stderr, "MISMATCH %s: %d of %d runs differ from the GM_TILE form\n", name, bad, runs); }
    printf("%-28s M %5d x %d  form mt=%2d: %d runs, half of them beside a 512 MB device copy: %s\n", name, M, n, mt, runs, bad ? "DIFF" : "all same");
    fflush(stdout);
    CK(hipFree(ha)); CK(hipFree(hb)); CK(hipStreamDestroy(hog));
    if (c.gd) CK(hipFree(c.gd));
    if (ref_chain.gd) CK(hipFree(ref_chain.gd));
    for (const Problem &p : ps) {
        CK(hipFree(p.y16)); CK(hipFree(p.h16)); CK(hipFree(p.w16)); CK(hipFree(p.out16)); CK(hipFree(p.ssq)); CK(hipFree(p.bias));
        CK(hipFree(p.c_state)); CK(hipFree(p.c_init)); CK(hipFree(p.p_out)); CK(hipFree(p.p_in)); CK(hipFree(p.slots));
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const std::string dims = argc > 2 ? argv[2] : "both";
    hipStream_t s; CK(hipStreamCreate(&s));
    if (getenv("PPB_TRACE")) { g_trace = dalloc<unsigned long long>(TRACE_WGS * 16); CK(hipMemset(g_trace, 0, TRACE_WGS * 16 * 8)); }
    if (dims == "large" || dims == "both") {
        // BASELINE configs[4]: 16 layers, d_model 768, cell 1536, ffn 3072; 512 sessions, one to three chunk steps per launch
        for (int n = 1; n <= 3; ++n) run_case("large gates", 512, 768, 1536, 0, n, s, iters);
        for (int n = 1; n <= 3; ++n) run_case("large ffn-up", 512, 768, 3072, 1, n, s, iters);
        run_case("large gates ragged", 437, 768, 1536, 0, 2, s, iters);
        run_case("large gates 1170 rows", 1170, 768, 1536, 0, 1, s, iters);
        run_case("large gates 2048 rows", 2048, 768, 1536, 0, 1, s, iters);
        run_case("large xpart (T x m rows)", 1536, 768, 1536, 2, 1, s, iters);
        run_case("large lstm h-half", 512, 768, 1536, 3, 1, s, iters);
        run_case("large ffn-up 1536 rows", 1536, 768, 3072, 1, 1, s, iters);
        run_case("large gates 100 rows", 100, 768, 1536, 0, 3, s, iters);
    }
    if (dims == "stress") {
        run_stress("large gates", 512, 768, 1536, 0, 3, 12, s, iters);
        run_stress("large gates", 437, 768, 1536, 0, 2, 12, s, iters);
        run_stress("large gates", 512, 768, 1536, 0, 2, 16, s, iters);
        run_stress("large gates", 512, 768, 1536, 0, 1, 8, s, iters);
        run_stress("large ffn-up", 512, 768, 3072, 1, 3, 12, s, iters);
        run_stress("large ffn-up", 512, 768, 3072, 1, 3, 16, s, iters);
        run_stress("v0 gates", 1024, 512, 1024, 0, 2, 16, s, iters);
    }
    if (dims == "cold") {
        for (int n = 1; n <= 3; ++n) run_cold_case("large gates", 512, 768, 1536, 0, n, 24 / n, s, iters);
        for (int n = 2; n <= 3; ++n) run_cold_case("large ffn-up", 512, 768, 3072, 1, n, 24, s, iters);
    }
    if (dims == "probe") {
        // shapes of the N = d_model GEMMs (projection K = cell, FFN down K = ffn) under the FFN-up epilogue, and the gates at twice the K:
        // what GM_PP would do there, and how a launch splits into per-tile fixed time and K-proportional time
        for (int n = 1; n <= 3; ++n) run_case("probe N768 K1536 (proj)", 512, 1536, 768, 1, n, s, iters);
        for (int n = 1; n <= 3; ++n) run_case("probe N768 K3072 (ffn-down)", 512, 3072, 768, 1, n, s, iters);
        // what a 2-way K split of the N = d_model GEMMs on 128 x 128 ping-pong tiles would cost before its reduction: twice the workgroups, half the K
        for (int n = 2; n <= 3; ++n) run_case("probe N768 K768 1024 rows", 1024, 768, 768, 1, n, s, iters);
        for (int n = 2; n <= 3; ++n) run_case("probe N768 K1536 1024 rows", 1024, 1536, 768, 1, n, s, iters);
        run_case("probe gates K 1536", 512, 768, 1536, 0, 1, s, iters);
        run_case("probe gates K 3072", 512, 1536, 1536, 0, 1, s, iters);
        run_case("probe gates K 1536 x2", 512, 768, 1536, 0, 2, s, iters);
        run_case("probe gates K 3072 x2", 512, 1536, 1536, 0, 2, s, iters);
        run_case("probe ffn-up K 768", 512, 768, 3072, 1, 1, s, iters);
        run_case("probe ffn-up K 1536", 512, 1536, 3072, 1, 1, s, iters);
    }
    if (dims == "v0" || dims == "both") {
        // aprilv0 dims on binary16 operands: d_model 512, cell 1024, ffn 2048
        for (int n = 1; n <= 3; ++n) run_case("v0 gates", 256, 512, 1024, 0, n, s, iters);
        run_case("v0 gates 1024 rows", 1024, 512, 1024, 0, 2, s, iters);
        run_case("v0 ffn-up", 256, 512, 2048, 1, 3, s, iters);
        run_case("v0 ffn-up 1024 rows", 1024, 512, 2048, 1, 2, s, iters);
        run_case("v0 gates ragged", 301, 512, 1024, 0, 1, s, iters);
        run_case("v0 xpart", 768, 512, 1024, 2, 1, s, iters);
        run_case("v0 lstm h-half", 256, 512, 1024, 3, 1, s, iters);
    }
    printf(g_fail ? "pp_bench: %d MISMATCHES\n" : "pp_bench: all forms bit-identical\n", g_fail);
    return g_fail ? 1 : 0;
}
