// Measurement + parity aid for the GM_TILE schedule (csrc/kernels_gemm_tile.hip), linked against the product's objects:
//   1. bitwise comparison of every output (rows, state rows, sums of squares) between the round-2 schedules (GM_FULLK fused /
//      GM_SLAB + row kernel) and GM_TILE at every (tile rows, slabs per workgroup) choice, fused and split + row kernel;
//   2. back-to-back launch time of each choice, one problem per launch and z-batched (n problems of one shape, own weights),
//      against the round-2 kernel on the same arguments.
// build: make -C april_asr_amd/csrc && hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iapril_asr_amd/csrc tools/tile_bench.hip \
//        april_asr_amd/csrc/build/kernels_gemm.o april_asr_amd/csrc/build/kernels_gemm_tile.o april_asr_amd/csrc/build/kernels_misc.o -o tools/tile_bench
// usage: tools/tile_bench [iters=200]
#include "kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace aprilx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class T> static T *dalloc(size_t n) { T *p; CK(hipMalloc((void **)&p, n * sizeof(T))); return p; }

struct Problem {
    int M, N, K, kz, epi;            // epi: EPI_HR (projection), EPI_RESID_SSQ (FFN down / embed), EPI_SLOT_STORE (encoder_proj)
    float *a, *w, *bias, *resid, *ssq_in, *out, *state, *ssq_out, *ws;
    int *slots;
};

static void fill(std::vector<float> &h, unsigned seed, float scale)
{
    unsigned s = seed * 2654435761u + 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((float)((s >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }
}

static int g_apad = 0;      // TB_APAD: floats of padding behind every activation row (leading dimension K + pad): L2 channel spread experiment

static Problem make_problem(int M, int N, int K, int kz, int epi, unsigned seed)
{
    Problem p{M, N, K, kz, epi};
    std::vector<float> h;
    {   // rows of K floats at a leading dimension of K + g_apad
        std::vector<float> dense((size_t)M * K); fill(dense, seed, 2.0f);
        h.assign((size_t)M * (K + g_apad), 0.0f);
        for (int m = 0; m < M; ++m) memcpy(&h[(size_t)m * (K + g_apad)], &dense[(size_t)m * K], (size_t)K * 4);
        p.a = dalloc<float>(h.size()); CK(hipMemcpy(p.a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    h.resize((size_t)K * N); fill(h, seed + 1, 0.1f); p.w = dalloc<float>(h.size()); CK(hipMemcpy(p.w, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)N); fill(h, seed + 2, 1.0f); p.bias = dalloc<float>(h.size()); CK(hipMemcpy(p.bias, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)M * N); fill(h, seed + 3, 1.0f); p.resid = dalloc<float>(h.size()); CK(hipMemcpy(p.resid, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)M * (N / 32 + K / 32)); fill(h, seed + 4, 1.0f); for (auto &v : h) v = v * v + 0.1f;
    p.ssq_in = dalloc<float>(h.size()); CK(hipMemcpy(p.ssq_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    p.out = dalloc<float>((size_t)M * N); p.state = dalloc<float>((size_t)M * N); p.ssq_out = dalloc<float>((size_t)M * (N / 32)); p.ws = dalloc<float>((size_t)8 * M * N);
    std::vector<int> hi((size_t)M);
    for (int i = 0; i < M; ++i) hi[(size_t)i] = (int)(((unsigned)i * 2654435761u) % (unsigned)M);       // a permutation when M is a power of two; otherwise collisions are harmless for timing
    {   // make it a permutation for any M (state rows must not be written twice with different values)
        std::vector<int> perm((size_t)M); for (int i = 0; i < M; ++i) perm[(size_t)i] = i;
        unsigned s = seed; for (int i = M - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; std::swap(perm[(size_t)i], perm[(size_t)((s >> 8) % (unsigned)(i + 1))]); }
        hi = perm;
    }
    p.slots = dalloc<int>((size_t)M); CK(hipMemcpy(p.slots, hi.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    return p;
}

// the GEMM of a problem: fused (row epilogue) or partial planes; tile_ok selects the schedule family
static GemmArgs gemm_of(const Problem &p, bool fused, bool tile_ok, int zcount)
{
    GemmArgs g;
    g.a0 = p.a; g.lda0 = p.K + g_apad; g.K0 = p.K; g.wp = p.w; g.M = p.M; g.N = p.N; g.K = p.K; g.kz = p.kz; g.tile_ok = tile_ok ? 1 : 0; g.zcount = zcount;
    if (p.epi == EPI_SLOT_STORE) { g.x_scale.ssq = p.ssq_in; g.x_scale.groups = p.K / 32; g.x_scale.inv_n = 1.0f / p.K; g.x_scale.eps = 0.25f; }
    if (!fused) { g.epi = EPI_PARTIAL; g.out = p.ws; g.m_stride = p.M; return g; }
    g.epi = p.epi; g.out = p.out; g.ldo = p.N; g.bias = p.bias;
    if (p.epi == EPI_HR) { g.state = p.state; g.ld_state = p.N; g.slot_idx = p.slots; g.resid = p.resid; g.ldr = p.N;
                           g.r_scale.ssq = p.ssq_in; g.r_scale.groups = p.N / 32; g.r_scale.inv_n = 1.0f / p.N; g.r_scale.eps = 0.25f; g.bias = nullptr; }
    else if (p.epi == EPI_RESID_SSQ) { g.resid = p.resid; g.ldr = p.N; g.ssq_out = p.ssq_out; }
    else { g.slot_idx = p.slots; g.state = nullptr; }
    return g;
}

static RowArgs row_of(const Problem &p, int parts)
{
    RowArgs r; r.ws = p.ws; r.parts = parts; r.m_stride = p.M; r.N = p.N; r.M = p.M; r.out = p.out; r.ldo = p.N;
    if (p.epi == EPI_HR) { r.mode = ROW_HR; r.resid = p.resid; r.ldr = p.N; r.slot_idx = p.slots; r.state = p.state; r.ld_state = p.N;
                           r.r_scale.ssq = p.ssq_in; r.r_scale.groups = p.N / 32; r.r_scale.inv_n = 1.0f / p.N; r.r_scale.eps = 0.25f; }
    else if (p.epi == EPI_RESID_SSQ) { r.mode = ROW_RESID_SSQ; r.bias = p.bias; r.resid = p.resid; r.ldr = p.N; r.ssq_out = p.ssq_out; }
    else { r.mode = ROW_SLOT_STORE; r.bias = p.bias; r.slot_idx = p.slots; r.r_scale.ssq = p.ssq_in; r.r_scale.groups = p.K / 32; r.r_scale.inv_n = 1.0f / p.K; r.r_scale.eps = 0.25f; }
    return r;
}

// one launch sequence for n problems of one shape: z-batched GEMM (+ z-batched row kernel when split)
struct Chain {
    std::vector<GemmArgs> gh; GemmArgs *gd = nullptr; std::vector<RowArgs> rh; RowArgs *rd = nullptr; bool split = false; int n = 0;
    void run(hipStream_t s) const
    {
        if (n == 1) { launch_gemm(gh[0], s); if (split) launch_row(rh[0], s); }
        else { launch_gemm_z(gh.data(), n, gd, s); if (split) launch_row_z(rh.data(), n, rd, s); }
    }
};

static Chain make_chain(const std::vector<Problem> &ps, bool tile_ok)
{
    Chain c; c.n = (int)ps.size();
    const Problem &p0 = ps[0];
    const bool fused = gemm_fullk(p0.M, p0.N, p0.kz, /*force*/ !tile_ok && c.n > 1, c.n, tile_ok ? 1 : 0);
    c.split = !fused;
    std::vector<GemmArgs> items;
    for (const Problem &p : ps) {
        GemmArgs g = gemm_of(p, fused, tile_ok, c.n);
        if (fused && !tile_ok && c.n > 1) g.force_fullk = 1;         // what the round-2 feed wavefront does
        items.push_back(g);
        if (!fused) c.rh.push_back(row_of(p, gemm_partials(p.M, p.N, p.kz, c.n, tile_ok ? 1 : 0)));
    }
    if (c.n == 1) c.gh = items;
    else {
        c.gh.resize(items.size()); stage_gemm_z(items.data(), c.n, c.gh.data());
        c.gd = dalloc<GemmArgs>(items.size()); CK(hipMemcpy(c.gd, c.gh.data(), items.size() * sizeof(GemmArgs), hipMemcpyHostToDevice));
        if (!fused) { c.rd = dalloc<RowArgs>(c.rh.size()); CK(hipMemcpy(c.rd, c.rh.data(), c.rh.size() * sizeof(RowArgs), hipMemcpyHostToDevice)); }
    }
    return c;
}

static std::vector<float> snapshot(const std::vector<Problem> &ps)
{
    std::vector<float> all;
    for (const Problem &p : ps) {
        std::vector<float> h((size_t)p.M * p.N);
        CK(hipMemcpy(h.data(), p.out, h.size() * 4, hipMemcpyDeviceToHost)); all.insert(all.end(), h.begin(), h.end());
        if (p.epi == EPI_HR) { CK(hipMemcpy(h.data(), p.state, h.size() * 4, hipMemcpyDeviceToHost)); all.insert(all.end(), h.begin(), h.end()); }
        if (p.epi == EPI_RESID_SSQ) { h.resize((size_t)p.M * (p.N / 32)); CK(hipMemcpy(h.data(), p.ssq_out, h.size() * 4, hipMemcpyDeviceToHost)); all.insert(all.end(), h.begin(), h.end()); }
    }
    return all;
}

static void clear_outputs(const std::vector<Problem> &ps)
{
    for (const Problem &p : ps) {
        CK(hipMemset(p.out, 0xff, (size_t)p.M * p.N * 4)); CK(hipMemset(p.state, 0xff, (size_t)p.M * p.N * 4)); CK(hipMemset(p.ssq_out, 0xff, (size_t)p.M * (p.N / 32) * 4));
        CK(hipMemset(p.ws, 0xff, (size_t)8 * p.M * p.N * 4));
    }
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
    return ms * 1e3 / iters;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    if (getenv("TB_APAD")) g_apad = atoi(getenv("TB_APAD"));
    // optional: only shape number `only` (0-based), only the pinned (mt, zs) -- for profiler runs
    const int only = argc > 2 ? atoi(argv[2]) : -1, only_mt = argc > 3 ? atoi(argv[3]) : -1, only_zs = argc > 4 ? atoi(argv[4]) : -1;
    hipStream_t s; CK(hipStreamCreate(&s));
    struct Shape { const char *name; int M, N, K, kz, epi, n; };
    const Shape shapes[] = {
        {"proj  256x3", 256, 512, 1024, 4, EPI_HR, 3},
        {"proj  256x2", 256, 512, 1024, 4, EPI_HR, 2},
        {"ffdn  256x3", 256, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"ffdn  256x2", 256, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"ffdn  256x1", 256, 512, 2048, 8, EPI_RESID_SSQ, 1},
        {"proj  250x2", 250, 512, 1024, 4, EPI_HR, 2},                  // ragged rows
        {"encp  256x1", 256, 512, 512, 8, EPI_SLOT_STORE, 1},
        {"embed 256x1", 256, 512, 2432, 2, EPI_RESID_SSQ, 1},           // odd chunk length (19 k blocks)
        {"proj   64x3", 64, 512, 1024, 4, EPI_HR, 3},
        {"ffdn   40x2", 40, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"proj 1024x2", 1024, 512, 1024, 4, EPI_HR, 2},
        {"ffdn 1024x2", 1024, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"proj 2048x2", 2048, 512, 1024, 4, EPI_HR, 2},
        {"ffdn 2048x2", 2048, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"ffdn 2048x3", 2048, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"ffdn 512x3 L", 512, 768, 3072, 8, EPI_RESID_SSQ, 3},          // larger encoder (configs[4] dims)
        {"proj 2560x2", 2560, 512, 1024, 4, EPI_HR, 2},                 // the capacity boundary (2560 .. 2816 sessions): 640 .. 1056 tiles of 64 x 64
        {"proj 2560x3", 2560, 512, 1024, 4, EPI_HR, 3},
        {"ffdn 2560x2", 2560, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"ffdn 2560x3", 2560, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"proj 2816x2", 2816, 512, 1024, 4, EPI_HR, 2},
        {"proj 2816x3", 2816, 512, 1024, 4, EPI_HR, 3},
        {"ffdn 2816x2", 2816, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"ffdn 2816x3", 2816, 512, 2048, 8, EPI_RESID_SSQ, 3},
    };
    int bad = 0;
    for (const Shape &sh : shapes) {
        if (only >= 0 && (&sh - shapes) != only) continue;
        if (getenv("TB_FIRST") && (&sh - shapes) < atoi(getenv("TB_FIRST"))) continue;
        std::vector<Problem> ps;
        for (int i = 0; i < sh.n; ++i) ps.push_back(make_problem(sh.M, sh.N, sh.K, sh.kz, sh.epi, 1000u * (unsigned)(&sh - shapes) + 10u * (unsigned)i));
        const double flops = 2.0 * sh.M * sh.N * sh.K * sh.n;
        // reference: the round-2 schedules
        gemm_tile_pin(0, 0, 0);
        Chain ref = make_chain(ps, false);
        clear_outputs(ps); ref.run(s); CK(hipStreamSynchronize(s));
        const std::vector<float> want = snapshot(ps);
        const double t_ref = time_chain(ref, s, iters);
        printf("%-12s M=%d N=%d K=%d kz=%d x%d | round-2 %s: %7.2f us (%.3f of peak)\n", sh.name, sh.M, sh.N, sh.K, sh.kz, sh.n, ref.split ? "split+row" : "fused", t_ref, flops / (t_ref * 1e-6) / 157.3e12);
        // GM_TILE: planner's choice first, then every pinned (mt, zs)
        struct Pin { int mt, zs; };
        std::vector<Pin> pins = {{0, 0}};
        for (int mt : {4, 2}) for (int zs = 1; zs <= sh.kz; zs *= 2) pins.push_back({mt, zs});
        for (const Pin &pin : pins) {
            if (only_mt >= 0 && (pin.mt != only_mt || pin.zs != only_zs)) continue;
            gemm_tile_pin(1, pin.mt, pin.zs);
            if (!gemm_tile_planned(sh.M, sh.N, sh.kz, sh.n)) { printf("    (GM_TILE not planned for this shape)\n"); break; }
            Chain c = make_chain(ps, true);
            clear_outputs(ps); c.run(s); CK(hipStreamSynchronize(s));
            const std::vector<float> got = snapshot(ps);
            size_t diff = 0; double maxd = 0;
            for (size_t i = 0; i < want.size(); ++i) if (memcmp(&want[i], &got[i], 4) != 0) { ++diff; maxd = std::max(maxd, (double)fabsf(want[i] - got[i])); }
            const double t = time_chain(c, s, iters);
            // a second comparison after the timing loop (races show up under back-to-back launches)
            clear_outputs(ps); for (int i = 0; i < 5; ++i) c.run(s); CK(hipStreamSynchronize(s));
            const std::vector<float> got2 = snapshot(ps);
            size_t diff2 = 0; for (size_t i = 0; i < want.size(); ++i) if (memcmp(&want[i], &got2[i], 4) != 0) ++diff2;
            if (getenv("TB_TRACE") && c.n == 1) {      // (binary built with -DAPRIL_GEMM_TRACE) per-phase s_memtime sums of wave 0, averaged over the workgroups
                const int nwg = 16384;
                unsigned long long *tr = dalloc<unsigned long long>((size_t)nwg * 8); CK(hipMemset(tr, 0, (size_t)nwg * 8 * 8));
                Chain ct = c; ct.gh[0].trace = tr; ct.run(s); CK(hipStreamSynchronize(s));
                std::vector<unsigned long long> ht((size_t)nwg * 8); CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
                double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nw = 0;
                for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 8 + 7]) { ++nw; for (int k = 0; k < 8; ++k) acc[k] += (double)ht[(size_t)i * 8 + k]; }
                if (nw) printf("      trace (%d workgroups, %.0f stages each; s_memtime ticks per stage): first k block %.0f | chunk ends %.0f | wait+barrier %.0f | issue + second k block %.0f | loop total %.0f ; prologue %.0f, epilogue %.0f ticks\n",
                               nw, acc[7] / nw, acc[0] / acc[7], acc[1] / acc[7], acc[2] / acc[7], acc[3] / acc[7], acc[4] / acc[7], acc[5] / nw, acc[6] / nw);
            }
            const int parts = c.split ? c.rh[0].parts : 1;
            printf("    tile mt=%d zs=%d%s -> %s planes=%d : %7.2f us (%.3f of peak, %.2fx)  %s\n", pin.mt, pin.zs, pin.mt == 0 ? " (planner)" : "", c.split ? "split+row" : "fused", parts, t,
                   flops / (t * 1e-6) / 157.3e12, t_ref / t, (diff || diff2) ? "MISMATCH" : "bit-identical");
            if (diff || diff2) { ++bad; printf("      mismatching floats: %zu / %zu of %zu (max |d| %.3g)\n", diff, diff2, want.size(), maxd); }
        }
        fflush(stdout);
    }
    gemm_tile_pin(-1, 0, 0);
    printf(bad ? "FAILED: %d configurations differ\n" : "all configurations bit-identical\n", bad);
    return bad ? 1 : 0;
}
