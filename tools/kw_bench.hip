// Measurement + parity aid for the GM_KW schedule (csrc/kernels_gemm_kw.hip), linked against the product's objects:
//   1. bitwise comparison of every output (rows, state rows, sums of squares) between the schedules the planner would pick with GM_KW
//      off (GM_FULLK / GM_SLAB fused tiles, the hand-scheduled FFN-up tiles) and GM_KW at each tile height;
//   2. back-to-back launch time of each, one problem per launch and z-batched (n problems of one shape, own weights and rows).
// build: make -C april_asr_amd/csrc && hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iapril_asr_amd/csrc tools/kw_bench.hip \
//        april_asr_amd/csrc/build/kernels_gemm.o april_asr_amd/csrc/build/kernels_gemm_tile.o april_asr_amd/csrc/build/kernels_gemm_kw.o \
//        april_asr_amd/csrc/build/kernels_recur.o april_asr_amd/csrc/build/kernels_misc.o -o tools/kw_bench
// usage: tools/kw_bench [iters=200] [only_shape]
#include "kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace aprilx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
template <class T> static T *dalloc(size_t n) { T *p; CK(hipMalloc((void **)&p, n * sizeof(T))); return p; }

struct Problem {
    int M, N, K, kz, epi; bool a_indexed = false;      // a_indexed: the activation rows are reached through the row -> slot indirection (aidx0)
    float *a, *w, *bias, *resid, *ssq_in, *out, *state, *ssq_out, *cst = nullptr, *cst0 = nullptr;      // cst: cell state (gates), cst0: its initial values
    int *slots;
    float *ks_ws = nullptr; unsigned *ks_cnt = nullptr;      // workspace of the K-cut stream kernels (<= 16 rows; kernels_recur.hip)
};
static void fill(std::vector<float> &h, unsigned seed, float scale)
{
    unsigned s = seed * 2654435761u + 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((float)((s >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }
}
static float *upload(const std::vector<float> &h) { float *p = dalloc<float>(h.size()); CK(hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return p; }
static Problem make_problem(int M, int N, int K, int kz, int epi, unsigned seed)
{
    Problem p{M, N, K, kz, epi};
    std::vector<float> h;
    h.resize((size_t)M * K); fill(h, seed, 2.0f); p.a = upload(h);
    h.resize((size_t)K * N); fill(h, seed + 1, 0.1f); p.w = upload(h);
    h.resize((size_t)N); fill(h, seed + 2, 1.0f); p.bias = upload(h);
    h.resize((size_t)M * N); fill(h, seed + 3, 1.0f); p.resid = upload(h);
    h.resize((size_t)M * (N / 32)); fill(h, seed + 4, 1.0f); for (auto &v : h) v = v * v + 0.1f; p.ssq_in = upload(h);
    p.out = dalloc<float>((size_t)M * N); p.state = dalloc<float>((size_t)M * N); p.ssq_out = dalloc<float>((size_t)M * (N / 32));
    std::vector<int> perm((size_t)M); for (int i = 0; i < M; ++i) perm[(size_t)i] = i;
    unsigned s = seed; for (int i = M - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; std::swap(perm[(size_t)i], perm[(size_t)((s >> 8) % (unsigned)(i + 1))]); }
    p.slots = dalloc<int>((size_t)M); CK(hipMemcpy(p.slots, perm.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    if (M <= 16 && (epi == EPI_HR || epi == EPI_RESID_SSQ)) { p.ks_ws = dalloc<float>((size_t)N * kz * 16); p.ks_cnt = dalloc<unsigned>((size_t)N / 16); CK(hipMemset(p.ks_cnt, 0, (size_t)N / 16 * 4)); }
    if (epi == EPI_LSTM) {      // gates: A = [y (M x K/2, rows) | h (M x K/2, by slot)], c state [M][N/4]
        h.resize((size_t)M * (N / 4)); fill(h, seed + 5, 1.0f); p.cst0 = upload(h); p.cst = dalloc<float>(h.size());
        h.resize((size_t)M * (K / 64)); fill(h, seed + 6, 1.0f); for (auto &v : h) v = v * v + 0.1f;
        CK(hipFree(p.ssq_in)); p.ssq_in = upload(h);
    }
    return p;
}
static bool g_ks_attach = false;      // hand the K-cut workspace to the GEMMs (the engine always does; off for the reference runs)
static GemmArgs gemm_of(const Problem &p, int zcount, int tile_ok = 0)
{
    GemmArgs g; g.tile_ok = tile_ok;
    g.a0 = p.a; g.lda0 = p.K; g.K0 = p.K; g.aidx0 = p.a_indexed ? p.slots : nullptr; g.wp = p.w; g.M = p.M; g.N = p.N; g.K = p.K; g.kz = p.kz; g.zcount = zcount;
    g.epi = p.epi; g.out = p.out; g.ldo = p.N; g.bias = p.bias;
    if (p.epi == EPI_LSTM) {
        g.K0 = p.K / 2; g.lda0 = p.K / 2; g.a1 = p.a + (size_t)p.M * (p.K / 2); g.lda1 = p.K / 2; g.aidx1 = p.slots; g.K1 = p.K / 2; g.aidx0 = nullptr;
        g.c_state = p.cst; g.slot_idx = p.slots; g.hidden = p.N / 4; g.ldo = p.N / 4;
        g.x_scale.ssq = p.ssq_in; g.x_scale.groups = (p.K / 2) / 32; g.x_scale.inv_n = 1.0f / (p.K / 2); g.x_scale.eps = 0.25f;
    }
    else if (p.epi == EPI_HR) { g.state = p.state; g.ld_state = p.N; g.slot_idx = p.slots; g.resid = p.resid; g.ldr = p.N;
                           g.r_scale.ssq = p.ssq_in; g.r_scale.groups = p.N / 32; g.r_scale.inv_n = 1.0f / p.N; g.r_scale.eps = 0.25f; g.bias = nullptr; g.force_fullk = 1; }
    else if (p.epi == EPI_RESID_SSQ) { g.resid = p.resid; g.ldr = p.N; g.ssq_out = p.ssq_out; g.force_fullk = 1; }
    if (g_ks_attach) { g.ks_ws = p.ks_ws; g.ks_cnt = p.ks_cnt; }
    return g;
}
struct Chain {
    std::vector<GemmArgs> gh; GemmArgs *gd = nullptr; int n = 0;
    void run(hipStream_t s) const { if (n == 1) launch_gemm(gh[0], s); else launch_gemm_z(gh.data(), n, gd, s); }
};
static Chain make_chain(const std::vector<Problem> &ps, int tile_ok = 0)
{
    Chain c; c.n = (int)ps.size();
    std::vector<GemmArgs> items;
    for (const Problem &p : ps) items.push_back(gemm_of(p, c.n, tile_ok));
    if (c.n == 1) { c.gh = items; GemmArgs probe; stage_gemm_z(items.data(), 1, &probe); c.gh[0].mode = probe.mode; }      // (launch_gemm plans again; the mode is for the report)
    else {
        c.gh.resize(items.size()); stage_gemm_z(items.data(), c.n, c.gh.data());
        c.gd = dalloc<GemmArgs>(items.size()); CK(hipMemcpy(c.gd, c.gh.data(), items.size() * sizeof(GemmArgs), hipMemcpyHostToDevice));
    }
    return c;
}
static std::vector<float> snapshot(const std::vector<Problem> &ps)
{
    std::vector<float> all;
    for (const Problem &p : ps) {
        std::vector<float> h((size_t)p.M * (p.epi == EPI_LSTM ? p.N / 4 : p.N));
        CK(hipMemcpy(h.data(), p.out, h.size() * 4, hipMemcpyDeviceToHost)); all.insert(all.end(), h.begin(), h.end());
        if (p.epi == EPI_LSTM) { CK(hipMemcpy(h.data(), p.cst, h.size() * 4, hipMemcpyDeviceToHost)); all.insert(all.end(), h.begin(), h.end()); }
        if (p.epi == EPI_HR) { CK(hipMemcpy(h.data(), p.state, h.size() * 4, hipMemcpyDeviceToHost)); all.insert(all.end(), h.begin(), h.end()); }
        if (p.epi == EPI_RESID_SSQ) { h.resize((size_t)p.M * (p.N / 32)); CK(hipMemcpy(h.data(), p.ssq_out, h.size() * 4, hipMemcpyDeviceToHost)); all.insert(all.end(), h.begin(), h.end()); }
    }
    return all;
}
static void clear_outputs(const std::vector<Problem> &ps)
{
    for (const Problem &p : ps) { if (p.cst) CK(hipMemcpy(p.cst, p.cst0, (size_t)p.M * (p.N / 4) * 4, hipMemcpyDeviceToDevice));      // (the cell state is updated in place: same start for every run)
                                  CK(hipMemset(p.out, 0xff, (size_t)p.M * p.N * 4)); CK(hipMemset(p.state, 0xff, (size_t)p.M * p.N * 4)); CK(hipMemset(p.ssq_out, 0xff, (size_t)p.M * (p.N / 32) * 4)); }
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
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    hipStream_t s; CK(hipStreamCreate(&s));
    struct Shape { const char *name; int M, N, K, kz, epi, n; bool idx = false; };
    const Shape shapes[] = {
        {"proj    1x1", 1, 512, 1024, 4, EPI_HR, 1},                     // one session streaming: 1 .. 3 problems of one row
        {"proj    1x2", 1, 512, 1024, 4, EPI_HR, 2},
        {"proj    1x3", 1, 512, 1024, 4, EPI_HR, 3},
        {"ffdn    1x1", 1, 512, 2048, 8, EPI_RESID_SSQ, 1},
        {"ffdn    1x2", 1, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"ffdn    1x3", 1, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"ffdn    2x3", 2, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"proj   1x12", 1, 512, 1024, 4, EPI_HR, 12},                    // the recurrent pair of a long feed: twelve layers per launch
        {"proj    4x3", 4, 512, 1024, 8, EPI_HR, 3},                     // <= 16 rows: the reference is the weight-stream kernel of kernels_recur.hip
        {"ffdn    8x2", 8, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"proj   16x3", 16, 512, 1024, 8, EPI_HR, 3},
        {"ffdn   16x3", 16, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"ffdn   13x1", 13, 512, 2048, 8, EPI_RESID_SSQ, 1},
        {"gates  64x1", 64, 4096, 1024, 1, EPI_LSTM, 1},
        {"gates  64x2", 64, 4096, 1024, 1, EPI_LSTM, 2},
        {"gates  64x3", 64, 4096, 1024, 1, EPI_LSTM, 3},
        {"gates  24x3", 24, 4096, 1024, 1, EPI_LSTM, 3},
        {"gates  40x2", 40, 4096, 1024, 1, EPI_LSTM, 2},
        {"gates  96x2", 96, 4096, 1024, 1, EPI_LSTM, 2},
        {"gates 128x2", 128, 4096, 1024, 1, EPI_LSTM, 2},
        {"gates 128x3", 128, 4096, 1024, 1, EPI_LSTM, 3},
        {"gates 256x2", 256, 4096, 1024, 1, EPI_LSTM, 2},
        {"gates  50x2 L", 50, 6144, 1536, 1, EPI_LSTM, 2},              // larger encoder dims
        {"proj  256x1", 256, 512, 1024, 8, EPI_HR, 1},
        {"proj  256x2", 256, 512, 1024, 8, EPI_HR, 2},
        {"proj  256x3", 256, 512, 1024, 8, EPI_HR, 3},
        {"ffdn  256x1", 256, 512, 2048, 8, EPI_RESID_SSQ, 1},
        {"ffdn  256x2", 256, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"ffdn  256x3", 256, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"ffup  256x1", 256, 2048, 512, 1, EPI_BIAS_DSWISH, 1},
        {"ffup  256x2", 256, 2048, 512, 1, EPI_BIAS_DSWISH, 2},
        {"ffup  256x3", 256, 2048, 512, 1, EPI_BIAS_DSWISH, 3},
        {"proj  250x2", 250, 512, 1024, 8, EPI_HR, 2},                  // ragged rows
        {"ffdn   40x2", 40, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"proj  200x2 i", 200, 512, 1024, 8, EPI_HR, 2, true},          // activation rows through the slot indirection, ragged
        {"ffup   70x3", 70, 2048, 512, 1, EPI_BIAS_DSWISH, 3},
        {"proj   64x3", 64, 512, 1024, 8, EPI_HR, 3},
        {"ffdn   64x3", 64, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"proj  128x3", 128, 512, 1024, 8, EPI_HR, 3},
        {"ffdn  128x3", 128, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"ffup  128x3", 128, 2048, 512, 1, EPI_BIAS_DSWISH, 3},
        {"proj  512x2", 512, 512, 1024, 8, EPI_HR, 2},
        {"ffdn  512x2", 512, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"ffup  512x2", 512, 2048, 512, 1, EPI_BIAS_DSWISH, 2},
        {"ffdn  768x3", 768, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"proj 1024x2", 1024, 512, 1024, 8, EPI_HR, 2},
        {"ffdn 1024x2", 1024, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"ffdn 2048x2", 2048, 512, 2048, 8, EPI_RESID_SSQ, 2},
        {"proj 2048x2", 2048, 512, 1024, 8, EPI_HR, 2},
        {"ffdn 2300x2", 2300, 512, 2048, 8, EPI_RESID_SSQ, 2},          // ragged 64-row tiles
        {"proj 2048x3", 2048, 512, 1024, 8, EPI_HR, 3},
        {"ffdn 2048x3", 2048, 512, 2048, 8, EPI_RESID_SSQ, 3},
        {"proj  512x3 L", 512, 768, 1536, 2, EPI_HR, 3},                // larger encoder (configs[4] dims), fp32
        {"ffdn  512x3 L", 512, 768, 3072, 2, EPI_RESID_SSQ, 3},
        {"ffup  512x3 L", 512, 3072, 768, 1, EPI_BIAS_DSWISH, 3},
    };
    int bad = 0;
    for (const Shape &sh : shapes) {
        if (only >= 0 && (&sh - shapes) != only) continue;
        std::vector<Problem> ps;
        for (int i = 0; i < sh.n; ++i) { ps.push_back(make_problem(sh.M, sh.N, sh.K, sh.kz, sh.epi, 1000u * (unsigned)(&sh - shapes) + 10u * (unsigned)i)); ps.back().a_indexed = sh.idx; }
        const double flops = 2.0 * sh.M * sh.N * sh.K * sh.n;
        gemm_kw_pin(0, 0, 0);
        Chain ref = make_chain(ps);
        clear_outputs(ps); ref.run(s); CK(hipStreamSynchronize(s));
        const std::vector<float> want = snapshot(ps);
        const double t_ref = time_chain(ref, s, iters);
        printf("%-13s M=%d N=%d K=%d kz=%d x%d | round-4 schedule (mode %d): %7.2f us (%.3f of peak)\n", sh.name, sh.M, sh.N, sh.K, sh.kz, sh.n, ref.gh[0].mode, t_ref, flops / (t_ref * 1e-6) / 157.3e12);
        if (sh.M <= 16 && (sh.epi == EPI_HR || sh.epi == EPI_RESID_SSQ)) {
            // the stream kernel with K cut across workgroups (in-launch hand-over, kernels_recur.hip): planner's cut, then every pinned one
            for (int cut : {1, 2, 4, 8}) {
                if (cut > sh.kz) continue;
                g_ks_attach = true; recur_ksplit_pin(cut); gemm_kw_pin(0, 0, 0);
                Chain c = make_chain(ps);
                g_ks_attach = false;
                const int got_cut = c.n == 1 ? recur_ksplit(c.gh[0], 1) : c.gh[0].ksplit;      // (launch_gemm plans a single problem again at every launch: the pin stays)
                clear_outputs(ps); c.run(s); CK(hipStreamSynchronize(s));
                const std::vector<float> got = snapshot(ps);
                size_t diff = 0; for (size_t i = 0; i < want.size(); ++i) if (memcmp(&want[i], &got[i], 4) != 0) ++diff;
                const double t = time_chain(c, s, iters);
                clear_outputs(ps); for (int i = 0; i < 7; ++i) c.run(s); CK(hipStreamSynchronize(s));      // back-to-back: the counters re-arm
                recur_ksplit_pin(-1);
                const std::vector<float> got2 = snapshot(ps);
                size_t diff2 = 0; for (size_t i = 0; i < want.size(); ++i) if (memcmp(&want[i], &got2[i], 4) != 0) ++diff2;
                printf("    stream kernel, K cut %s-> %d workgroups per granule : %7.2f us (%.2fx)  %s\n", cut == 1 ? "(planner) " : "", got_cut, t, t_ref / t, (diff || diff2) ? "MISMATCH" : "bit-identical");
                if (diff || diff2) { ++bad; printf("      mismatching floats: %zu / %zu of %zu\n", diff, diff2, want.size()); }
            }
        }
        if (sh.epi != EPI_BIAS_DSWISH && gemm_tile_planned(sh.M, sh.N, sh.kz, sh.n) && gemm_fullk(sh.M, sh.N, sh.kz, true, sh.n, 1)) {
            // GM_TILE with the row work fused, where the engine's planner would take it (for the crossover between the two)
            gemm_kw_pin(0, 0, 0);
            Chain ct = make_chain(ps, 1);
            if (ct.gh[0].mode == GM_TILE) {
                clear_outputs(ps); ct.run(s); CK(hipStreamSynchronize(s));
                const std::vector<float> got = snapshot(ps);
                size_t diff = 0; for (size_t i = 0; i < want.size(); ++i) if (memcmp(&want[i], &got[i], 4) != 0) ++diff;
                const double t = time_chain(ct, s, iters);
                printf("    GM_TILE (fused, planner) : %7.2f us (%.3f of peak, %.2fx)  %s\n", t, flops / (t * 1e-6) / 157.3e12, t_ref / t, diff ? "MISMATCH" : "bit-identical");
                if (diff) ++bad;
            }
        }
        for (int mt : {0, 2, 1, 4}) {
            if ((mt == 1 || mt == 4) && sh.epi == EPI_BIAS_DSWISH) continue;
            if (mt == 1 && sh.epi == EPI_LSTM) continue;
            gemm_kw_pin(1, mt, 1);
            Chain c = make_chain(ps, mt == 0 ? 1 : 0);      // (the planner's own choice as the engine makes it: GM_TILE allowed, GM_KW may take a fused GM_TILE plan over)
            if (c.gh[0].mode != GM_KW) { printf("    (GM_KW not planned for this shape, mt pin %d%s)\n", mt, c.gh[0].mode == GM_TILE ? ": the planner keeps GM_TILE" : ""); continue; }
            clear_outputs(ps); c.run(s); CK(hipStreamSynchronize(s));
            const std::vector<float> got = snapshot(ps);
            size_t diff = 0; double maxd = 0;
            for (size_t i = 0; i < want.size(); ++i) if (memcmp(&want[i], &got[i], 4) != 0) { ++diff; maxd = std::max(maxd, (double)fabsf(want[i] - got[i])); }
            const double t = time_chain(c, s, iters);
            clear_outputs(ps); for (int i = 0; i < (sh.epi == EPI_LSTM ? 1 : 5); ++i) c.run(s); CK(hipStreamSynchronize(s));      // (the gates update the cell state in place: one run)
            const std::vector<float> got2 = snapshot(ps);
            size_t diff2 = 0; for (size_t i = 0; i < want.size(); ++i) if (memcmp(&want[i], &got2[i], 4) != 0) ++diff2;
            if (getenv("KB_TRACE")) {      // (binary built with -DAPRIL_GEMM_TRACE) s_memtime stamps of wave 0: start, loop start, loop end, meet done, end
                const int nwg = 4096;
                unsigned long long *tr = dalloc<unsigned long long>((size_t)nwg * 16); CK(hipMemset(tr, 0, (size_t)nwg * 16 * 8));
                Chain ct = c; for (auto &g : ct.gh) g.trace = tr;
                if (ct.n > 1) CK(hipMemcpy(ct.gd, ct.gh.data(), ct.gh.size() * sizeof(GemmArgs), hipMemcpyHostToDevice));
                ct.run(s); CK(hipStreamSynchronize(s));
                std::vector<unsigned long long> ht((size_t)nwg * 16); CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
                double ph[4] = {0, 0, 0, 0}, lastw = 0, pa[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int nw = 0; unsigned long long t0 = ~0ull, t1 = 0;
                for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 16 + 4]) { ++nw; for (int k = 0; k < 4; ++k) ph[k] += (double)(ht[(size_t)i * 16 + k + 1] - ht[(size_t)i * 16 + k]); lastw += (double)(ht[(size_t)i * 16 + 5] - ht[(size_t)i * 16 + 1]);
                    for (int k = 0; k < 10; ++k) pa[k] += (double)ht[(size_t)i * 16 + 6 + k]; t0 = std::min(t0, ht[(size_t)i * 16]); t1 = std::max(t1, ht[(size_t)i * 16 + 4]); }
                if (nw) printf("      loop phases, ticks per workgroup: wave 0: reads + k block 0 %.0f | LDS wait %.0f | DMA + B issue %.0f | vmcnt wait %.0f | k block 1 + B issue + fold %.0f ;  wave 4: %.0f | %.0f | %.0f | %.0f | %.0f\n",
                               pa[0] / nw, pa[1] / nw, pa[2] / nw, pa[3] / nw, pa[4] / nw, pa[5] / nw, pa[6] / nw, pa[7] / nw, pa[8] / nw, pa[9] / nw);
                if (nw) printf("      trace (%d workgroups; s_memtime ticks): prologue %.0f | K loop of wave 0 %.0f (last wave %.0f) | meet %.0f | epilogue %.0f ; first start -> last end %.0f\n", nw, ph[0] / nw, ph[1] / nw, lastw / nw, ph[2] / nw, ph[3] / nw, (double)(t1 - t0));
                if (ct.n > 1) { for (auto &g : ct.gh) g.trace = nullptr; CK(hipMemcpy(ct.gd, c.gh.data(), c.gh.size() * sizeof(GemmArgs), hipMemcpyHostToDevice)); }
            }
            printf("    GM_KW mt=%d%s : %7.2f us (%.3f of peak, %.2fx)  %s\n", c.gh[0].M > 0 ? mt : mt, mt == 0 ? " (planner)" : "", t, flops / (t * 1e-6) / 157.3e12, t_ref / t, (diff || diff2) ? "MISMATCH" : "bit-identical");
            if (diff || diff2) { ++bad; printf("      mismatching floats: %zu / %zu of %zu (max |d| %.3g)\n", diff, diff2, want.size(), maxd); }
        }
        fflush(stdout);
    }
    gemm_kw_pin(-1, 0, -1);
    printf(bad ? "FAILED: %d configurations differ\n" : "all configurations bit-identical\n", bad);
    return bad ? 1 : 0;
}
