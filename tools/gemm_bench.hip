// Measurement aid: times launch_gemm (the product's GEMM kernel, linked from csrc/build/kernels_gemm.o) on one shape.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DAPRIL_GEMM_TRACE] -Iapril_asr_amd/csrc -c april_asr_amd/csrc/kernels_gemm.hip -o kg.o
//          hipcc --offload-arch=gfx950 -O2 -Iapril_asr_amd/csrc -c tools/gemm_bench.hip -o gb.o && hipcc --offload-arch=gfx950 gb.o kg.o -o tools/gemm_bench
//   usage: tools/gemm_bench M N K [epi=2 (bias+dswish) | 0 (partials, kz given) | 1 (gates + LSTM cell, A scaled on load) |
//                                  3 (projection: state + residual) | 4 (bias + residual + sums of squares) | 5 (slot store)] [kz=1] [iters=200] [wcopies=1]
//   epi 3..5 run on the full-K schedule when launch_gemm's planner allows it for the shape, otherwise this tool falls back to epi 0
//   wcopies > 1 cycles through that many copies of W so the weights stream from HBM as in the product (12 layers x 28 MB)
#include "kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace aprilx;

int main(int argc, char **argv)
{
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int epi = argc > 4 ? atoi(argv[4]) : 2, kz = argc > 5 ? atoi(argv[5]) : 1, iters = argc > 6 ? atoi(argv[6]) : 200, wc = argc > 7 ? atoi(argv[7]) : 1;
    float *a, *w, *out, *bias;
    hipMalloc(&a, (size_t)M * K * 4); hipMalloc(&w, (size_t)K * N * 4 * wc); hipMalloc(&out, (size_t)8 * M * N * 4); hipMalloc(&bias, (size_t)N * 4);
    std::vector<float> h((size_t)std::max((size_t)M * K, (size_t)K * N));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(a, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < wc; ++i) hipMemcpy(w + (size_t)i * K * N, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, (size_t)N * 4);
    GemmArgs g;
    g.a0 = a; g.lda0 = K; g.K0 = K; g.wp = w; g.M = M; g.N = N; g.K = K; g.kz = kz; g.epi = epi;
    g.out = out; g.ldo = N; g.m_stride = M; g.bias = bias;
    float *ssq; hipMalloc(&ssq, (size_t)M * (K / 32 + N / 32 + 2) * 4);
    { std::vector<float> one((size_t)M * (K / 32 + N / 32 + 2), 1.0f); hipMemcpy(ssq, one.data(), one.size() * 4, hipMemcpyHostToDevice); }
    int *slots; hipMalloc(&slots, (size_t)M * 4);
    { std::vector<int> hi((size_t)M); for (int i = 0; i < M; ++i) hi[(size_t)i] = (int)(((unsigned)i * 2654435761u) % (unsigned)M); hipMemcpy(slots, hi.data(), (size_t)M * 4, hipMemcpyHostToDevice); }
    if (epi >= 3) {
        if (!gemm_fullk(M, N, kz, false)) { printf("(no full-K plan for M=%d N=%d kz=%d: timing the split-K partial GEMM instead)\n", M, N, kz); g.epi = 0; }
        else {
            float *st, *res; hipMalloc(&st, (size_t)M * N * 4); hipMalloc(&res, (size_t)M * N * 4); hipMemset(res, 0, (size_t)M * N * 4);
            g.slot_idx = slots; g.state = st; g.ld_state = N; g.resid = res; g.ldr = N; g.ssq_out = ssq;
            g.r_scale.ssq = ssq; g.r_scale.groups = N / 32; g.r_scale.inv_n = 1.0f / N; g.r_scale.eps = 0.25f;
        }
    }
    if (epi == 1) {   // the product's gates call: A = [x | h[slot]] in two K segments, fused LSTM cell (c in place, u out)
        int *idx; hipMalloc(&idx, (size_t)M * 4);
        std::vector<int> hi((size_t)M);
        for (int i = 0; i < M; ++i) hi[(size_t)i] = (int)(((unsigned)i * 2654435761u) % (unsigned)M);   // scattered slots
        hipMemcpy(idx, hi.data(), (size_t)M * 4, hipMemcpyHostToDevice);
        float *cst; hipMalloc(&cst, (size_t)M * (N / 4) * 4); hipMemset(cst, 0, (size_t)M * (N / 4) * 4);
        g.K0 = K / 2; g.lda0 = K / 2; g.a1 = a + (size_t)M * (K / 2); g.lda1 = K / 2; g.aidx1 = idx; g.K1 = K / 2;
        g.c_state = cst; g.slot_idx = idx; g.hidden = N / 4; g.ldo = N / 4;
        g.x_scale.ssq = ssq; g.x_scale.groups = (K / 2) / 32; g.x_scale.inv_n = 1.0f / (K / 2); g.x_scale.eps = 0.25f;
    }
    if (getenv("GB_TILE_OK")) g.tile_ok = atoi(getenv("GB_TILE_OK"));      // 2 = the GM_TILE form (the engine's choice for gates launches from 2048 rows)
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch_gemm(g, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) { g.wp = w + (size_t)(i % wc) * K * N; launch_gemm(g, s); }
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (getenv("GEMM_TRACE")) {   // one extra launch with per-workgroup s_memtime stamps
        const int nwg = 8192;
        unsigned long long *tr; hipMalloc(&tr, (size_t)nwg * 8 * 8); hipMemset(tr, 0, (size_t)nwg * 8 * 8);
        g.trace = tr; g.wp = w; launch_gemm(g, s); hipStreamSynchronize(s); g.trace = nullptr;
        std::vector<unsigned long long> ht((size_t)nwg * 8);
        hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost);
        if (getenv("GB_TILE_OK")) {   // GM_TILE (kernels_gemm_tile.hip built with -DAPRIL_GEMM_TRACE): per-workgroup SUMS of phase intervals, [7] = stages
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nw = 0;
            for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 8 + 7]) { ++nw; for (int k = 0; k < 8; ++k) acc[k] += (double)ht[(size_t)i * 8 + k]; }
            if (nw) printf("  tile trace (%d workgroups, %.0f stages each; s_memtime ticks per stage): first k block %.0f | chunk ends %.0f | wait+barrier %.0f | issue + second k block %.0f | loop total %.0f ; fragment waits + prologue %.0f, fragment waits + epilogue %.0f ticks per workgroup\n",
                           nw, acc[7] / nw, acc[0] / acc[7], acc[1] / acc[7], acc[2] / acc[7], acc[3] / acc[7], acc[4] / acc[7], acc[5] / nw, acc[6] / nw);
            return 0;
        }
        unsigned long long t0 = ~0ull, t1 = 0; int n = 0;
        for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 8]) { ++n; t0 = std::min(t0, ht[(size_t)i * 8]); t1 = std::max(t1, ht[(size_t)i * 8 + 4]); }
        double seg[5] = {0, 0, 0, 0, 0}, start_spread = 0;
        for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 8]) {
            const unsigned long long *q = &ht[(size_t)i * 8];
            seg[0] += (double)(q[1] - q[0]); seg[1] += (double)(q[2] - q[1]); seg[2] += (double)(q[3] - q[2]); seg[3] += (double)(q[4] - q[3]);
            start_spread += (double)(q[0] - t0);
        }
        if (getenv("GEMM_TRACE_CU")) {   // timeline of the workgroups that ran on one CU (same XCC clock): ticks relative to the first start there
            // HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se ; XCC_ID [3:0]
            auto key = [&](int i) { const unsigned long long v = ht[(size_t)i * 8 + 5]; const unsigned hw = (unsigned)v; return ((v >> 32) & 15) << 16 | (hw & 0xff00); };
            int pick = -1; for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 8]) { pick = i; break; }
            const unsigned long long k0 = key(pick);
            std::vector<int> on; for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 8] && key(i) == k0) on.push_back(i);
            std::sort(on.begin(), on.end(), [&](int a, int b) { return ht[(size_t)a * 8] < ht[(size_t)b * 8]; });
            const unsigned long long base = ht[(size_t)on[0] * 8];
            {   // all CUs of the same XCC (one clock domain): whole-XCC span and the spread of per-CU spans
                const unsigned long long xk = k0 >> 16;
                std::vector<unsigned long long> keys; unsigned long long lo = ~0ull, hi = 0;
                for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 8] && (key(i) >> 16) == xk) { keys.push_back(key(i)); lo = std::min(lo, ht[(size_t)i * 8]); hi = std::max(hi, ht[(size_t)i * 8 + 4]); }
                std::sort(keys.begin(), keys.end()); keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
                std::vector<unsigned long long> spans, starts, ends;
                for (unsigned long long k : keys) { unsigned long long a = ~0ull, b = 0; for (int i = 0; i < nwg; ++i) if (ht[(size_t)i * 8] && key(i) == k) { a = std::min(a, ht[(size_t)i * 8]); b = std::max(b, ht[(size_t)i * 8 + 4]); } spans.push_back(b - a); starts.push_back(a - lo); ends.push_back(b - lo); }
                std::sort(spans.begin(), spans.end()); std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
                printf("  XCC %llu: %zu CUs, whole span %llu cycles; per-CU span min/med/max %llu/%llu/%llu; first start of a CU min/med/max %llu/%llu/%llu; last end min/med/max %llu/%llu/%llu\n",
                       xk, keys.size(), hi - lo, spans.front(), spans[spans.size() / 2], spans.back(), starts.front(), starts[starts.size() / 2], starts.back(), ends.front(), ends[ends.size() / 2], ends.back());
            }
            printf("  CU key %llx: %zu workgroups\n", k0, on.size());
            for (int i : on) { const unsigned long long *q = &ht[(size_t)i * 8];
                printf("    wg %5d  start %7llu  loop %7llu..%7llu  meet-end %7llu  end %7llu\n", i, q[0] - base, q[1] - base, q[2] - base, q[3] - base, q[4] - base); }
        }
        printf("  trace: %d workgroups, span %llu ticks; mean ticks: setup %.0f | K loop %.0f | meet %.0f | epilogue %.0f | start after first %.0f\n",
               n, t1 - t0, seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n, start_spread / n);
    }
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("M=%d N=%d K=%d epi=%d kz=%d wc=%d : %.2f us/launch (back-to-back)  %.1f TFLOP/s  %.1f%% of 157.3\n", M, N, K, epi, kz, wc, us, tf, tf / 157.3 * 100);
    return 0;
}
