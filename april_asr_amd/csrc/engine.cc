// See engine.h.
#include "engine.h"
#include <array>
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace aprilx {

// ---------------------------------------------------------------- packing
static size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

void plan_layout(const NetDims &d, bool has_dec_conv_b, PackedLayout &L)
{
    L.dims = d;
    L.has_dec_conv_b = has_dec_conv_b;
    L.vocab_pad = (d.vocab + 31) & ~31;            // a multiple of 32: the joiner GEMM's full-K tiles are 32 columns wide
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off = align64(off + n); return o; };
    // conv 0/1: OIHW as exported; conv 2 runs as an im2col GEMM: MFMA-packed [k3][conv_ch[2]], k = ci*9 + i*3 + j
    L.k3 = (d.conv_ch[1] * 9 + 63) & ~63;          // K of every GEMM is a multiple of 64 (4 waves x 16-wide blocks)
    L.conv_w[0] = take((size_t)d.conv_ch[0] * 9); L.conv_b[0] = take((size_t)d.conv_ch[0]);
    L.conv_w[1] = take((size_t)d.conv_ch[1] * d.conv_ch[0] * 9); L.conv_b[1] = take((size_t)d.conv_ch[1]);
    L.conv_w[2] = take((size_t)L.k3 * d.conv_ch[2]); L.conv_b[2] = take((size_t)d.conv_ch[2]);
    L.w_embed = take((size_t)d.embed_in * d.d_model); L.b_embed = take((size_t)d.d_model);
    L.layers.resize((size_t)d.n_layers);
    for (auto &l : L.layers) {
        l.wg = take((size_t)2 * d.d_model * 4 * d.hidden); l.bg = take((size_t)4 * d.hidden);
        l.whr = take((size_t)d.hidden * d.d_model);
        l.wff1 = take((size_t)d.d_model * d.ffn); l.bff1 = take((size_t)d.ffn);
        l.wff2 = take((size_t)d.ffn * d.d_model); l.bff2 = take((size_t)d.d_model);
    }
    L.w_encproj = take((size_t)d.d_model * d.joiner); L.b_encproj = take((size_t)d.joiner);
    L.emb = take((size_t)d.vocab * d.d_model);
    L.dec_conv = take((size_t)d.d_model * (d.d_model / d.dec_groups) * d.context);
    L.dec_conv_b = take((size_t)d.d_model);
    L.w_decproj = take((size_t)d.d_model * d.joiner); L.b_decproj = take((size_t)d.joiner);
    L.w_out = take((size_t)d.joiner * L.vocab_pad); L.b_out = take((size_t)L.vocab_pad);
    L.total = off;
}

std::vector<std::pair<size_t, size_t>> gemm_sections(const PackedLayout &L)
{
    const NetDims &d = L.dims;
    std::vector<std::pair<size_t, size_t>> v;
    v.emplace_back(L.w_embed, (size_t)d.embed_in * d.d_model);
    for (const PackedLayout::Layer &o : L.layers) {
        v.emplace_back(o.wg, (size_t)2 * d.d_model * 4 * d.hidden); v.emplace_back(o.whr, (size_t)d.hidden * d.d_model);
        v.emplace_back(o.wff1, (size_t)d.d_model * d.ffn); v.emplace_back(o.wff2, (size_t)d.ffn * d.d_model);
    }
    v.emplace_back(L.w_encproj, (size_t)d.d_model * d.joiner); v.emplace_back(L.w_decproj, (size_t)d.d_model * d.joiner);
    v.emplace_back(L.w_out, (size_t)d.joiner * L.vocab_pad);
    return v;
}

// W is K x N row-major; see kernels.h for the packed order.  colmap (optional) gives, for each
// packed column, the source column.
static void pack_mfma(const float *W, int K, int N, int Npad, const std::vector<int> *colmap, float *dst)
{
    const int KB = K / 16, NTt = Npad / 16;
    for (int t = 0; t < NTt; ++t)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane) {
                const int np = t * 16 + (lane & 15);
                const int src = np < N ? (colmap ? (*colmap)[(size_t)np] : np) : -1;
                float *o = dst + (((size_t)t * KB + kb) * 64 + lane) * 4;
                for (int j = 0; j < 4; ++j) {
                    const int k = kb * 16 + (lane >> 4) * 4 + j;
                    o[j] = src >= 0 ? W[(size_t)k * N + src] : 0.0f;
                }
            }
}

void pack_weights(const HostModel &m, PackedLayout &L, std::vector<float> &blob)
{
    const NetDims &d = L.dims;
    blob.assign(L.total, 0.0f);
    float *B = blob.data();
    for (int i = 0; i < 3; ++i) {
        if (i < 2) memcpy(B + L.conv_w[i], m.conv_w[i].data(), m.conv_w[i].size() * 4);
        memcpy(B + L.conv_b[i], m.conv_b[i].data(), m.conv_b[i].size() * 4);
    }
    {   // third conv as a K x N matrix: W3[k][o] = w[o][ci][i][j], k = ci*9 + i*3 + j (zero rows up to k3)
        const int kreal = d.conv_ch[1] * 9, c2 = d.conv_ch[2];
        std::vector<float> w3((size_t)L.k3 * c2, 0.0f);
        for (int o = 0; o < c2; ++o) for (int k = 0; k < kreal; ++k) w3[(size_t)k * c2 + o] = m.conv_w[2][(size_t)o * kreal + k];
        pack_mfma(w3.data(), L.k3, c2, c2, nullptr, B + L.conv_w[2]);
    }
    {   // the conv GEMM leaves its output position-major ([f][c]); the graph flattens channel-major (c*f_out + f):
        // permute the embed linear's input rows instead of transposing activations
        const int c2 = d.conv_ch[2], F = d.f_out;
        std::vector<float> we((size_t)d.embed_in * d.d_model);
        for (int c = 0; c < c2; ++c) for (int f = 0; f < F; ++f)
            memcpy(&we[((size_t)f * c2 + c) * d.d_model], &m.w_embed[((size_t)c * F + f) * d.d_model], (size_t)d.d_model * 4);
        pack_mfma(we.data(), d.embed_in, d.d_model, d.d_model, nullptr, B + L.w_embed);
    }
    memcpy(B + L.b_embed, m.b_embed.data(), (size_t)d.d_model * 4);
    L.embed_eps = m.embed_norm_eps;
    L.norm_eps.resize((size_t)d.n_layers);
    // gate columns become unit-major: packed column u*4+g <- source column g*hidden+u
    std::vector<int> gmap((size_t)4 * d.hidden);
    for (int u = 0; u < d.hidden; ++u) for (int g = 0; g < 4; ++g) gmap[(size_t)u * 4 + g] = g * d.hidden + u;
    for (int l = 0; l < d.n_layers; ++l) {
        const LayerWeights &lw = m.layers[(size_t)l];
        const PackedLayout::Layer &o = L.layers[(size_t)l];
        pack_mfma(lw.w_gates.data(), 2 * d.d_model, 4 * d.hidden, 4 * d.hidden, &gmap, B + o.wg);
        for (int n = 0; n < 4 * d.hidden; ++n) B[o.bg + n] = lw.b_gates[(size_t)gmap[(size_t)n]];
        pack_mfma(lw.w_hr.data(), d.hidden, d.d_model, d.d_model, nullptr, B + o.whr);
        pack_mfma(lw.w_ff1.data(), d.d_model, d.ffn, d.ffn, nullptr, B + o.wff1);
        memcpy(B + o.bff1, lw.b_ff1.data(), (size_t)d.ffn * 4);
        pack_mfma(lw.w_ff2.data(), d.ffn, d.d_model, d.d_model, nullptr, B + o.wff2);
        memcpy(B + o.bff2, lw.b_ff2.data(), (size_t)d.d_model * 4);
        L.norm_eps[(size_t)l] = lw.norm_eps;
    }
    pack_mfma(m.w_encproj.data(), d.d_model, d.joiner, d.joiner, nullptr, B + L.w_encproj);
    memcpy(B + L.b_encproj, m.b_encproj.data(), (size_t)d.joiner * 4);
    memcpy(B + L.emb, m.emb.data(), m.emb.size() * 4);
    memcpy(B + L.dec_conv, m.dec_conv.data(), m.dec_conv.size() * 4);
    if (!m.dec_conv_b.empty()) memcpy(B + L.dec_conv_b, m.dec_conv_b.data(), m.dec_conv_b.size() * 4);
    pack_mfma(m.w_decproj.data(), d.d_model, d.joiner, d.joiner, nullptr, B + L.w_decproj);
    memcpy(B + L.b_decproj, m.b_decproj.data(), (size_t)d.joiner * 4);
    pack_mfma(m.w_out.data(), d.joiner, d.vocab, L.vocab_pad, nullptr, B + L.w_out);
    memcpy(B + L.b_out, m.b_out.data(), (size_t)d.vocab * 4);
}

// ---------------------------------------------------------------- engine
// (every allocation holds the process-wide legacy lock, engine.h: an allocation of one engine's stepping thread -- staging regrowth, a
// new launch plan -- must not coincide with another engine's graph capture; recursive, so the constructor's own guard nests)
template <typename T> static T *dmalloc(size_t n) { HipLegacyLock legacy; T *p = nullptr; HIP_CHECK(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T))); return p; }
template <typename T> static T *hmalloc(size_t n) { HipLegacyLock legacy; T *p = nullptr; HIP_CHECK(hipHostMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault)); return p; }

std::recursive_mutex &hip_legacy_mutex() { static std::recursive_mutex m; return m; }

static int pick_kz(int K, int N, int kblk = 16)
{
    int kz = 256 / std::max(1, N / 16);
    kz = std::max(1, std::min(kz, 8));
    while (kz > 1 && ((K / kblk) / kz < 4 || (K / kblk) % (4 * kz) != 0)) kz >>= 1;   // every wave gets whole blocks of every slab
    return kz;
}

Engine::Engine(const EngineConfig &cfg, const PackedLayout &layout, const float *blob_host, const float *blob_device,
               const ModelParams &params, const FbankHostTables &ft, const std::vector<uint8_t> &tok_class)
    : cfg_(cfg), L_(layout), P_(params)
{
    HipLegacyLock legacy;                      // (allocations, memsets and copies on the legacy stream: not while another engine captures a graph)
    HIP_CHECK(hipSetDevice(cfg_.device));
    {
        // APRIL_STREAM_PRIO (measurement): 1 = the layer stream at the device's highest priority, front end and search at the lowest;
        // 2 = the other way round; 0 = all at the default priority
        const char *e = getenv("APRIL_STREAM_PRIO");
        const int mode = e && *e ? atoi(e) : 0;
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (mode == 0 || least == greatest) {
            HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
            HIP_CHECK(hipStreamCreateWithFlags(&f_stream_, hipStreamNonBlocking));
            HIP_CHECK(hipStreamCreateWithFlags(&s_stream_, hipStreamNonBlocking));
        } else {
            const int pm = mode == 1 ? greatest : least, po = mode == 1 ? least : greatest;
            HIP_CHECK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, pm));
            HIP_CHECK(hipStreamCreateWithPriority(&f_stream_, hipStreamNonBlocking, po));
            HIP_CHECK(hipStreamCreateWithPriority(&s_stream_, hipStreamNonBlocking, po));
        }
    }
    search_stream_ = stream_;
    HIP_CHECK(hipStreamCreateWithFlags(&pf_stream_, hipStreamNonBlocking));
    for (hipEvent_t &e : pf_ev_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int i = 0; i < 32; ++i) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); join_ev_.push_back(e); }
    {
        const char *e = getenv("APRIL_SPLIT_STREAMS");
        split_streams_ = e && *e ? std::max(0, std::min(2, atoi(e))) : 2;
    }
    const NetDims &d = L_.dims;
    w_ = dmalloc<float>(L_.total);
    if (blob_device) HIP_CHECK(hipMemcpy(w_, blob_device, L_.total * 4, hipMemcpyDeviceToDevice));
    else if (blob_host) HIP_CHECK(hipMemcpy(w_, blob_host, L_.total * 4, hipMemcpyHostToDevice));
    // (neither: the caller fills weights_mut() -- peer copy or RCCL broadcast -- and then calls finish_weights())

    const size_t S = (size_t)cfg_.max_slots, MB = (size_t)cfg_.max_batch;
    // Feature ring per session.  The reference keeps segment_size * 32 frames (src/fbank.c:147); a session that is fed faster
    // than real time (a whole file in one call) is processed a ring's worth of chunks at a time, and the offline wavefront
    // (run_lm_wavefront) needs many more blocks of time steps than layers to fill: 8192 frames = ~80 s of audio per pass
    // (a minute of audio in one call is ONE wavefront: 65.7 ms against 68.8 ms in three passes of a 2048-frame ring),
    // 2.6 MB per slot (10.7 GB at 4096 slots, 3.7 % of this GPU's memory).  Results do not depend on the ring size.
    {
        const char *e = getenv("APRIL_RING_FRAMES");
        ring_frames_ = std::max(P_.segment_size * 32, e && *e ? atoi(e) : 8192);
    }
    h_ = dmalloc<float>((size_t)d.n_layers * S * d.d_model);
    c_ = dmalloc<float>((size_t)d.n_layers * S * d.hidden);
    if (!(getenv("APRIL_RING_FRAMES") && *getenv("APRIL_RING_FRAMES"))) {
        // the default ring is sized for this GPU's 288 GB (10.7 GB at 4096 slots); where that is more than a quarter of what the
        // device has free right now -- several models / lanes / ranks on one device, a smaller GPU -- the ring shrinks (more
        // passes per long feed, same results), so that the allocations that FOLLOW it (state, work buffers, fp16 copies,
        // decoder table) still fit: those abort the process when they fail
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
            const size_t per_frame = S * (size_t)d.mel * sizeof(float);
            const size_t fit = (free_b / 4) / per_frame;
            if (fit < (size_t)ring_frames_) {
                const int shrunk = std::max(P_.segment_size * 32, (int)std::min<size_t>(fit, 8192) / 64 * 64);
                LOGW("engine: %zu MB of feature rings (%d frames x %zu slots) exceed a quarter of the %zu MB free on device %d: %d frames per session",
                     per_frame * (size_t)ring_frames_ >> 20, ring_frames_, S, free_b >> 20, cfg_.device, shrunk);
                ring_frames_ = shrunk;
            }
        } else (void)hipGetLastError();
        float *p = nullptr;
        if (hipMalloc((void **)&p, S * ring_frames_ * d.mel * sizeof(float)) == hipSuccess) ring_ = p;
        else {
            (void)hipGetLastError();
            const size_t asked = S * (size_t)ring_frames_ * d.mel * sizeof(float);
            ring_frames_ = std::max(P_.segment_size * 32, 2048);
            LOGW("engine: no room for %zu MB of feature rings, falling back to %d frames per session", asked >> 20, ring_frames_);
        }
    }
    if (!ring_) ring_ = dmalloc<float>(S * ring_frames_ * d.mel);
    eout_ = dmalloc<float>(S * d.joiner);
    dout_ = dmalloc<float>(S * d.joiner);
    gstate_ = dmalloc<GreedyState>(S);
    HIP_CHECK(hipMemset(h_, 0, (size_t)d.n_layers * S * d.d_model * 4));
    HIP_CHECK(hipMemset(c_, 0, (size_t)d.n_layers * S * d.hidden * 4));
    HIP_CHECK(hipMemset(ring_, 0, S * ring_frames_ * d.mel * 4));
    HIP_CHECK(hipMemset(eout_, 0, S * d.joiner * 4));
    HIP_CHECK(hipMemset(dout_, 0, S * d.joiner * 4));
    {   // every slot starts with context [blank, blank] (april_session.c:432-438), no emission yet (:64-66)
        std::vector<GreedyState> init(S);
        for (auto &g : init) { g.ctx0 = P_.blank_id; g.ctx1 = P_.blank_id; g.last_tok = -1; g.last_emit_ms = 0; }
        HIP_CHECK(hipMemcpy(gstate_, init.data(), S * sizeof(GreedyState), hipMemcpyHostToDevice));
    }
    cls_ = dmalloc<uint8_t>((size_t)d.vocab);
    HIP_CHECK(hipMemcpy(cls_, tok_class.data(), (size_t)d.vocab, hipMemcpyHostToDevice));

    kz_embed_ = pick_kz(d.embed_in, d.d_model);
    kz_hr_ = pick_kz(d.hidden, d.d_model);
    kz_ff2_ = pick_kz(d.ffn, d.d_model);
    kz_proj_ = pick_kz(d.d_model, d.joiner);
    kz_out_ = pick_kz(d.joiner, L_.vocab_pad);
    {   // fp16 tile path: all four layer GEMMs must cut into 4 chunks of whole 32-k blocks (and the gates' y half into whole stages)
        const char *e = getenv("APRIL_F16_TILE");
        const bool want = cfg_.precision == 1 && !(e && *e && atoi(e) == 0);
        f16_tile_ = want && d.d_model % 128 == 0 && d.hidden % 128 == 0 && d.ffn % 128 == 0 && d.hidden % 16 == 0;
        {   // weight prefetch from a side stream: MEASUREMENT FORM, off (APRIL_PREFETCH=1).  Measured round 6: configs[4] fp16 1.78 -> 2.5 ms
            // per step, 256 sessions fp32 1.334 -> 2.10 ms -- the two cross-stream edges per launch inside the captured graph cost far
            // more than the ~3 us of HBM latency a launch saves (tools/pp_bench `cold`)
            const char *pe = getenv("APRIL_PREFETCH");
            prefetch_ = pe && *pe && atoi(pe) != 0;
        }
        if (f16_tile_) {
            kzx_hr_ = pick_kz(d.hidden, d.d_model, 32); kzx_ff2_ = pick_kz(d.ffn, d.d_model, 32);
            for (int p = 0; p < 2; ++p) y16_buf_[p] = dmalloc<uint16_t>(MB * d.d_model);
            y16_ = y16_buf_[0]; xb16_ = dmalloc<uint16_t>(MB * d.d_model);
            u16_ = dmalloc<uint16_t>(MB * d.hidden); ff16_ = dmalloc<uint16_t>(MB * d.ffn);
            h16_ = dmalloc<uint16_t>((size_t)d.n_layers * S * d.d_model);
            HIP_CHECK(hipMemset(h16_, 0, (size_t)d.n_layers * S * d.d_model * 2));
        }
    }
    ws_mstride_ = cfg_.max_batch;
    const size_t ws_n = (size_t)std::max({kz_embed_ * d.d_model, kz_hr_ * d.d_model, kz_ff2_ * d.d_model, kz_proj_ * d.joiner, kz_out_ * L_.vocab_pad});
    ws_ = dmalloc<float>(ws_n * MB);
    ws_g_ = dmalloc<float>((size_t)std::max(kz_proj_ * d.joiner, kz_out_ * L_.vocab_pad) * MB);    // the search's own workspace: it runs beside encoder stages
    {   // K-cut hand-over of the projection / FFN-down stream kernels (<= 16 rows): [layer][which][d_model / granule][kz][16 rows][granule columns]
        // floats = d_model * kz * 16 per problem, one counter word per granule (zero between launches: the last workgroup re-arms it).
        // Not in APRIL_CHAIN_STREAMS mode: there the chunks of one layer run beside each other on their own streams.
        const bool chains = getenv("APRIL_CHAIN_STREAMS") && atoi(getenv("APRIL_CHAIN_STREAMS")) != 0;
        const bool ks_on = getenv("APRIL_RECUR_KSPLIT") && atoi(getenv("APRIL_RECUR_KSPLIT")) != 0;      // (a measurement form, off by default: kernels_recur.hip recur_ksplit)
        if (ks_on && !chains && cfg_.precision == 0) {
            ks_ws_stride_ = (size_t)d.d_model * (size_t)std::max(kz_hr_, kz_ff2_) * 16;
            ks_cnt_stride_ = (size_t)d.d_model / 16;
            ks_ws_ = dmalloc<float>(2 * (size_t)d.n_layers * ks_ws_stride_);
            ks_cnt_ = dmalloc<unsigned>(2 * (size_t)d.n_layers * ks_cnt_stride_);
            HipLegacyLock legacy;
            HIP_CHECK(hipMemset(ks_cnt_, 0, 2 * (size_t)d.n_layers * ks_cnt_stride_ * sizeof(unsigned)));
        }
    }
    xin_ = dmalloc<float>(MB * d.embed_in);
    a3_ = dmalloc<float>(MB * d.f_out * L_.k3);
    HIP_CHECK(hipMemset(a3_, 0, MB * d.f_out * L_.k3 * 4));      // padded k columns (if any) stay zero
    // buffers that the front end / index fetch of flight k + 1 writes while flight k's layers or search still read them exist
    // once per flight parity (begin_flight() selects): encoder rows y + their sums of squares, the step's index arrays, the
    // round flags, the record offset, the batched encoder outputs (eout_lm_, allocated on first use)
    for (int p = 0; p < 2; ++p) { y_buf_[p] = dmalloc<float>(MB * d.d_model); ssq_buf_[p] = dmalloc<float>(MB * (d.d_model / SSQ_COLS)); }
    y_ = y_buf_[0]; ssq_ = ssq_buf_[0];
    ws_fe_ = dmalloc<float>((size_t)kz_embed_ * d.d_model * MB);      // the front end's own split-K planes (embed at small batches)
    ws_sr_ = dmalloc<float>((size_t)std::max(kz_proj_ * d.joiner, d.d_model) * MB);      // encoder_proj's, when it runs on the search stream
    xb_ = dmalloc<float>(MB * d.d_model);
    u_ = dmalloc<float>(MB * d.hidden);
    ff_ = dmalloc<float>(MB * d.ffn);
    de_ = dmalloc<float>(MB * d.d_model);
    HIP_CHECK(hipMemset(de_, 0, MB * d.d_model * 4));            // rows whose context did not change are read (never stored) by the decoder projection
    logits_ = dmalloc<float>(3 * MB * d.vocab);
    logits_h_ = hmalloc<float>(3 * MB * d.vocab);
    // step bookkeeping
    // (two flights may be in the air -- the one the GPU runs and the one the host enqueues behind it -- so every per-flight
    // ring exists twice: flight parity p owns [p * cap, (p + 1) * cap) of the index blocks, step tables and records)
    ring_cap_ = std::max<size_t>((size_t)1 << 20, 3 * MB * 8 + 2 * S); ring_h_ = hmalloc<int>(2 * ring_cap_);
    step_cap_ = 1 << 14; step_off_h_ = hmalloc<int>((size_t)2 * step_cap_); rec_off_h_ = hmalloc<int>((size_t)2 * step_cap_);
    rec_cap_ = std::max<size_t>((size_t)1 << 20, 3 * MB * 8); rec_d_ = dmalloc<StepRecord>(2 * rec_cap_); rec_h_ = hmalloc<StepRecord>(2 * rec_cap_);
    for (int i = 0; i < 2; ++i) HIP_CHECK(hipEventCreateWithFlags(&flight_done_[i], hipEventDisableTiming));
    counter_d_ = dmalloc<int>(1);
    for (int p = 0; p < 2; ++p) {
        rec_off_buf_[p] = dmalloc<int>(1); flags_buf_[p] = dmalloc<int>(8); step_buf_[p] = dmalloc<int>(4 * MB);
        HIP_CHECK(hipMemset(rec_off_buf_[p], 0, 4)); HIP_CHECK(hipMemset(flags_buf_[p], 0, 32));
    }
    rec_off_d_ = rec_off_buf_[0]; flags_d_ = flags_buf_[0]; step_d_ = step_buf_[0];
    active_d_ = dmalloc<int>(MB); dirty_d_ = dmalloc<int>(MB);
    dec_slots_d_ = dmalloc<int>(std::max(S, MB));
    HIP_CHECK(hipMemset(counter_d_, 0, 4));

    upload_tables(ft);
    use_graphs_ = !(getenv("APRIL_NO_GRAPHS") && atoi(getenv("APRIL_NO_GRAPHS")));
    free_.reserve(S);
    for (int i = cfg_.max_slots - 1; i >= 0; --i) free_.push_back(i);
    LOGI("engine: device %d, %d slots, max batch %d, weights %.1f MB, kz(embed,hr,ff2,proj,out)=%d,%d,%d,%d,%d",
         cfg_.device, cfg_.max_slots, cfg_.max_batch, L_.total * 4.0 / 1e6, kz_embed_, kz_hr_, kz_ff2_, kz_proj_, kz_out_);
}

void Engine::finish_weights()
{
    HipLegacyLock legacy;
    HIP_CHECK(hipSetDevice(cfg_.device));
    if (cfg_.precision == 1 && !wh_) {
        // fp16 operand mode (BASELINE configs[4]): every Linear / LSTM weight matrix gets an fp16 copy in the same
        // packed element order (round-to-nearest-even, on the device); convolutions, biases, embeddings stay fp32
        wh_ = dmalloc<uint16_t>(L_.total);
        for (const auto &sec : gemm_sections(L_)) launch_cvt_f16(w_ + sec.first, wh_ + sec.first, sec.second, nullptr);
        if (f16_tile_) {
            // the layer GEMMs once more, in the order of the 32-k MFMA's B fragment (the other GEMMs -- embed, encoder_proj,
            // decoder, joiner -- stay on the round-2 kernels and the copies above)
            const NetDims &d = L_.dims;
            wx_ = dmalloc<uint16_t>(L_.total);
            for (const PackedLayout::Layer &o : L_.layers) {
                launch_repack_x32(w_ + o.wg, wx_ + o.wg, 2 * d.d_model, 4 * d.hidden, nullptr);
                launch_repack_x32(w_ + o.whr, wx_ + o.whr, d.hidden, d.d_model, nullptr);
                launch_repack_x32(w_ + o.wff1, wx_ + o.wff1, d.d_model, d.ffn, nullptr);
                launch_repack_x32(w_ + o.wff2, wx_ + o.wff2, d.ffn, d.d_model, nullptr);
            }
        }
        HIP_CHECK(hipDeviceSynchronize());
    }
    if (!conv_wt_) {   // the many-chunk form of the conv front end reads its weights channel-fastest (kernels_misc.hip conv12_wide_kernel)
        const NetDims &d = L_.dims;
        conv_wt_ = dmalloc<float>((size_t)9 * d.conv_ch[0] + (size_t)9 * d.conv_ch[0] * d.conv_ch[1]);
        launch_conv_weight_transpose(w_ + L_.conv_w[0], w_ + L_.conv_w[1], d.conv_ch[0], d.conv_ch[1], conv_wt_, conv_wt_ + (size_t)9 * d.conv_ch[0], nullptr);
        HIP_CHECK(hipDeviceSynchronize());
    }
    build_dec_table();
}

// The decoder network is a pure function of the two context tokens (embedding, grouped conv over the context, ReLU, projection:
// reference src/april_session.c:151-163), so its output for EVERY context is computed once at load -- vocab^2 rows of `joiner`
// floats (500^2 x 512 x 4 B = 512 MB of this GPU's 288 GB) by the same kernels that would run per emitted token -- and the
// joiner reads the row of a session's current context.  A context change then costs nothing: no decoder launches in the chunk
// chain (2..3 per search round), none at session start or after a flush.  Not built for context != 2 or when the table would
// exceed APRIL_DEC_TABLE_MB (default 2048; 0 disables): those models run the decoder per context change as before.
void Engine::build_dec_table()
{
    if (dec_table_) return;
    const NetDims &d = L_.dims;
    const char *e = getenv("APRIL_DEC_TABLE_MB");
    const size_t limit_mb = e && *e ? (size_t)std::max(0, atoi(e)) : 2048;
    const size_t rows = (size_t)d.vocab * (size_t)d.vocab;
    if (d.context != 2 || rows * (size_t)d.joiner * 4 > limit_mb * 1024 * 1024 || rows * (size_t)d.joiner * 4 >= ((size_t)1 << 32)) return;   // (32-bit byte offsets in the GEMM's A addressing)
    const int MB = cfg_.max_batch;
    float *table = dmalloc<float>(rows * (size_t)d.joiner);
    std::vector<int> iota((size_t)MB), ctx((size_t)MB * 2);
    for (int i = 0; i < MB; ++i) iota[(size_t)i] = i;
    int *ctx_d = dmalloc<int>((size_t)MB * 2);
    HIP_CHECK(hipMemcpy(dec_slots_d_, iota.data(), (size_t)MB * 4, hipMemcpyHostToDevice));
    for (size_t base = 0; base < rows; base += (size_t)MB) {
        const int n = (int)std::min<size_t>((size_t)MB, rows - base);
        for (int i = 0; i < n; ++i) { const size_t r = base + (size_t)i; ctx[(size_t)2 * i] = (int)(r / (size_t)d.vocab); ctx[(size_t)2 * i + 1] = (int)(r % (size_t)d.vocab); }
        HIP_CHECK(hipMemcpy(ctx_d, ctx.data(), (size_t)n * 8, hipMemcpyHostToDevice));
        DecEmbedArgs a; a.dec = dec_params(); a.ctx = ctx_d; a.M = n; a.out = de_; a.ldo = d.d_model;
        launch_dec_embed(a, stream_);
        run_decproj(n, dec_slots_d_, nullptr, nullptr, 1, table + base * (size_t)d.joiner);
        HIP_CHECK(hipStreamSynchronize(stream_));
    }
    (void)hipFree(ctx_d);
    dec_table_ = table;
    LOGI("engine: decoder table %zu rows x %d (%.0f MB)", rows, d.joiner, rows * (double)d.joiner * 4 / 1e6);
}

Engine::~Engine()
{
    HipLegacyLock legacy;
    (void)hipSetDevice(cfg_.device);
    (void)hipStreamSynchronize(f_stream_); (void)hipStreamSynchronize(stream_); (void)hipStreamSynchronize(s_stream_);
    dump_stream_trace();
    for (StreamTrace &t : trace_) for (hipEvent_t e : t.ev) if (e) (void)hipEventDestroy(e);
    if (trace_base_) (void)hipEventDestroy(trace_base_);
    for (hipStream_t cs : chain_streams_) (void)hipStreamDestroy(cs);
    for (hipEvent_t e : chain_ev_) (void)hipEventDestroy(e);
    for (hipEvent_t e : join_ev_) (void)hipEventDestroy(e);
    for (auto &e : ev_pool_) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &g : step_graphs_) (void)hipGraphExecDestroy(g.second);
    for (auto &g : lm_graphs_) (void)hipGraphExecDestroy(g.second);
    for (auto &g : lm_search_graphs_) (void)hipGraphExecDestroy(g.second);
    for (auto &p : sw_plans_) { if (p.second.graph) (void)hipGraphExecDestroy(p.second.graph); for (hipGraphExec_t x : p.second.g3) if (x) (void)hipGraphExecDestroy(x); if (p.second.dev) (void)hipFree(p.second.dev); if (p.second.rdev) (void)hipFree(p.second.rdev); if (p.second.pf_dev) (void)hipFree(p.second.pf_dev); }
    if (lm_stream_) { (void)hipStreamSynchronize(lm_stream_); (void)hipStreamDestroy(lm_stream_); }
    for (hipEvent_t e : lm_events_) (void)hipEventDestroy(e);
    if (zargs_h_) { (void)hipHostFree(zargs_h_); (void)hipFree(zargs_d_); }
    for (int i = 0; i < 3; ++i) if (zargs_done_[i]) (void)hipEventDestroy(zargs_done_[i]);
    for (void *p : {(void *)lm_now_d_, (void *)lm_rows_d_, (void *)lm_rec_off_d_}) if (p) (void)hipFree(p);
    if (ws_g_) (void)hipFree(ws_g_);
    if (ks_ws_) (void)hipFree(ks_ws_);
    if (conv_wt_) (void)hipFree(conv_wt_);
    if (ks_cnt_) (void)hipFree(ks_cnt_);
    if (dec_table_) (void)hipFree(dec_table_);
    if (p_lm_) (void)hipFree(p_lm_);
    for (int p = 0; p < 2; ++p)
        for (void *q : {(void *)eout_lm_buf_[p], (void *)y16_buf_[p], (void *)y_buf_[p], (void *)ssq_buf_[p], (void *)rec_off_buf_[p], (void *)flags_buf_[p], (void *)step_buf_[p]}) if (q) (void)hipFree(q);
    if (ws_fe_) (void)hipFree(ws_fe_);
    if (ws_sr_) (void)hipFree(ws_sr_);
    for (void *p : {(void *)wx_, (void *)xb16_, (void *)u16_, (void *)ff16_, (void *)h16_}) if (p) (void)hipFree(p);      // fp16 tile path
    for (void *p : {(void *)w_, (void *)wh_, (void *)h_, (void *)c_, (void *)ring_, (void *)eout_, (void *)dout_, (void *)gstate_, (void *)cls_, (void *)ws_, (void *)xin_,
                    (void *)a3_, (void *)xb_, (void *)u_, (void *)ff_, (void *)de_, (void *)logits_, (void *)rec_d_, (void *)counter_d_,
                    (void *)active_d_, (void *)dirty_d_, (void *)dec_slots_d_,
                    (void *)ds_desc_[0], (void *)ds_desc_[1], (void *)ds_pcm_[0], (void *)ds_pcm_[1]})
        if (p) (void)hipFree(p);
    for (void *p : {(void *)ring_h_, (void *)step_off_h_, (void *)rec_off_h_, (void *)rec_h_, (void *)logits_h_, (void *)hs_desc_[0], (void *)hs_desc_[1], (void *)hs_pcm_[0], (void *)hs_pcm_[1]})
        if (p) (void)hipHostFree(p);
    for (int b = 0; b < 2; ++b) if (fb_done_[b]) (void)hipEventDestroy(fb_done_[b]);
    for (int b = 0; b < 2; ++b) if (flight_done_[b]) (void)hipEventDestroy(flight_done_[b]);
    for (void *p : table_allocs_) (void)hipFree(p);
    (void)hipStreamDestroy(stream_); (void)hipStreamDestroy(f_stream_); (void)hipStreamDestroy(s_stream_); if (pf_stream_) (void)hipStreamDestroy(pf_stream_); for (hipEvent_t e : pf_ev_) if (e) (void)hipEventDestroy(e);
}

void Engine::upload_tables(const FbankHostTables &ft)
{
    auto up = [&](const void *src, size_t bytes) {
        void *p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        if (bytes) HIP_CHECK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        table_allocs_.push_back(p);
        return p;
    };
    ft_.window = (const float *)up(ft.window.data(), ft.window.size() * 4);
    ft_.mel = (const float *)up(ft.mel.data(), ft.mel.size() * 4);
    ft_.mel_lo = (const int *)up(ft.mel_lo.data(), ft.mel_lo.size() * 4);
    ft_.mel_hi = (const int *)up(ft.mel_hi.data(), ft.mel_hi.size() * 4);
    ft_.nfct = (int)ft.factors.size();
    for (int k = 0; k < ft_.nfct; ++k) {
        ft_.fct[k] = ft.factors[(size_t)k];
        ft_.tw[k] = (const double *)up(ft.tw[(size_t)k].data(), ft.tw[(size_t)k].size() * 8);
        ft_.tws[k] = ft.tws[(size_t)k].empty() ? nullptr : (const double *)up(ft.tws[(size_t)k].data(), ft.tws[(size_t)k].size() * 8);
    }
    ft_.padded = ft.padded; ft_.nbins = ft.nbins;
    pad_value_ = ft.pad_value;
}

int Engine::alloc_slot()
{
    std::lock_guard<std::mutex> g(slot_mu_);
    if (free_.empty() && !zero_pending_.empty()) zero_pending_slots();
    if (free_.empty()) return -1;
    const int s = free_.back();
    free_.pop_back();
    ++live_;
    return s;
}

// (slot_mu_ held) reset every slot that was freed since the last call: up to 64 slots per launch instead of one launch per
// aas_free -- tearing down 2048 sessions cost 2048 launches (4.3 ms of the stream, profiles/r03_b2048_kernel_stats.csv)
void Engine::zero_pending_slots()
{
    if (zero_pending_.empty()) return;
    // Teardown-tolerant: sessions may be freed while the process is exiting and the HIP runtime is already gone.
    if (hipSetDevice(cfg_.device) == hipSuccess) {
        std::lock_guard<std::mutex> cg(capture_mu_);           // never enqueue into a stream that is being captured (step())
        const NetDims &d = L_.dims;
        for (size_t o = 0; o < zero_pending_.size(); o += 64) {
            ZeroSlotArgs z;
            z.h = h_; z.c = c_; z.n_layers = d.n_layers; z.slots = (size_t)cfg_.max_slots; z.d_model = d.d_model; z.hidden = d.hidden;
            z.eout = eout_; z.dout = dout_; z.joiner = d.joiner; z.state = gstate_; z.blank = P_.blank_id; z.h16 = h16_;
            z.n_list = (int)std::min<size_t>(64, zero_pending_.size() - o);
            for (int i = 0; i < z.n_list; ++i) z.list[i] = zero_pending_[o + (size_t)i];
            launch_zero_slot(z, stream_);
        }
    }
    for (int s : zero_pending_) free_.push_back(s);
    zero_pending_.clear();
}

void Engine::free_slot(int slot)
{
    // the slot goes back to the free list once it has been reset to the reference's calloc'd tensors (april_session.c:40-58)
    // and a [blank, blank] context -- in a batch: when 256 freed slots have gathered, when the free list runs dry, or before
    // the next flight is launched (begin_flight), stream-ordered ahead of any later use of the slot
    std::lock_guard<std::mutex> g(slot_mu_);
    zero_pending_.push_back(slot);
    --live_;
    if (zero_pending_.size() >= 256) zero_pending_slots();
}

// Waits for the three streams and touches nothing else: safe from any thread (the readers below are called by client threads
// while the stepping thread keeps enqueueing).
void Engine::sync_streams()
{
    HIP_CHECK(hipStreamSynchronize(f_stream_)); HIP_CHECK(hipStreamSynchronize(stream_)); HIP_CHECK(hipStreamSynchronize(s_stream_));
}

// Stepping thread only (or under capture_mu_): everything enqueued so far is done, so no stream has work another one has not seen.
// A client thread must NOT clear these flags -- between its stream waits and the store the stepping thread may have enqueued a new
// fbank and set one; it uses sync_streams(), a stale "unseen" only costs a redundant event.
void Engine::sync()
{
    sync_streams();
    f_unseen_by_m_ = s_unseen_by_m_ = m_unseen_by_f_ = m_unseen_by_s_ = false;
    if (profiling_) collect_timing();
}

// Graph captures use hipStreamCaptureModeRelaxed AND hold hip_legacy_mutex() (engine.h): a capture begun in the thread-local (or
// global) mode makes every "potentially unsafe" HIP call -- hipMalloc, hipMemcpy, hipFree ... -- of every OTHER thread that is in the
// default global mode fail with hipErrorStreamCaptureUnsupported for as long as the capture lasts; in relaxed mode a legacy-stream copy
// of another thread still fails ("would make the legacy stream depend on a capturing blocking stream", although the streams here are
// non-blocking) and invalidates the capture.  Client threads make exactly such calls: another model being loaded,
// aprilx_session_read_frames / aprilx_session_context on an idle session while other sessions stream (found in round 5 by running
// tests/sched_harness/driver.cc against the real engine: the reader's hipMemcpy aborted the process).  The stepping thread itself
// issues nothing but kernel launches, async copies and event records inside a capture.
// ---------------------------------------------------------------- streams
// Three in-order streams.  stream_ (M) carries the layer chain and every launch of the general paths; f_stream_ (F) the PCM
// upload and the fbank kernel of every flight and, for a feed that runs as a split wavefront (lm_step mode 1), its index fetch
// and conv / embed front end; s_stream_ (S) the search of such a feed.  Inside one flight FE -> layers -> search are chained by
// events; ACROSS flights the front end of flight k + 1 and the search of flight k run beside the layers of flights k / k + 1
// (they share no buffer: see the per-parity buffers in the constructor).  The general paths (chunk-by-chunk steps, long feeds,
// decoder refreshes, traced steps) stay on M and first wait for whatever F and S still have in the air; a split feed after
// general work makes F and S wait for M the same way.  The flags say which stream has work the other has not waited for yet.
// APRIL_STREAM_TRACE=<file> (measurement): hipEvent time stamps around the three parts of every split feed -- front end on F,
// layers on M, search on S -- written as one line per feed when the engine is destroyed: the evidence that neighbouring feeds'
// parts overlap (rocprofv3's kernel trace serialises the queues, so it cannot show it).
Engine::StreamTrace *Engine::trace_slot()
{
    static const char *path = getenv("APRIL_STREAM_TRACE");
    if (!path || !*path) return nullptr;
    if (!trace_base_) { HIP_CHECK(hipEventCreate(&trace_base_)); HIP_CHECK(hipEventRecord(trace_base_, stream_)); }
    if (trace_.size() >= 4096) return nullptr;
    trace_.emplace_back();
    StreamTrace &t = trace_.back();
    for (hipEvent_t &e : t.ev) HIP_CHECK(hipEventCreate(&e));
    return &t;
}

void Engine::dump_stream_trace()
{
    const char *path = getenv("APRIL_STREAM_TRACE");
    if (!path || !*path || trace_.empty()) return;
    FILE *f = fopen(path, "w");
    if (!f) return;
    fprintf(f, "# one line per split feed; times in us since the first split feed; FE = index fetch + conv + embed on stream F, LY = layer wavefront on M, SR = encoder_proj + search on S\n");
    fprintf(f, "# feed sessions chunks  FE_start FE_end  LY_start LY_end  SR_start SR_end   overlap: FE inside the previous feed's LY window? SR inside the next feed's LY window?\n");
    std::vector<std::array<float, 6>> ts;
    for (StreamTrace &t : trace_) {
        std::array<float, 6> a{};
        bool ok = t.used;
        for (int i = 0; i < 6 && ok; ++i) ok = hipEventElapsedTime(&a[(size_t)i], trace_base_, t.ev[i]) == hipSuccess;
        if (ok) ts.push_back(a); else (void)hipGetLastError();
    }
    for (size_t i = 0; i < ts.size(); ++i) {
        const auto &a = ts[i];
        const bool fe_in_prev = i > 0 && a[0] * 1e3f < ts[i - 1][3] * 1e3f && a[1] * 1e3f > ts[i - 1][2] * 1e3f;
        const bool sr_in_next = i + 1 < ts.size() && a[5] > ts[i + 1][2] && a[4] < ts[i + 1][3];
        fprintf(f, "%4zu %5d %2d  %10.1f %10.1f  %10.1f %10.1f  %10.1f %10.1f   %s %s\n", i, trace_[i].m, trace_[i].T, a[0] * 1e3, a[1] * 1e3, a[2] * 1e3, a[3] * 1e3, a[4] * 1e3, a[5] * 1e3,
                fe_in_prev ? "FE<prevLY" : "-", sr_in_next ? "SR<nextLY" : "-");
    }
    fclose(f);
}

void Engine::join(hipStream_t waiter, hipStream_t src)
{
    hipEvent_t e = join_ev_[join_pos_++ % join_ev_.size()];
    HIP_CHECK(hipEventRecord(e, src));
    HIP_CHECK(hipStreamWaitEvent(waiter, e, 0));
}

void Engine::general_prologue()
{
    if (f_unseen_by_m_) { join(stream_, f_stream_); f_unseen_by_m_ = false; }
    if (s_unseen_by_m_) { join(stream_, s_stream_); s_unseen_by_m_ = false; }
    m_unseen_by_f_ = m_unseen_by_s_ = true;
    flight_tail_s_ = false;
}

// ---------------------------------------------------------------- profiling
void Engine::set_profiling(bool on) { std::lock_guard<std::mutex> cg(capture_mu_); sync(); profiling_ = on; }
void Engine::reset_timing() { for (auto &t : timing_) t = KernelTiming(); }
void Engine::set_gates_clock(bool on)
{
    std::lock_guard<std::mutex> cg(capture_mu_);
    HIP_CHECK(hipSetDevice(cfg_.device));
    sync();
    if (on) {
        if (!gclk_slots_) gclk_slots_ = dmalloc<unsigned long long>((size_t)GCLK_SLOTS * STAMP_WORDS);
        HIP_CHECK(hipMemsetAsync(gclk_slots_, 0, (size_t)GCLK_SLOTS * STAMP_WORDS * 8, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        gclk_ms_ = 0; gclk_launches_ = 0; gclk_rows_ = 0;
        for (int i = 0; i < 4; ++i) { gclk_ms_n_[i] = 0; gclk_launches_n_[i] = 0; }
        gclk_ = true;
        return;
    }
    if (gclk_ && gclk_slots_) {
        // every plan built while the clock was on: its slots' sums (10 ns ticks of s_memrealtime) and launch counts
        std::vector<unsigned long long> h((size_t)GCLK_SLOTS * STAMP_WORDS);
        HIP_CHECK(hipMemcpyAsync(h.data(), gclk_slots_, h.size() * 8, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (auto &kv : sw_plans_)
            for (size_t i = 0; i < kv.second.stamp_slots.size(); ++i) {
                const auto &sl = kv.second.stamp_slots[i];
                const unsigned long long ticks = h[(size_t)sl.first * STAMP_WORDS + 2], n = h[(size_t)sl.first * STAMP_WORDS + 3];
                gclk_ms_ += (double)ticks * 1e-5; gclk_launches_ += (long)n; gclk_rows_ += (long)n * sl.second;
                const int bn = std::min(std::max(kv.second.stamp_n[i], 1), 4) - 1;
                gclk_ms_n_[bn] += (double)ticks * 1e-5; gclk_launches_n_[bn] += (long)n;
            }
    }
    gclk_ = false;
}
void Engine::timed_begin(int cls)
{
    ++launch_count_;
    if (!profiling_) return;
    if (ev_used_ == ev_pool_.size()) { Ev e; HIP_CHECK(hipEventCreate(&e.a)); HIP_CHECK(hipEventCreate(&e.b)); e.cls = cls; ev_pool_.push_back(e); }
    ev_pool_[ev_used_].cls = cls;
    // gates launches: the kernel's own dispatch time stamps (hipExtLaunchKernel through APRIL_LAUNCH) instead of event packets around
    // it -- the clock of rocprofv3's per-kernel duration, which bench.py's roofline is priced on; other classes: brackets on the stream
    if (cls == T_GATES) gemm_profile_next_launch(ev_pool_[ev_used_].a, ev_pool_[ev_used_].b);
    else HIP_CHECK(hipEventRecord(ev_pool_[ev_used_].a, stream_));
}
void Engine::timed_end(int cls)
{
    if (!profiling_) return;
    if (cls == T_GATES) {
        // (a launch path that does not go through APRIL_LAUNCH left the pair untouched: drop the sample instead of reading unrecorded events)
        if (gemm_profile_pending()) { gemm_profile_next_launch(nullptr, nullptr); return; }
    } else HIP_CHECK(hipEventRecord(ev_pool_[ev_used_].b, stream_));
    ++ev_used_;
}
void Engine::collect_timing()
{
    for (size_t i = 0; i < ev_used_; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev_pool_[i].a, ev_pool_[i].b) == hipSuccess) { timing_[ev_pool_[i].cls].ms += ms; timing_[ev_pool_[i].cls].launches++; }
    }
    ev_used_ = 0;
}

// ---------------------------------------------------------------- fbank
void Engine::fbank(int n_frames, const FbankFrameDesc *desc, const std::pair<const int16_t *, size_t> *parts, size_t n_parts, size_t n_pcm, HostPool *pool)
{
    if (n_frames <= 0) return;
    HIP_CHECK(hipSetDevice(cfg_.device));
    if (n_frames > desc_cap_ || n_pcm > pcm_cap_) {
        sync();
        HipLegacyLock regrow_guard;                   // (frees imply a device synchronisation: not beside another engine's capture; order: capture_mu_, then this)
        for (int b = 0; b < 2; ++b) {
            if (hs_desc_[b]) { (void)hipHostFree(hs_desc_[b]); (void)hipFree(ds_desc_[b]); }
            if (hs_pcm_[b]) { (void)hipHostFree(hs_pcm_[b]); (void)hipFree(ds_pcm_[b]); }
        }
        desc_cap_ = std::max({n_frames * 2, desc_cap_, 1024});
        pcm_cap_ = std::max({n_pcm * 2, pcm_cap_, (size_t)1 << 16});
        // one staging buffer per flip: the PCM windows, then (16-byte aligned, right behind the samples of THIS call) the frame
        // descriptors -> one host-to-device copy per call instead of two
        const size_t units = pcm_cap_ + 8 + ((size_t)desc_cap_ * sizeof(FbankFrameDesc) + 1) / 2;
        for (int b = 0; b < 2; ++b) {
            hs_pcm_[b] = hmalloc<int16_t>(units); ds_pcm_[b] = dmalloc<int16_t>(units);
            hs_desc_[b] = nullptr; ds_desc_[b] = nullptr;
            if (!fb_done_[b]) HIP_CHECK(hipEventCreateWithFlags(&fb_done_[b], hipEventDisableTiming));
        }
    }
    const int b = fb_flip_;
    fb_flip_ ^= 1;
    HIP_CHECK(hipEventSynchronize(fb_done_[b]));          // the launch that used this pair two calls ago has consumed it
    const size_t doff = (n_pcm * sizeof(int16_t) + 15) / 16 * 16;          // byte offset of the descriptors
    memcpy(reinterpret_cast<char *>(hs_pcm_[b]) + doff, desc, (size_t)n_frames * sizeof(FbankFrameDesc));
    if (pool && n_parts >= 256) {
        part_off_.resize(n_parts);
        size_t off = 0;
        for (size_t i = 0; i < n_parts; ++i) { part_off_[i] = off; off += parts[i].second; }
        int16_t *dst = hs_pcm_[b];
        pool->run(n_parts, 64, [&](size_t i) { memcpy(dst + part_off_[i], parts[i].first, parts[i].second * sizeof(int16_t)); });
    } else {
        size_t off = 0;
        for (size_t i = 0; i < n_parts; ++i) { memcpy(hs_pcm_[b] + off, parts[i].first, parts[i].second * sizeof(int16_t)); off += parts[i].second; }
    }
    std::lock_guard<std::mutex> cg(capture_mu_);
    if (m_unseen_by_f_) { join(f_stream_, stream_); m_unseen_by_f_ = false; }      // (general-path work may still read ring rows this call overwrites)
    HIP_CHECK(hipMemcpyAsync(ds_pcm_[b], hs_pcm_[b], doff + (size_t)n_frames * sizeof(FbankFrameDesc), hipMemcpyHostToDevice, f_stream_));
    FbankArgs a;
    a.t = ft_; a.pcm = ds_pcm_[b]; a.desc = reinterpret_cast<const FbankFrameDesc *>(reinterpret_cast<const char *>(ds_pcm_[b]) + doff); a.n_frames = n_frames; a.ring = ring_; a.ring_frames = ring_frames_; a.pad_value = pad_value_;
    if (profiling_) {        // (the per-class hipEvents live on M: a profiled fbank runs there, behind its upload)
        join(stream_, f_stream_);
        timed_begin(T_FBANK); launch_fbank(a, stream_); timed_end(T_FBANK);
        join(f_stream_, stream_);
    } else launch_fbank(a, f_stream_);
    HIP_CHECK(hipEventRecord(fb_done_[b], f_stream_));       // no host wait here: the encoder launches queue right behind
    f_unseen_by_m_ = true;
}

// ---------------------------------------------------------------- encoder
// Launch count per chunk at batch sizes where the full-K schedule applies: 2 (conv) + 1 (embed) + 4 per layer + 1.
// BasicNorm never runs as a kernel: EPI_RESID_SSQ leaves y and its per-32-column sums of squares, and every consumer
// of the normalised row (the next layer's gate GEMM, its residual, encoder_proj) folds the row scale in (kernels.h).
void Engine::run_encoder_rows(int n, const int *d_slots, const int *d_tails, const float *x_direct)
{
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    const int G = d.d_model / SSQ_COLS;
    auto scale_of = [&](float eps) { RowScale r; r.ssq = ssq_; r.groups = G; r.inv_n = 1.0f / (float)(d.d_norm ? d.d_norm : d.d_model); r.eps = eps; return r; };
    // conv front end
    ConvEmbedArgs ca;
    ca.ring = ring_; ca.ring_frames = ring_frames_; ca.mel = d.mel; ca.seg = d.seg;
    ca.slot_idx = d_slots; ca.ring_tail = d_tails; ca.x_direct = x_direct;
    for (int i = 0; i < 3; ++i) { ca.w[i] = w_ + L_.conv_w[i]; ca.b[i] = w_ + L_.conv_b[i]; ca.ch[i] = d.conv_ch[i]; ca.stride[i] = d.conv_stride[i]; }
    ca.ch1_per_group = 1;                                 // largest divisor of the second conv's channel count that is <= 8
    for (int k = 8; k > 1; --k) if (d.conv_ch[1] % k == 0) { ca.ch1_per_group = k; break; }
    ca.out = a3_; ca.ldo = L_.k3; ca.M = n;
    if (conv_wt_) { ca.w0t = conv_wt_; ca.w1t = conv_wt_ + (size_t)9 * d.conv_ch[0]; }
    timed_begin(T_CONV); launch_conv_embed(ca, stream_); timed_end(T_CONV);
    {   // third conv: [n*f_out, k3] x [k3, c2] + bias, DoubleSwish -> xin[n][f_out*c2]
        GemmArgs g; g.a0 = a3_; g.lda0 = L_.k3; g.K0 = L_.k3; g.wp = w_ + L_.conv_w[2];
        g.M = n * d.f_out; g.N = d.conv_ch[2]; g.K = L_.k3; g.kz = 1; g.epi = EPI_BIAS_DSWISH; g.out = xin_; g.ldo = d.conv_ch[2]; g.bias = w_ + L_.conv_b[2];
        timed_begin(T_CONV); launch_gemm(g, stream_); timed_end(T_CONV);
    }
    // y = A x W + bias (+ residual) with sums of squares: fused into the GEMM where its tiles own all of K, else split-K + row kernel
    auto resid_ssq = [&](const float *a, int K, size_t w_off, int kz, const float *bias, const float *resid, int ks_layer = -1) {
        GemmArgs g; g.a0 = a; g.lda0 = K; g.K0 = K; lin(g, w_off);
        g.M = n; g.N = d.d_model; g.K = K; g.kz = kz; g.tile_ok = tile_ok();
        if (gemm_fullk(n, d.d_model, kz, false, 1, tile_ok())) {
            g.epi = EPI_RESID_SSQ; g.bias = bias; g.resid = resid; g.ldr = d.d_model; g.out = y_; g.ldo = d.d_model; g.ssq_out = ssq_;
            if (ks_layer >= 0) attach_ksplit(g, ks_layer, 1);
            timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
            return;
        }
        g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
        timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
        RowArgs r; r.mode = ROW_RESID_SSQ; r.ws = ws_; r.parts = gemm_partials(n, d.d_model, kz, 1, tile_ok()); r.m_stride = ws_mstride_; r.N = d.d_model; r.M = n;
        r.bias = bias; r.resid = resid; r.ldr = d.d_model; r.out = y_; r.ldo = d.d_model; r.ssq_out = ssq_;
        timed_begin(T_ROW); launch_row(r, stream_); timed_end(T_ROW);
    };
    // embed linear + bias (BasicNorm deferred)
    resid_ssq(xin_, d.embed_in, L_.w_embed, kz_embed_, w_ + L_.b_embed, nullptr);
    float eps_in = L_.embed_eps;                          // epsilon of the BasicNorm that produced this layer's input
    if (f16_tile_) {
        // fp16 tile path: the four GEMMs of every layer read binary16 activations (row block 0 of the work buffers, d_slots ==
        // step_d_) -- the same argument blocks as the feed wavefront at chunk 0; embed and encoder_proj stay on the fp32-A kernels
        launch_cvt_f16(y_, y16_, (size_t)n * d.d_model, stream_);
        for (int l = 0; l < d.n_layers; ++l) {
            timed_begin(T_GATES); launch_gemm(sw_args_gates(l, n, 0), stream_); timed_end(T_GATES);
            launch_rowepi(lm_args_whr(l, n, 0), 0, stream_);
            timed_begin(T_GEMM_OTHER); launch_gemm(lm_args_ff1(l, n, 0, 1), stream_); timed_end(T_GEMM_OTHER);
            launch_rowepi(lm_args_ff2(l, n, 0, 1), 0, stream_);
        }
        eps_in = L_.norm_eps[(size_t)d.n_layers - 1];
    }
    for (int l = 0; l < (f16_tile_ ? 0 : d.n_layers); ++l) {
        const PackedLayout::Layer &o = L_.layers[(size_t)l];
        float *h_l = h_ + (size_t)l * S * d.d_model;
        float *c_l = c_ + (size_t)l * S * d.hidden;
        const RowScale xs = scale_of(eps_in);
        {   // gates = [norm(y) | h_prev] x Wg ; fused LSTM cell
            GemmArgs g; g.a0 = y_; g.lda0 = d.d_model; g.K0 = d.d_model; g.x_scale = xs;
            g.a1 = h_l; g.lda1 = d.d_model; g.aidx1 = d_slots; g.K1 = d.d_model;
            lin(g, o.wg); g.M = n; g.N = 4 * d.hidden; g.K = 2 * d.d_model; g.kz = 1; g.epi = EPI_LSTM;
            g.out = u_; g.ldo = d.hidden; g.bias = w_ + o.bg; g.c_state = c_l; g.slot_idx = d_slots; g.hidden = d.hidden;
            if (cfg_.precision == 0 && gates_tile_rows(n)) g.tile_ok = 2;
            timed_begin(T_GATES); launch_gemm(g, stream_); timed_end(T_GATES);
        }
        {   // h' = u x Whr ; state write + residual: xb = norm(y) + h'
            GemmArgs g; g.a0 = u_; g.lda0 = d.hidden; g.K0 = d.hidden; lin(g, o.whr);
            g.M = n; g.N = d.d_model; g.K = d.hidden; g.kz = kz_hr_; g.tile_ok = tile_ok();
            if (gemm_fullk(n, d.d_model, kz_hr_, false, 1, tile_ok())) {
                g.epi = EPI_HR; g.state = h_l; g.ld_state = d.d_model; g.slot_idx = d_slots; g.resid = y_; g.ldr = d.d_model; g.r_scale = xs; g.out = xb_; g.ldo = d.d_model;
                attach_ksplit(g, l, 0);
                timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
            } else {
                g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
                timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
                RowArgs r; r.mode = ROW_HR; r.ws = ws_; r.parts = gemm_partials(n, d.d_model, kz_hr_, 1, tile_ok()); r.m_stride = ws_mstride_; r.N = d.d_model; r.M = n;
                r.resid = y_; r.ldr = d.d_model; r.r_scale = xs; r.out = xb_; r.ldo = d.d_model; r.slot_idx = d_slots; r.state = h_l; r.ld_state = d.d_model;
                timed_begin(T_ROW); launch_row(r, stream_); timed_end(T_ROW);
            }
        }
        {   // FFN up + DoubleSwish
            GemmArgs g; g.a0 = xb_; g.lda0 = d.d_model; g.K0 = d.d_model; lin(g, o.wff1);
            g.M = n; g.N = d.ffn; g.K = d.d_model; g.kz = 1; g.epi = EPI_BIAS_DSWISH; g.out = ff_; g.ldo = d.ffn; g.bias = w_ + o.bff1;
            timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
        }
        // FFN down + bias + residual -> y (the last reader of the previous y was the projection above)
        resid_ssq(ff_, d.ffn, o.wff2, kz_ff2_, w_ + o.bff2, xb_, l);
        eps_in = L_.norm_eps[(size_t)l];
    }
    {   // encoder_proj(norm(y)) -> eout[slot]
        const RowScale ys = scale_of(eps_in);
        GemmArgs g; g.a0 = y_; g.lda0 = d.d_model; g.K0 = d.d_model; lin(g, L_.w_encproj);
        g.M = n; g.N = d.joiner; g.K = d.d_model; g.kz = kz_proj_; g.tile_ok = tile_ok();
        if (gemm_fullk(n, d.joiner, kz_proj_, false, 1, tile_ok())) {
            g.epi = EPI_SLOT_STORE; g.bias = w_ + L_.b_encproj; g.out = eout_; g.ldo = d.joiner; g.slot_idx = d_slots; g.x_scale = ys;
            timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
        } else {
            g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
            timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
            RowArgs r; r.mode = ROW_SLOT_STORE; r.ws = ws_; r.parts = gemm_partials(n, d.joiner, kz_proj_, 1, tile_ok()); r.m_stride = ws_mstride_; r.N = d.joiner; r.M = n;
            r.bias = w_ + L_.b_encproj; r.out = eout_; r.ldo = d.joiner; r.slot_idx = d_slots; r.r_scale = ys;
            timed_begin(T_ROW); launch_row(r, stream_); timed_end(T_ROW);
        }
    }
}

// ---------------------------------------------------------------- decoder / joiner rounds
DecEmbedParams Engine::dec_params() const
{
    const NetDims &d = L_.dims;
    DecEmbedParams p; p.emb = w_ + L_.emb; p.conv_w = w_ + L_.dec_conv; p.conv_b = L_.has_dec_conv_b ? w_ + L_.dec_conv_b : nullptr;
    p.d = d.d_model; p.groups = d.dec_groups; p.context = d.context; p.vocab = d.vocab;
    return p;
}

// dout[slot] = de x Wp + b for rows with row_mask != 0 (all rows when null)
void Engine::run_decproj(int n, const int *d_slots, const int *row_mask, const int *run_flag, int run_gen, float *out)
{
    const NetDims &d = L_.dims;
    if (!out) out = dout_;
    GemmArgs g; g.a0 = de_; g.lda0 = d.d_model; g.K0 = d.d_model; lin(g, L_.w_decproj);
    g.M = n; g.N = d.joiner; g.K = d.d_model; g.kz = kz_proj_; g.tile_ok = tile_ok(); g.run_flag = run_flag; g.run_gen = run_gen;
    if (gemm_fullk(n, d.joiner, kz_proj_, false, 1, tile_ok())) {
        g.epi = EPI_SLOT_STORE; g.bias = w_ + L_.b_decproj; g.out = out; g.ldo = d.joiner; g.slot_idx = d_slots; g.row_mask = row_mask;
        timed_begin(T_DEC); launch_gemm(g, search_stream_); timed_end(T_DEC);
        return;
    }
    g.epi = EPI_PARTIAL; g.out = ws_g_; g.m_stride = ws_mstride_;
    timed_begin(T_DEC); launch_gemm(g, search_stream_); timed_end(T_DEC);
    RowArgs r; r.mode = ROW_SLOT_STORE; r.ws = ws_g_; r.parts = gemm_partials(n, d.joiner, kz_proj_, 1, tile_ok()); r.m_stride = ws_mstride_; r.N = d.joiner; r.M = n;
    r.bias = w_ + L_.b_decproj; r.out = out; r.ldo = d.joiner; r.slot_idx = d_slots; r.row_mask = row_mask; r.run_flag = run_flag; r.run_gen = run_gen;
    timed_begin(T_DEC); launch_row(r, search_stream_); timed_end(T_DEC);
}

// The reference's loop "joiner -> process_logits, up to three times, early-emit 1,0,0" (src/april_session.c:449-454)
// for all rows of the step at once, decisions included: rows that resolved to blank are masked out of the later rounds.
void Engine::run_greedy_rounds(int n, bool dump_logits, int chunk, const float *eout_rows)
{
    const int MB = cfg_.max_batch;
    GreedyIo io;
    io.gen = chunk + 1; io.now = step_d_ + 2 * MB + (size_t)chunk * n; io.eout = eout_rows; io.rec_off = rec_off_d_; io.rec_slot0 = chunk * 3;
    io.dump = dump_logits ? logits_ + (size_t)chunk * 3 * n * L_.dims.vocab : nullptr;
    run_greedy_rounds(n, io);
}

void Engine::run_greedy_rounds(int n, const GreedyIo &io)
{
    const NetDims &d = L_.dims;
    const int gen = io.gen;
    const int *d_slots = step_d_;
    for (int round = 0; round < 3; ++round) {
        {   // logits = tanh(eout + dout) x Wout (+ bias in the decision kernel)
            GemmArgs g; g.a0b = dout_; g.lda0 = d.joiner; g.K0 = d.joiner; g.a_op = AOP_TANH_ADD;
            if (dec_table_) { g.a0b = dec_table_; g.ctx_state = gstate_; g.ctx_vocab = d.vocab; }     // dout = the table row of the slot's context
            if (io.eout) { g.a0 = io.eout; g.aidx0 = io.eout_rows; g.same_idx_b = 0; g.aidx0b = d_slots; }     // layer-major: this chunk's rows of the batched encoder output
            else { g.a0 = eout_; g.aidx0 = d_slots; }
            lin(g, L_.w_out); g.M = n; g.N = L_.vocab_pad; g.K = d.joiner; g.kz = kz_out_; g.epi = EPI_PARTIAL; g.out = ws_g_; g.m_stride = ws_mstride_;
            if (round > 0) { g.run_flag = flags_d_ + round; g.run_gen = gen; }
            timed_begin(T_DEC); launch_gemm(g, search_stream_); timed_end(T_DEC);
        }
        DecideArgs a;
        a.ws = ws_g_; a.parts = gemm_partials(n, L_.vocab_pad, kz_out_); a.m_stride = ws_mstride_; a.N = L_.vocab_pad; a.M = n; a.n_valid = d.vocab;
        a.bias = w_ + L_.b_out; a.blank = P_.blank_id; a.early_emit = round == 0 ? 1.0f : 0.0f;
        a.slot_idx = d_slots; a.now_ms = io.now; a.active = active_d_; a.dirty = dirty_d_; a.tok_class = cls_; a.state = gstate_;
        a.rec_ring = rec_d_; a.rec_off = io.rec_off; a.round = round; a.gen = gen; a.rec_slot = io.rec_slot0 + round;
        a.logits_dump = io.dump ? io.dump + (size_t)round * n * d.vocab : nullptr;
        a.dec = dec_params(); a.de_out = dec_table_ ? nullptr : de_; a.ld_de = d.d_model;
        a.run_flags = flags_d_; a.rerun_flags = flags_d_ + 4;
        timed_begin(T_DEC); launch_decide(a, search_stream_); timed_end(T_DEC);
        if (!dec_table_) run_decproj(n, d_slots, dirty_d_, flags_d_ + 4 + round, gen);
    }
}

void Engine::run_chain(int m, bool dump_logits)
{
    const int MB = cfg_.max_batch;
    AdvanceArgs a;
    a.host_ring = ring_h_; a.host_step_off = step_off_h_; a.host_rec_off = rec_off_h_; a.counter = counter_d_; a.index_mask = 2 * step_cap_ - 1;
    a.dst = step_d_; a.dst_stride = MB; a.n_arrays = 3; a.len[0] = a.len[1] = a.len[2] = m; a.rec_off = rec_off_d_;
    a.flags = flags_d_; a.n_flags = 8;
    launch_advance(a, stream_);
    run_encoder_rows(m, step_d_, step_d_ + MB, nullptr);
    run_greedy_rounds(m, dump_logits);
}

// ---------------------------------------------------------------- layer-major step
// Index arrays on the device (stride max_batch): [0] slots (m), [1] ring tails (T x m), [2] session times (T x m),
// [3] slot of every row (T x m).  Row r = t * m + i.  The work is cut into STAGES over a block of time steps [t0, t1):
//   embed(block)      conv front end + embed linear                 -> y, ssq rows of the block
//   layer(l, block)   input half of the gates for the block's rows at once; per time step recurrent half + cell and the
//                     projection; feed-forward over the block's rows   -> y, ssq rows of the block
//   proj(block)       encoder_proj                                   -> eout_lm rows of the block
// layer(l, b) needs layer(l - 1, b) (its input rows) and layer(l, b - 1) (the recurrent state): stages of different layers
// on different blocks are independent: their launches are z-batched into one (see run_lm_wavefront).  Work buffers are row-partitioned,
// so concurrent stages never share a byte.
void Engine::lm_stage_embed(int m, int t0, int t1, hipStream_t st, bool own_ws)
{
    const NetDims &d = L_.dims;
    const int MB = cfg_.max_batch;
    const size_t r0 = (size_t)t0 * m;
    const int rows = (t1 - t0) * m;
    const int *d_tails = step_d_ + MB, *d_rowslot = step_d_ + 3 * MB;
    ConvEmbedArgs ca;
    ca.ring = ring_; ca.ring_frames = ring_frames_; ca.mel = d.mel; ca.seg = d.seg;
    ca.slot_idx = d_rowslot + r0; ca.ring_tail = d_tails + r0;
    for (int i = 0; i < 3; ++i) { ca.w[i] = w_ + L_.conv_w[i]; ca.b[i] = w_ + L_.conv_b[i]; ca.ch[i] = d.conv_ch[i]; ca.stride[i] = d.conv_stride[i]; }
    ca.ch1_per_group = 1;
    for (int k = 8; k > 1; --k) if (d.conv_ch[1] % k == 0) { ca.ch1_per_group = k; break; }
    ca.out = a3_ + r0 * d.f_out * L_.k3; ca.ldo = L_.k3; ca.M = rows;
    if (conv_wt_) { ca.w0t = conv_wt_; ca.w1t = conv_wt_ + (size_t)9 * d.conv_ch[0]; }
    timed_begin(T_CONV); launch_conv_embed(ca, st); timed_end(T_CONV);
    {
        GemmArgs g; g.a0 = a3_ + r0 * d.f_out * L_.k3; g.lda0 = L_.k3; g.K0 = L_.k3; g.wp = w_ + L_.conv_w[2];
        g.M = rows * d.f_out; g.N = d.conv_ch[2]; g.K = L_.k3; g.kz = 1; g.epi = EPI_BIAS_DSWISH; g.out = xin_ + r0 * d.embed_in; g.ldo = d.conv_ch[2]; g.bias = w_ + L_.conv_b[2];
        timed_begin(T_CONV); launch_gemm(g, st); timed_end(T_CONV);
    }
    lm_resid_ssq(xin_ + r0 * d.embed_in, d.embed_in, L_.w_embed, kz_embed_, w_ + L_.b_embed, nullptr, r0, rows, st, own_ws ? ws_fe_ : ws_);
    if (f16_tile_) launch_cvt_f16(y_ + r0 * d.d_model, y16_ + r0 * d.d_model, (size_t)rows * d.d_model, st);      // layer 0 reads binary16 rows
}

// y[r0 .. r0 + rows) = A x W + bias (+ residual) with sums of squares; fused where the tiles own all of K
void Engine::lm_resid_ssq(const float *a, int K, size_t w_off, int kz, const float *bias, const float *resid, size_t r0, int rows, hipStream_t st, float *ws)
{
    if (!ws) ws = ws_;
    const NetDims &d = L_.dims;
    const int G = d.d_model / SSQ_COLS;
    GemmArgs g; g.a0 = a; g.lda0 = K; g.K0 = K; lin(g, w_off);
    g.M = rows; g.N = d.d_model; g.K = K; g.kz = kz; g.tile_ok = tile_ok();
    float *yo = y_ + r0 * d.d_model, *so = ssq_ + r0 * G;
    if (gemm_fullk(rows, d.d_model, kz, false, 1, tile_ok())) {
        g.epi = EPI_RESID_SSQ; g.bias = bias; g.resid = resid; g.ldr = d.d_model; g.out = yo; g.ldo = d.d_model; g.ssq_out = so;
        timed_begin(T_GEMM_OTHER); launch_gemm(g, st); timed_end(T_GEMM_OTHER);
        return;
    }
    g.epi = EPI_PARTIAL; g.out = ws + r0 * d.d_model; g.m_stride = ws_mstride_;
    timed_begin(T_GEMM_OTHER); launch_gemm(g, st); timed_end(T_GEMM_OTHER);
    RowArgs r; r.mode = ROW_RESID_SSQ; r.ws = ws + r0 * d.d_model; r.parts = gemm_partials(rows, d.d_model, kz, 1, tile_ok()); r.m_stride = ws_mstride_; r.N = d.d_model; r.M = rows;
    r.bias = bias; r.resid = resid; r.ldr = d.d_model; r.out = yo; r.ldo = d.d_model; r.ssq_out = so;
    timed_begin(T_ROW); launch_row(r, st); timed_end(T_ROW);
}

// The GEMMs of one layer stage as argument blocks: launched one by one (lm_stage_layer) or, for all layers of a wavefront, in
// one launch each (run_lm_wavefront).
GemmArgs Engine::lm_args_xpart(int l, int m, int t0, int t1) const
{   // input half of the gates for the block's rows: P = (p0 + p1) * scale   (waves 0,1; the recurrent half sits this launch out)
    const NetDims &d = L_.dims;
    const int G = d.d_model / SSQ_COLS;
    const PackedLayout::Layer &o = L_.layers[(size_t)l];
    const size_t b0 = (size_t)t0 * m;
    GemmArgs g; g.a0 = y_ + b0 * d.d_model; g.lda0 = d.d_model; g.K0 = d.d_model;
    g.x_scale.ssq = ssq_ + b0 * G; g.x_scale.groups = G; g.x_scale.inv_n = 1.0f / (float)(d.d_norm ? d.d_norm : d.d_model); g.x_scale.eps = l == 0 ? L_.embed_eps : L_.norm_eps[(size_t)l - 1];
    g.a1 = g.a0; g.lda1 = d.d_model; g.K1 = d.d_model;          // never read (wave_mask)
    lin(g, o.wg); g.M = (t1 - t0) * m; g.N = 4 * d.hidden; g.K = 2 * d.d_model; g.kz = 1; g.epi = EPI_XPART; g.wave_mask = 0x3;
    g.out = p_lm_ + b0 * 4 * d.hidden; g.ldo = 4 * d.hidden;
    if (f16_tile_) {                  // binary16 operands (y16), P stays fp32
        g.a0 = reinterpret_cast<const float *>(y16_ + b0 * d.d_model); g.a1 = g.a0;
        lin16(g, o.wg);
    }
    return g;
}

GemmArgs Engine::lm_args_gates(int l, int m, int t) const
{   // recurrent half + LSTM cell: ((P + p2) + p3) + bias
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    const PackedLayout::Layer &o = L_.layers[(size_t)l];
    const size_t r0 = (size_t)t * m;
    GemmArgs g; g.a0 = y_ + r0 * d.d_model; g.lda0 = d.d_model; g.K0 = d.d_model;      // never read (wave_mask)
    g.a1 = h_ + (size_t)l * S * d.d_model; g.lda1 = d.d_model; g.aidx1 = step_d_; g.K1 = d.d_model;
    lin(g, o.wg); g.M = m; g.N = 4 * d.hidden; g.K = 2 * d.d_model; g.kz = 1; g.epi = EPI_LSTM; g.wave_mask = 0xC;
    g.p_add = p_lm_ + r0 * 4 * d.hidden; g.ldp = 4 * d.hidden;
    g.out = u_ + r0 * d.hidden; g.ldo = d.hidden; g.bias = w_ + o.bg; g.c_state = c_ + (size_t)l * S * d.hidden; g.slot_idx = step_d_; g.hidden = d.hidden;
    if (f16_tile_) {                  // [never read | h16(slot)], u leaves as binary16 only
        g.a0 = reinterpret_cast<const float *>(y16_ + r0 * d.d_model); g.a1 = reinterpret_cast<const float *>(h16_ + (size_t)l * S * d.d_model);
        lin16(g, o.wg); g.out = nullptr; g.out16 = u16_ + r0 * d.hidden;
    }
    return g;
}

// fp32 gates GEMM on the GM_TILE schedule (same chains, same bits as the hand-scheduled K-split tiles -- tests/test_gpu_gates_tile.py):
// measured 2..5 % per feed faster from ~2000 rows per launch (1024 sessions: RTF 0.0533 vs 0.055..0.058, 2048: 0.096..0.097 vs
// 0.099..0.104), slower at 256 sessions (53.5 vs 46.5 us per launch) and at one session.  APRIL_GATES_TILE: 0 never, 1 always,
// 2 (default) from APRIL_GATES_TILE_ROWS (2048) rows per launch.
bool Engine::gates_tile_rows(long rows) const
{
    static const int mode = getenv("APRIL_GATES_TILE") ? atoi(getenv("APRIL_GATES_TILE")) : 2;
    static const long min_rows = getenv("APRIL_GATES_TILE_ROWS") ? atol(getenv("APRIL_GATES_TILE_ROWS")) : 2048;
    return mode == 1 || (mode == 2 && rows >= min_rows);
}

// (the same for the FFN-up GEMM; measurement knob APRIL_FF1_TILE_ROWS, default off)
bool Engine::ff1_tile_rows(long rows) const
{
    static const long min_rows = getenv("APRIL_FF1_TILE_ROWS") ? atol(getenv("APRIL_FF1_TILE_ROWS")) : 0;
    return min_rows > 0 && rows >= min_rows;
}

GemmArgs Engine::sw_args_gates(int l, int m, int t) const
{   // the one-launch gates GEMM of a chunk step (run_encoder_rows) on the rows of chunk t: [norm(y) | h_prev] x Wg, fused LSTM cell
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    const int G = d.d_model / SSQ_COLS;
    const PackedLayout::Layer &o = L_.layers[(size_t)l];
    const size_t r0 = (size_t)t * m;
    GemmArgs g; g.a0 = y_ + r0 * d.d_model; g.lda0 = d.d_model; g.K0 = d.d_model;
    g.x_scale.ssq = ssq_ + r0 * G; g.x_scale.groups = G; g.x_scale.inv_n = 1.0f / (float)(d.d_norm ? d.d_norm : d.d_model); g.x_scale.eps = l == 0 ? L_.embed_eps : L_.norm_eps[(size_t)l - 1];
    g.a1 = h_ + (size_t)l * S * d.d_model; g.lda1 = d.d_model; g.aidx1 = step_d_; g.K1 = d.d_model;
    lin(g, o.wg); g.M = m; g.N = 4 * d.hidden; g.K = 2 * d.d_model; g.kz = 1; g.epi = EPI_LSTM;
    g.out = u_ + r0 * d.hidden; g.ldo = d.hidden; g.bias = w_ + o.bg; g.c_state = c_ + (size_t)l * S * d.hidden; g.slot_idx = step_d_; g.hidden = d.hidden;
    if (gates_tile_rows(m) && cfg_.precision == 0) g.tile_ok = 2;      // (one problem per launch; sw_plan decides again for z-batched launches)
    if (f16_tile_) {                  // binary16 operands: [y16 | h16(slot)], u leaves as binary16 only (its one reader is the projection)
        g.a0 = reinterpret_cast<const float *>(y16_ + r0 * d.d_model); g.a1 = reinterpret_cast<const float *>(h16_ + (size_t)l * S * d.d_model);
        lin16(g, o.wg); g.out = nullptr; g.out16 = u16_ + r0 * d.hidden;
    }
    return g;
}

void Engine::attach_ksplit(GemmArgs &g, int l, int which) const
{
    if (!ks_ws_) return;
    g.ks_ws = ks_ws_ + ((size_t)l * 2 + (size_t)which) * ks_ws_stride_;
    g.ks_cnt = ks_cnt_ + ((size_t)l * 2 + (size_t)which) * ks_cnt_stride_;
}

GemmArgs Engine::lm_args_whr(int l, int m, int t) const
{   // h' = u x Whr ; state write + residual, in one launch however few workgroups (sequential step)
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    const int G = d.d_model / SSQ_COLS;
    const PackedLayout::Layer &o = L_.layers[(size_t)l];
    const size_t r0 = (size_t)t * m;
    GemmArgs g; g.a0 = u_ + r0 * d.hidden; g.lda0 = d.hidden; g.K0 = d.hidden; lin(g, o.whr);
    g.M = m; g.N = d.d_model; g.K = d.hidden; g.kz = kz_hr_; g.tile_ok = tile_ok(); g.force_fullk = 1;
    g.epi = EPI_HR; g.state = h_ + (size_t)l * S * d.d_model; g.ld_state = d.d_model; g.slot_idx = step_d_; g.resid = y_ + r0 * d.d_model; g.ldr = d.d_model;
    g.r_scale.ssq = ssq_ + r0 * G; g.r_scale.groups = G; g.r_scale.inv_n = 1.0f / (float)(d.d_norm ? d.d_norm : d.d_model); g.r_scale.eps = l == 0 ? L_.embed_eps : L_.norm_eps[(size_t)l - 1];
    g.out = xb_ + r0 * d.d_model; g.ldo = d.d_model;
    attach_ksplit(g, l, 0);
    if (f16_tile_) {
        g.a0 = reinterpret_cast<const float *>(u16_ + r0 * d.hidden); lin16(g, o.whr); g.kz = kzx_hr_;
        g.state16 = h16_ + (size_t)l * S * d.d_model; g.out16 = xb16_ + r0 * d.d_model;
    }
    return g;
}

GemmArgs Engine::lm_args_ff1(int l, int m, int t0, int t1) const
{   // FFN up + DoubleSwish, the block's rows
    const NetDims &d = L_.dims;
    const PackedLayout::Layer &o = L_.layers[(size_t)l];
    const size_t b0 = (size_t)t0 * m;
    GemmArgs g; g.a0 = xb_ + b0 * d.d_model; g.lda0 = d.d_model; g.K0 = d.d_model; lin(g, o.wff1);
    g.M = (t1 - t0) * m; g.N = d.ffn; g.K = d.d_model; g.kz = 1; g.epi = EPI_BIAS_DSWISH; g.out = ff_ + b0 * d.ffn; g.ldo = d.ffn; g.bias = w_ + o.bff1;
    if (f16_tile_) { g.a0 = reinterpret_cast<const float *>(xb16_ + b0 * d.d_model); lin16(g, o.wff1); g.out = nullptr; g.out16 = ff16_ + b0 * d.ffn; }
    else if (cfg_.precision == 0 && ff1_tile_rows((long)(t1 - t0) * m)) g.tile_ok = 2;
    return g;
}

GemmArgs Engine::lm_args_ff2(int l, int m, int t0, int t1) const
{   // FFN down + bias + residual + sums of squares in one launch (the wavefront form; lm_resid_ssq picks by occupancy)
    const NetDims &d = L_.dims;
    const int G = d.d_model / SSQ_COLS;
    const PackedLayout::Layer &o = L_.layers[(size_t)l];
    const size_t b0 = (size_t)t0 * m;
    GemmArgs g; g.a0 = ff_ + b0 * d.ffn; g.lda0 = d.ffn; g.K0 = d.ffn; lin(g, o.wff2);
    g.M = (t1 - t0) * m; g.N = d.d_model; g.K = d.ffn; g.kz = kz_ff2_; g.tile_ok = tile_ok(); g.force_fullk = 1;
    g.epi = EPI_RESID_SSQ; g.bias = w_ + o.bff2; g.resid = xb_ + b0 * d.d_model; g.ldr = d.d_model; g.out = y_ + b0 * d.d_model; g.ldo = d.d_model; g.ssq_out = ssq_ + b0 * G;
    attach_ksplit(g, l, 1);
    if (f16_tile_) { g.a0 = reinterpret_cast<const float *>(ff16_ + b0 * d.ffn); lin16(g, o.wff2); g.kz = kzx_ff2_; g.out16 = y16_ + b0 * d.d_model; }
    return g;
}

// A row-epilogue GEMM given in its fused form (EPI_HR / EPI_RESID_SSQ): the partial-plane form of the same GEMM over the
// workspace rows of its output, and the row problem that finishes it (same arithmetic as the fused epilogue)
static GemmArgs partial_form(const GemmArgs &f, float *ws, int m_stride)
{
    GemmArgs g = f;
    g.epi = EPI_PARTIAL; g.force_fullk = 0; g.out = ws; g.m_stride = m_stride;
    g.bias = nullptr; g.resid = nullptr; g.state = nullptr; g.slot_idx = nullptr; g.ssq_out = nullptr; g.r_scale = RowScale(); g.out16 = nullptr; g.state16 = nullptr;
    return g;
}
static RowArgs row_form(const GemmArgs &f, const float *ws, int m_stride, int parts)
{
    RowArgs r; r.ws = ws; r.parts = parts; r.m_stride = m_stride; r.N = f.N; r.M = f.M;
    r.resid = f.resid; r.ldr = f.ldr; r.out = f.out; r.ldo = f.ldo; r.out16 = f.out16;
    if (f.epi == EPI_HR) { r.mode = ROW_HR; r.r_scale = f.r_scale; r.slot_idx = f.slot_idx; r.state = f.state; r.ld_state = f.ld_state; r.state16 = f.state16; }
    else { r.mode = ROW_RESID_SSQ; r.bias = f.bias; r.ssq_out = f.ssq_out; }
    return r;
}

// one row-epilogue GEMM outside the z-batched chains: fused when the plan keeps all of K in the workgroup, else planes + row kernel
void Engine::launch_rowepi(GemmArgs f, size_t ws_row0, hipStream_t st)
{
    f.force_fullk = 0;
    if (gemm_fullk(f.M, f.N, f.kz, false, 1, f.tile_ok)) { timed_begin(T_GEMM_OTHER); launch_gemm(f, st); timed_end(T_GEMM_OTHER); return; }
    float *ws = ws_ + ws_row0 * f.N;
    timed_begin(T_GEMM_OTHER); launch_gemm(partial_form(f, ws, ws_mstride_), st); timed_end(T_GEMM_OTHER);
    timed_begin(T_ROW); launch_row(row_form(f, ws, ws_mstride_, gemm_partials(f.M, f.N, f.kz, 1, f.tile_ok)), st); timed_end(T_ROW);
}

void Engine::lm_stage_layer(int l, int m, int t0, int t1, hipStream_t st)
{
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    const PackedLayout::Layer &o = L_.layers[(size_t)l];
    const size_t b0 = (size_t)t0 * m;
    const int brows = (t1 - t0) * m;
    timed_begin(T_GATES); launch_gemm(lm_args_xpart(l, m, t0, t1), st); timed_end(T_GATES);
    if (f16_tile_) {                  // fp16 tile engines: the same stage on the tile kernels (row epilogues fused or planes + row kernel, the planner's choice)
        for (int t = t0; t < t1; ++t) {
            timed_begin(T_GATES); launch_gemm(lm_args_gates(l, m, t), st); timed_end(T_GATES);
            launch_rowepi(lm_args_whr(l, m, t), (size_t)t * m, st);
        }
        timed_begin(T_GEMM_OTHER); launch_gemm(lm_args_ff1(l, m, t0, t1), st); timed_end(T_GEMM_OTHER);
        launch_rowepi(lm_args_ff2(l, m, t0, t1), b0, st);
        return;
    }
    for (int t = t0; t < t1; ++t) {
        const size_t r0 = (size_t)t * m;
        timed_begin(T_GATES); launch_gemm(lm_args_gates(l, m, t), st); timed_end(T_GATES);
        if (gemm_fullk(m, d.d_model, kz_hr_, true, 1, tile_ok())) {
            timed_begin(T_GEMM_OTHER); launch_gemm(lm_args_whr(l, m, t), st); timed_end(T_GEMM_OTHER);
        } else {
            const GemmArgs f = lm_args_whr(l, m, t);
            GemmArgs g; g.a0 = f.a0; g.lda0 = f.lda0; g.K0 = f.K0; lin(g, o.whr); g.M = m; g.N = d.d_model; g.K = d.hidden; g.kz = kz_hr_; g.tile_ok = tile_ok();
            g.epi = EPI_PARTIAL; g.out = ws_ + r0 * d.d_model; g.m_stride = ws_mstride_;
            timed_begin(T_GEMM_OTHER); launch_gemm(g, st); timed_end(T_GEMM_OTHER);
            RowArgs r; r.mode = ROW_HR; r.ws = ws_ + r0 * d.d_model; r.parts = gemm_partials(m, d.d_model, kz_hr_, 1, tile_ok()); r.m_stride = ws_mstride_; r.N = d.d_model; r.M = m;
            r.resid = f.resid; r.ldr = d.d_model; r.r_scale = f.r_scale; r.out = f.out; r.ldo = d.d_model;
            r.slot_idx = step_d_; r.state = h_ + (size_t)l * S * d.d_model; r.ld_state = d.d_model;
            timed_begin(T_ROW); launch_row(r, st); timed_end(T_ROW);
        }
    }
    timed_begin(T_GEMM_OTHER); launch_gemm(lm_args_ff1(l, m, t0, t1), st); timed_end(T_GEMM_OTHER);
    lm_resid_ssq(ff_ + b0 * d.ffn, d.ffn, o.wff2, kz_ff2_, w_ + o.bff2, xb_ + b0 * d.d_model, b0, brows, st);
}

void Engine::lm_stage_proj(int m, int t0, int t1, hipStream_t st, float *ws)
{
    if (!ws) ws = ws_;
    const NetDims &d = L_.dims;
    const int G = d.d_model / SSQ_COLS;
    const size_t b0 = (size_t)t0 * m;
    const int brows = (t1 - t0) * m;
    RowScale ys; ys.ssq = ssq_ + b0 * G; ys.groups = G; ys.inv_n = 1.0f / (float)(d.d_norm ? d.d_norm : d.d_model); ys.eps = L_.norm_eps[(size_t)d.n_layers - 1];
    GemmArgs g; g.a0 = y_ + b0 * d.d_model; g.lda0 = d.d_model; g.K0 = d.d_model; lin(g, L_.w_encproj);
    g.M = brows; g.N = d.joiner; g.K = d.d_model; g.kz = kz_proj_; g.tile_ok = tile_ok();
    float *eo = eout_lm_ + b0 * d.joiner;
    if (gemm_fullk(brows, d.joiner, kz_proj_, true, 1, tile_ok())) {
        g.force_fullk = 1;
        g.epi = EPI_SLOT_STORE; g.bias = w_ + L_.b_encproj; g.out = eo; g.ldo = d.joiner; g.x_scale = ys;
        timed_begin(T_GEMM_OTHER); launch_gemm(g, st); timed_end(T_GEMM_OTHER);
    } else {
        g.epi = EPI_PARTIAL; g.out = ws + b0 * d.d_model; g.m_stride = ws_mstride_;     // (joiner width == a d_model-wide slice or less: see the constructor's workspace size)
        timed_begin(T_GEMM_OTHER); launch_gemm(g, st); timed_end(T_GEMM_OTHER);
        RowArgs r; r.mode = ROW_SLOT_STORE; r.ws = ws + b0 * d.d_model; r.parts = gemm_partials(brows, d.joiner, kz_proj_, 1, tile_ok()); r.m_stride = ws_mstride_; r.N = d.joiner; r.M = brows;
        r.bias = w_ + L_.b_encproj; r.out = eo; r.ldo = d.joiner; r.r_scale = ys;
        timed_begin(T_ROW); launch_row(r, st); timed_end(T_ROW);
    }
}

// Time steps per block of the long-feed wavefront (run_lm_wavefront).  A macro step costs its block's recurrent launches (2 per
// time step, whatever the number of active layers) plus ~80 us of block stages (input halves, feed-forward, front end: weight
// streams that do not depend on the block length), and the wavefront runs NB + L + 1 macro steps, L + 1 of them fill / drain:
// (T / blk + L + 1) (c_step blk + c_block) is smallest near blk = sqrt(T c_block / ((L + 1) c_step)) ~ 0.75 sqrt(T) at aprilv0
// size.  Measured, 60 s in one call (T = 1498): blk 10 / 16 / 20 / 24 / 32 -> 65.5 / 61.6 / 61.2 / 60.6 / 60.2 ms.
// APRIL_LM_BLOCK pins it.  Results do not depend on the block length (same chains in the same order).
static int lm_block_env()
{
    static const int v = getenv("APRIL_LM_BLOCK") ? std::max(1, atoi(getenv("APRIL_LM_BLOCK"))) : 0;
    return v;
}
static int lm_block_steps(int T)
{
    if (lm_block_env() > 0) return lm_block_env();
    const int b = (int)(0.75 * std::sqrt((double)std::max(1, T)) + 0.5);
    return std::max(6, std::min(32, b));
}
// (feeds up to this many chunks run as one layer-major chain, captured as a graph: too few blocks for a wavefront to fill)
static int lm_wavefront_min_chunks() { return lm_block_env() > 0 ? lm_block_env() : 10; }

static bool lm_wavefront_on()
{
    static const int v = getenv("APRIL_LM_WAVEFRONT") ? atoi(getenv("APRIL_LM_WAVEFRONT")) : 1;
    return v != 0;
}

void Engine::run_lm_chain(int m, int T, bool dump_logits)
{
    const NetDims &d = L_.dims;
    const int MB = cfg_.max_batch;
    const int L = d.n_layers;
    AdvanceArgs a;
    a.host_ring = ring_h_; a.host_step_off = step_off_h_; a.host_rec_off = rec_off_h_; a.counter = counter_d_; a.index_mask = 2 * step_cap_ - 1;
    a.dst = step_d_; a.dst_stride = MB; a.n_arrays = 4; a.len[0] = m; a.len[1] = a.len[2] = a.len[3] = m * T; a.rec_off = rec_off_d_;
    a.flags = flags_d_; a.n_flags = 8;
    launch_advance(a, stream_);
    lm_stage_embed(m, 0, T, stream_);
    for (int l = 0; l < L; ++l) lm_stage_layer(l, m, 0, T, stream_);
    lm_stage_proj(m, 0, T, stream_);
    // the search stays sequential in time (the decoder input of chunk t + 1 depends on the tokens of chunk t)
    for (int t = 0; t < T; ++t) run_greedy_rounds(m, dump_logits, t, eout_lm_ + (size_t)t * m * d.joiner);
}

// Wavefront form of the layer-major step (long feeds: T > block).  Time is cut into blocks of `blk` steps; at macro step W
// layer l works on block W - 1 - l, the front end on block W, encoder_proj on block W - L - 1 -- all of them independent of
// one another.  The SAME launch of different layers is therefore ONE launch (gemm_f32_zkernel: blockIdx.z picks the layer's
// argument block): per macro step one launch for the input halves of the gates, 2 per time step of the block for the
// recurrence, one each for the feed-forward GEMMs -- (2 blk + 3) launches for L layer stages instead of L (2 blk + 3), on ONE
// stream, no cross-stream events.  At one session every recurrent kernel is a latency-bound launch of a few microseconds, and
// twelve of them in one launch cost about the same as one.  The search (sequential in time) follows on the engine's stream,
// one block behind, as ONE captured graph per block whose arguments do not depend on the block (launch_block_setup).
// Arithmetic: the kernels and their arguments are those of lm_stage_layer; only the grouping into launches differs.
void Engine::run_lm_wavefront(int m, int T, bool dump_logits)
{
    const NetDims &d = L_.dims;
    const int MB = cfg_.max_batch;
    const int L = d.n_layers;
    const int blk = lm_block_steps(T);
    const int NB = (T + blk - 1) / blk;
    AdvanceArgs a;
    a.host_ring = ring_h_; a.host_step_off = step_off_h_; a.host_rec_off = rec_off_h_; a.counter = counter_d_; a.index_mask = 2 * step_cap_ - 1;
    a.dst = step_d_; a.dst_stride = MB; a.n_arrays = 4; a.len[0] = m; a.len[1] = a.len[2] = a.len[3] = m * T; a.rec_off = rec_off_d_;
    a.flags = flags_d_; a.n_flags = 8;
    launch_advance(a, stream_);
    hipStream_t cs = lm_stream_;
    HIP_CHECK(hipEventRecord(lm_events_[0], stream_));                 // the index block is on the device
    HIP_CHECK(hipStreamWaitEvent(cs, lm_events_[0], 0));

    // argument blocks of this step: (NB + L + 1) macro steps x at most (2 blk + 6) launches x L layers, in one of three
    // regions that are reused round robin once the step that used them has run (the host runs ahead of the GPU by whole steps)
    const size_t need = (size_t)(NB + L + 1) * (size_t)(2 * blk + 6) * (size_t)L;
    if (need > zargs_region_) {
        HIP_CHECK(hipStreamSynchronize(cs));
        HipLegacyLock regrow_guard;                   // (see fbank())
        if (zargs_h_) { (void)hipHostFree(zargs_h_); (void)hipFree(zargs_d_); }
        zargs_region_ = need + need / 4;
        zargs_h_ = hmalloc<GemmArgs>(3 * zargs_region_); zargs_d_ = dmalloc<GemmArgs>(3 * zargs_region_);
        for (int i = 0; i < 3; ++i) zargs_busy_[i] = false;
    }
    const int zslot = zargs_next_;
    zargs_next_ = (zargs_next_ + 1) % 3;
    if (zargs_busy_[zslot]) HIP_CHECK(hipEventSynchronize(zargs_done_[zslot]));
    zargs_pos_ = (size_t)zslot * zargs_region_;

    static const bool timing = getenv("APRIL_LM_TIMING") != nullptr;        // measurement: host time of the enqueue by part
    double tacc[6] = {0, 0, 0, 0, 0, 0};
    auto now = [&]() { return timing ? std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0; };
    struct Batch { size_t off; int n; };
    std::vector<Batch> plan;                               // the z-batched launches of all macro steps, in order
    std::vector<size_t> plan_end((size_t)(NB + L + 1), 0); // macro step W owns plan[plan_end[W - 1] .. plan_end[W])
    std::vector<GemmArgs> items;
    struct Act { int l, t0, t1; };
    std::vector<Act> act;
    double tq = now();
    auto lap = [&](int i) { if (timing) { const double t = now(); tacc[i] += t - tq; tq = t; } };
    // The argument blocks of the WHOLE step first, then ONE host-to-device copy: a copy between the kernels of every macro step
    // stalls the stream each time (measured: 13 us per macro step, 2.5 ms per minute of audio); the blocks depend on (m, T) and
    // buffer addresses only.
    const size_t first = zargs_pos_;
    for (int W = 0; W <= NB + L; ++W) {
        act.clear();
        for (int l = 0; l < L; ++l) { const int b = W - 1 - l; if (b >= 0 && b < NB) act.push_back({l, b * blk, std::min(T, (b + 1) * blk)}); }
        auto flush = [&]() {
            if (items.empty()) return;
            stage_gemm_z(items.data(), (int)items.size(), zargs_h_ + zargs_pos_);
            plan.push_back({zargs_pos_, (int)items.size()});
            zargs_pos_ += items.size(); items.clear();
        };
        // the block stages come in at most two lengths (full blocks, the tail): one launch per length
        auto by_len = [&](auto make) {
            for (int pass = 0; pass < 2; ++pass) {
                for (const Act &x : act) if (((x.t1 - x.t0) == blk) == (pass == 0)) items.push_back(make(x));
                flush();
            }
        };
        by_len([&](const Act &x) { return lm_args_xpart(x.l, m, x.t0, x.t1); });
        for (int i = 0; i < blk; ++i) {
            for (const Act &x : act) if (x.t0 + i < x.t1) items.push_back(lm_args_gates(x.l, m, x.t0 + i));
            flush();
            for (const Act &x : act) if (x.t0 + i < x.t1) items.push_back(lm_args_whr(x.l, m, x.t0 + i));
            flush();
        }
        by_len([&](const Act &x) { return lm_args_ff1(x.l, m, x.t0, x.t1); });
        by_len([&](const Act &x) { return lm_args_ff2(x.l, m, x.t0, x.t1); });
        plan_end[(size_t)W] = plan.size();
    }
    lap(0);
    if (zargs_pos_ > first)
        HIP_CHECK(hipMemcpyAsync(zargs_d_ + first, zargs_h_ + first, (zargs_pos_ - first) * sizeof(GemmArgs), hipMemcpyHostToDevice, cs));
    lap(1);
    for (int W = 0; W <= NB + L; ++W) {
        if (W < NB) lm_stage_embed(m, W * blk, std::min(T, (W + 1) * blk), cs);
        lap(2);
        for (size_t k = W ? plan_end[(size_t)W - 1] : 0; k < plan_end[(size_t)W]; ++k) launch_gemm_z(zargs_h_ + plan[k].off, plan[k].n, zargs_d_ + plan[k].off, cs);
        lap(3);
        const int bp = W - L - 1;
        if (bp >= 0) {
            const int t0 = bp * blk, t1 = std::min(T, t0 + blk), len = t1 - t0;
            lm_stage_proj(m, t0, t1, cs);
            hipEvent_t e = lm_events_[1 + (size_t)(bp % (int)(lm_events_.size() - 1))];
            HIP_CHECK(hipEventRecord(e, cs));
            HIP_CHECK(hipStreamWaitEvent(stream_, e, 0));
            // the block's search: bookkeeping at fixed places, then the block-independent launch chain
            BlockSetupArgs bs;
            bs.now_src = step_d_ + 2 * MB + (size_t)t0 * m; bs.now_dst = lm_now_d_; bs.rows_dst = lm_rows_d_; bs.row0 = t0 * m; bs.count = len * m;
            bs.rec_off_step = rec_off_d_; bs.rec_off_block = lm_rec_off_d_; bs.rec_add = t0 * 3 * m; bs.flags = flags_d_; bs.n_flags = 8;
            launch_block_setup(bs, stream_);
            lap(4);
            auto search = [&]() {
                for (int i = 0; i < len; ++i) {
                    GreedyIo io;
                    io.gen = i + 1; io.now = lm_now_d_ + (size_t)i * m; io.eout = eout_lm_; io.eout_rows = lm_rows_d_ + (size_t)i * m;
                    io.rec_off = lm_rec_off_d_; io.rec_slot0 = i * 3;
                    io.dump = dump_logits ? logits_ + (size_t)(t0 + i) * 3 * m * d.vocab : nullptr;
                    run_greedy_rounds(m, io);
                }
            };
            if (!use_graphs_ || dump_logits) { search(); continue; }
            const std::pair<int, int> key(m * 2 + flight_parity_, len);      // (round flags and encoder outputs are per flight parity)
            auto it = lm_search_graphs_.find(key);
            if (it == lm_search_graphs_.end()) {
                if (lm_search_graphs_.size() >= 64) { HIP_CHECK(hipStreamSynchronize(stream_)); /* execs launched earlier in this flight may still run */ for (auto &g : lm_search_graphs_) (void)hipGraphExecDestroy(g.second); lm_search_graphs_.clear(); }
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                HipLegacyLock capture_guard_1;      // (no legacy-stream call of any thread during the capture: see engine.h)
                HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed));
                search();
                HIP_CHECK(hipStreamEndCapture(stream_, &graph));
                HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                HIP_CHECK(hipGraphDestroy(graph));
                it = lm_search_graphs_.emplace(key, exec).first;
            }
            HIP_CHECK(hipGraphLaunch(it->second, stream_));
            lap(5);
        }
    }
    HIP_CHECK(hipEventRecord(zargs_done_[zslot], cs));
    zargs_busy_[zslot] = true;
    if (timing) fprintf(stderr, "lm wavefront %d x %d: host us: args %.0f, arg copy %.0f, front end %.0f, layer launches %.0f, proj + events + setup %.0f, search graph %.0f\n",
                        m, T, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5]);
}

// ---------------------------------------------------------------- chunk steps of one feed as a wavefront
// A 100 ms feed carries 2..3 chunks per session.  Chunk t + 1 of layer l does not depend on chunk t of layer l + 1, so the T
// chunk steps of a feed run as a wavefront over the layers: at macro step W layer l works on chunk W - 1 - l, and the same
// launch of the (up to T) active layers is ONE z-batched launch.  Same kernels and arguments as T chunk steps in a row
// (run_encoder_rows), on the rows of chunk t instead of rows 0..m-1: bit-identical.  What it buys: 4 (L + T - 1) layer
// launches per feed instead of 4 L T, and -- the reason it pays at 256 sessions -- a launch holds T x the workgroups, so the
// planner picks larger tiles for the N = d_model GEMMs (32x32 instead of 16x32 at 256 rows: 1.5 x fewer operand bytes per
// flop, the bound of those kernels) and co-resident workgroups overlap each other's prologue / epilogue.
// The argument blocks depend on (m, T) only: built once per shape, kept in device memory, and the whole chain is one graph.
static bool lm_gates_split_on()
{
    static const bool on = [] { const char *e = getenv("APRIL_GATES_SPLIT"); return !(e && *e && atoi(e) == 0); }();
    return on;
}

Engine::SwPlan &Engine::sw_plan(int m, int T)
{
    // (the argument blocks point into the parity's buffers; plans built under the gates clock carry stamp slots and are kept apart:
    // negative m)
    const std::pair<int, int> key(gclk_ ? -m : m, T * 2 + flight_parity_);
    auto it = sw_plans_.find(key);
    if (it != sw_plans_.end()) return it->second;
    if (sw_plans_.size() >= 64) {
        sync();
        HipLegacyLock evict_guard;                    // (see fbank())
        for (auto &p : sw_plans_) { if (p.second.graph) (void)hipGraphExecDestroy(p.second.graph); for (hipGraphExec_t x : p.second.g3) if (x) (void)hipGraphExecDestroy(x); if (p.second.dev) (void)hipFree(p.second.dev); if (p.second.rdev) (void)hipFree(p.second.rdev); if (p.second.pf_dev) (void)hipFree(p.second.pf_dev); }
        sw_plans_.clear();
    }
    SwPlan &p = sw_plans_[key];
    const NetDims &d = L_.dims;
    const int L = d.n_layers;
    std::vector<GemmArgs> items;
    std::vector<RowArgs> rows;
    for (int W = 0; W <= T + L; ++W) {
        for (int kind = 0; kind < 4; ++kind) {
            items.clear(); rows.clear();
            int n_act = 0;
            for (int l = 0; l < L; ++l) { const int t = W - 1 - l; if (t >= 0 && t < T) ++n_act; }
            if (n_act == 0) continue;
            // The N = d_model GEMMs (projection, FFN down) on the GM_TILE schedule: when the n_act problems of this launch give the
            // chip too few 64 x 64 tiles, K is cut across workgroups and ONE z-batched row launch finishes the slab tree with the
            // epilogue (state write + residual / bias + residual + sums of squares) -- the same arithmetic as the fused form.
            const int kz = kind == 1 ? kz_hr() : kz_ff2();
            const int tk = f16_tile_ ? 2 : tile_ok();
            const bool split = (kind == 1 || kind == 3) && tk && (tk == 2 || gemm_tile_planned(m, d.d_model, kz, n_act)) && !gemm_fullk(m, d.d_model, kz, false, n_act, tk);
            for (int l = 0; l < L; ++l) {
                const int t = W - 1 - l;
                if (t < 0 || t >= T) continue;
                GemmArgs g = kind == 0 ? sw_args_gates(l, m, t) : kind == 1 ? lm_args_whr(l, m, t) : kind == 2 ? lm_args_ff1(l, m, t, t + 1) : lm_args_ff2(l, m, t, t + 1);
                if (kind == 0 && cfg_.precision == 0) g.tile_ok = gates_tile_rows((long)m * n_act) ? 2 : 0;
                if (kind == 2 && cfg_.precision == 0) g.tile_ok = ff1_tile_rows((long)m * n_act) ? 2 : 0;
                if (split) {
                    float *ws = ws_ + (size_t)t * m * d.d_model;
                    rows.push_back(row_form(g, ws, ws_mstride_, gemm_partials(m, d.d_model, kz, n_act, tk)));
                    g = partial_form(g, ws, ws_mstride_);
                } else if (tk == 2) g.force_fullk = 0;
                items.push_back(g);
            }
            // The gates launch of four and more problems (merged flights: two feeds stepped as one wavefront of up to seven chunk steps) goes
            // out as launches of two or three: the hand-scheduled 64 x 64 kernel holds 512 workgroups at a time, two problems are exactly one
            // round and three run on walking workgroups, while four to six in one grid measured 82.5 / 110 / 125 us against 76.8 / 97.3 / 117.8
            // for the pieces (rocprofv3 by grid size, round 6) -- and every launch of a kernel name then holds one problem count, which is
            // what makes a per-kernel trace comparable with the gates clock.  (fp32, below the GM_TILE row count only.)
            std::vector<int> groups;
            {
                int left = (int)items.size();
                const bool cut = kind == 0 && cfg_.precision == 0 && left >= 4 && !gates_tile_rows((long)m * n_act) && lm_gates_split_on();
                while (cut && left > 0) { const int take = (left == 4 || left == 2) ? 2 : (left >= 3 ? 3 : left); groups.push_back(take); left -= take; }
                if (groups.empty()) groups.push_back((int)items.size());
            }
            size_t first = 0;
            for (size_t gi = 0; gi < groups.size(); ++gi) {
                const int gn = groups[gi];
                if (kind == 0 && gclk_ && gclk_slots_ && gclk_used_ < GCLK_SLOTS) {      // one slot per gates LAUNCH: all its problems point at it
                    p.stamp_slots.push_back(std::make_pair(gclk_used_, (long)m * gn)); p.stamp_n.push_back(gn);
                    for (int i = 0; i < gn; ++i) items[first + (size_t)i].stamp = gclk_slots_ + (size_t)gclk_used_ * STAMP_WORDS;
                    ++gclk_used_;
                }
                SwPlan::Batch b; b.off = p.host.size(); b.n = gn; b.macro = W; b.kind = kind; b.roff = p.rhost.size(); b.rn = gi == 0 ? (int)rows.size() : 0;
                b.pf_off = p.pf_host.size(); b.pf_n = b.n;
                for (int i = 0; i < gn; ++i) { const GemmArgs &g = items[first + (size_t)i]; PrefetchItem pi; pi.ptr = g.wp; pi.bytes = (unsigned long long)g.N * (unsigned long long)g.K * (g.wt == 1 ? 2u : 4u); p.pf_host.push_back(pi); }
                p.host.resize(p.host.size() + (size_t)gn);
                stage_gemm_z(items.data() + first, gn, p.host.data() + b.off);
                if (gi == 0) p.rhost.insert(p.rhost.end(), rows.begin(), rows.end());
                p.batches.push_back(b);
                first += (size_t)gn;
            }
        }
    }
    p.dev = dmalloc<GemmArgs>(p.host.size());
    // (stream-ordered on the engine's own stream: a blocking hipMemcpy goes through the legacy stream, which HIP refuses while
    // ANOTHER engine's stepping thread is capturing a graph -- two engines on one device, several models in one process)
    HIP_CHECK(hipMemcpyAsync(p.dev, p.host.data(), p.host.size() * sizeof(GemmArgs), hipMemcpyHostToDevice, stream_));
    if (!p.rhost.empty()) {
        p.rdev = dmalloc<RowArgs>(p.rhost.size());
        HIP_CHECK(hipMemcpyAsync(p.rdev, p.rhost.data(), p.rhost.size() * sizeof(RowArgs), hipMemcpyHostToDevice, stream_));
    }
    if (prefetch_ && !p.pf_host.empty()) {
        p.pf_dev = dmalloc<PrefetchItem>(p.pf_host.size());
        HIP_CHECK(hipMemcpyAsync(p.pf_dev, p.pf_host.data(), p.pf_host.size() * sizeof(PrefetchItem), hipMemcpyHostToDevice, stream_));
    }
    return p;
}

// Measurement (APRIL_CHAIN_STREAMS=1): the layer part of a split feed as T per-chunk chains on T streams instead of z-batched
// wavefront launches on one.  Chunk t's chain is G -> P -> U -> D layer after layer on the rows of chunk t (one 256-row problem per
// launch); the only coupling is the recurrent state: G(l, t) waits for P(l, t - 1) by an event.  Same kernels, same arguments,
// same chains as the z-batched launches (row-partitioned work buffers), so the results are bit-identical; what changes is
// that the chains' launches overlap each other's ramp, prologue and epilogue.  Eager launches (events cross the streams).
void Engine::run_sw_layers_chains(int m, int T)
{
    const NetDims &d = L_.dims;
    const int L = d.n_layers;
    while ((int)chain_streams_.size() < T - 1) { hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); chain_streams_.push_back(s); }
    while (chain_ev_.size() < (size_t)L * (size_t)T) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); chain_ev_.push_back(e); }
    auto st = [&](int t) { return t == 0 ? stream_ : chain_streams_[(size_t)t - 1]; };
    for (int t = 1; t < T; ++t) join(st(t), stream_);                    // behind the front end (and the flight before)
    for (int W = 1; W <= T + L - 1; ++W) {                               // issue in wavefront order so that no stream runs dry on the host's account
        for (int l = 0; l < L; ++l) {
            const int t = W - 1 - l;
            if (t < 0 || t >= T) continue;
            hipStream_t s = st(t);
            if (t > 0) HIP_CHECK(hipStreamWaitEvent(s, chain_ev_[(size_t)l * T + (size_t)t - 1], 0));
            launch_gemm(sw_args_gates(l, m, t), s);
            launch_rowepi(lm_args_whr(l, m, t), (size_t)t * m, s);          // (fused, or planes + row kernel: the planner's choice for one problem)
            if (t + 1 < T) HIP_CHECK(hipEventRecord(chain_ev_[(size_t)l * T + (size_t)t], s));
            launch_gemm(lm_args_ff1(l, m, t, t + 1), s);
            launch_rowepi(lm_args_ff2(l, m, t, t + 1), (size_t)t * m, s);
        }
    }
    for (int t = 1; t < T; ++t) join(stream_, st(t));
}

// part 0: index fetch + front end (stream fe), 1: the layer wavefront + encoder_proj (stream ly), 2: the searches (stream sr);
// parts < 0: all three in order on one stream
void Engine::run_sw_chain(int m, int T, bool dump_logits, const SwPlan &p, int part, hipStream_t st)
{
    const NetDims &d = L_.dims;
    const int MB = cfg_.max_batch;
    if (part < 0 || part == 0) {
        AdvanceArgs a;
        a.host_ring = ring_h_; a.host_step_off = step_off_h_; a.host_rec_off = rec_off_h_; a.counter = counter_d_; a.index_mask = 2 * step_cap_ - 1;
        a.dst = step_d_; a.dst_stride = MB; a.n_arrays = 4; a.len[0] = m; a.len[1] = a.len[2] = a.len[3] = m * T; a.rec_off = rec_off_d_;
        a.flags = flags_d_; a.n_flags = 8;
        launch_advance(a, st);
        // Front end and encoder_proj do not take part in the wavefront: one launch chain each over all T x m rows (as in the
        // layer-major step), before the first and after the last macro step; the searches follow in time order.
        lm_stage_embed(m, 0, T, st, part == 0);
    }
    if (part < 0 || part == 1) {
        static const int cls_of[4] = {T_GATES, T_GEMM_OTHER, T_GEMM_OTHER, T_GEMM_OTHER};
        // weight prefetch (kernels.h launch_prefetch): while batch i runs, the side stream touches the weights of batch i + 1 -- released
        // by an event in front of batch i, so it is never more than one launch ahead (the cache holds a few launches' weights, not a step's)
        const bool pf = prefetch_ && p.pf_dev && !profiling_;
        for (size_t bi = 0; bi < p.batches.size(); ++bi) {      // macro steps in order; inside one: gates, projection, FFN up, FFN down of the active layers
            const SwPlan::Batch &b = p.batches[bi];
            if (pf && bi + 1 < p.batches.size()) {
                const SwPlan::Batch &nx = p.batches[bi + 1];
                hipEvent_t e = pf_ev_[pf_pos_++ & 7];
                HIP_CHECK(hipEventRecord(e, st));
                HIP_CHECK(hipStreamWaitEvent(pf_stream_, e, 0));
                launch_prefetch(p.pf_dev + nx.pf_off, nx.pf_n, pf_stream_);
            }
            timed_begin(cls_of[b.kind]); launch_gemm_z(p.host.data() + b.off, b.n, p.dev + b.off, st); timed_end(cls_of[b.kind]);
            if (b.rn > 0) { timed_begin(T_ROW); launch_row_z(p.rhost.data() + b.roff, b.rn, p.rdev + b.roff, st); timed_end(T_ROW); }
        }
        if (pf && p.batches.size() > 1) {
            hipEvent_t e = pf_ev_[pf_pos_++ & 7];
            HIP_CHECK(hipEventRecord(e, pf_stream_));
            HIP_CHECK(hipStreamWaitEvent(st, e, 0));
        }
        if (part < 0) lm_stage_proj(m, 0, T, st);
    }
    if (part < 0 || part == 2) {
        // (split feed: encoder_proj feeds only the search, so it leaves the layer stream with it; its split-K planes, if any, go to
        // a workspace of its own: ws_ belongs to the layers of the next flight by then)
        if (part == 2) lm_stage_proj(m, 0, T, st, ws_sr_);
        hipStream_t keep = search_stream_;
        search_stream_ = st;
        for (int t = 0; t < T; ++t) run_greedy_rounds(m, dump_logits, t, eout_lm_ + (size_t)t * m * d.joiner);
        search_stream_ = keep;
    }
}

int Engine::lm_step(int m, int T, const int *slots, const int *ring_tails, const int *now_ms, float *logits_out, int mode)
{
    const int rows = m * T;
    if (m <= 0 || T <= 0 || rows > cfg_.max_batch || !flight_has_room(rows + m, 1)) { LOGE("engine: layer-major step %d x %d does not fit (max batch %d)", m, T, cfg_.max_batch); abort(); }
    HIP_CHECK(hipSetDevice(cfg_.device));
    const NetDims &d = L_.dims;
    if (!p_lm_) {
        p_lm_ = dmalloc<float>((size_t)cfg_.max_batch * 4 * d.hidden);
        for (int p = 0; p < 2; ++p) eout_lm_buf_[p] = dmalloc<float>((size_t)cfg_.max_batch * d.joiner);
        eout_lm_ = eout_lm_buf_[flight_parity_];
    }
    if (!lm_stream_) {               // the wavefront's stream, events and block bookkeeping (created outside any stream capture)
        HIP_CHECK(hipStreamCreateWithFlags(&lm_stream_, hipStreamNonBlocking));
        for (int i = 0; i < 9; ++i) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); lm_events_.push_back(e); }
        for (int i = 0; i < 3; ++i) HIP_CHECK(hipEventCreateWithFlags(&zargs_done_[i], hipEventDisableTiming));
        lm_now_d_ = dmalloc<int>((size_t)cfg_.max_batch); lm_rows_d_ = dmalloc<int>((size_t)cfg_.max_batch); lm_rec_off_d_ = dmalloc<int>(1);
    }
    const int k = next_step_index();
    int *blk = ring_h_ + ring_pos_;
    memcpy(blk, slots, (size_t)m * 4);
    memcpy(blk + m, ring_tails, (size_t)rows * 4);
    memcpy(blk + m + rows, now_ms, (size_t)rows * 4);
    for (int t = 0; t < T; ++t) memcpy(blk + m + 2 * rows + (size_t)t * m, slots, (size_t)m * 4);
    step_off_h_[k] = (int)ring_pos_; rec_off_h_[k] = (int)rec_pos_;
    ring_pos_ += (size_t)m + 3 * (size_t)rows; rec_pos_ += (size_t)3 * rows;
    std::lock_guard<std::mutex> cg(capture_mu_);
    if (mode == 1) {                 // the chunk steps of one feed as a wavefront over the layers (run_sw_chain)
        // a shape is captured into graphs the SECOND time it is seen (by either flight parity): capturing costs a few
        // milliseconds, which a batch shape that occurs once (sessions joining and leaving) never earns back; its launches go
        // out one by one (~0.25 ms of host time).  When it is captured, it is captured for BOTH parities: which parity a shape
        // meets depends on the history of flights (feeds alternate between 2 and 3 chunks, parities alternate too, and one merged
        // or empty flight flips the pairing), and a capture in the middle of a stream is a 3 ms hiccup.
        const int par = flight_parity_;
        sw_plan(m, T);
        SwPlan *pp = &sw_plans_.find(std::make_pair(gclk_ ? -m : m, T * 2 + par))->second;
        int &uses = sw_uses_[std::make_pair(m, T)];
        const bool graphs = use_graphs_ && !profiling_ && !logits_out && (pp->graph || pp->g3[0] || ++uses >= 2);
        // (split: only the FIRST step of a flight: the per-parity buffers keep neighbouring FLIGHTS apart, a second step of the same
        // flight -- a flush runs several -- would have its index fetch and front end overwrite what the first step's layers and
        // search still read; it takes the one-stream path, behind everything the first step put on F and S)
        // (gates clock on: one stream for everything, so that no other stream's kernel shares the CUs with a gates launch while it times itself;
        // the feeds still arrive pipelined, the GPU stays busy and at speed)
        const bool split = graphs && split_streams_ > 0 && overlap_hint_ && flight_steps_ == 1 && !gclk_;
        hipStream_t fe = split_streams_ >= 2 ? f_stream_ : stream_;
        if (graphs && (split ? !pp->g3[0] : !pp->graph)) {
            for (int k2 = 0; k2 < 2; ++k2) {
                const int q = k2 == 0 ? 1 - par : par;          // the other parity first, this one last (the member pointers end where they were)
                select_parity(q);
                SwPlan &pl = sw_plan(m, T);
                if (split) {
                    if (pl.g3[0]) continue;
                    hipStream_t on[3] = {fe, stream_, s_stream_};
                    for (int part = 0; part < 3; ++part) {
                        hipGraph_t graph = nullptr;
                        HipLegacyLock capture_guard_2;      // (no legacy-stream call of any thread during the capture: see engine.h)
                        HIP_CHECK(hipStreamBeginCapture(on[part], hipStreamCaptureModeRelaxed));
                        run_sw_chain(m, T, false, pl, part, on[part]);
                        HIP_CHECK(hipStreamEndCapture(on[part], &graph));
                        HIP_CHECK(hipGraphInstantiate(&pl.g3[part], graph, nullptr, nullptr, 0));
                        HIP_CHECK(hipGraphDestroy(graph));
                    }
                } else {
                    if (pl.graph) continue;
                    hipGraph_t graph = nullptr;
                    HipLegacyLock capture_guard_3;      // (no legacy-stream call of any thread during the capture: see engine.h)
                    HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed));
                    run_sw_chain(m, T, false, pl, -1, stream_);
                    HIP_CHECK(hipStreamEndCapture(stream_, &graph));
                    HIP_CHECK(hipGraphInstantiate(&pl.graph, graph, nullptr, nullptr, 0));
                    HIP_CHECK(hipGraphDestroy(graph));
                }
            }
            pp = &sw_plans_.find(std::make_pair(gclk_ ? -m : m, T * 2 + par))->second;      // (a plan-cache eviction in between would have moved it)
        }
        SwPlan &p = *pp;
        if (split) {
            // split feed: front end on F, layers on M, search on S, chained by events inside the feed; across feeds the three
            // parts of neighbouring flights overlap (see "streams" above).  split_streams_ == 1 keeps the front end on M.
            if (fe == f_stream_) {
                if (m_unseen_by_f_) { join(f_stream_, stream_); m_unseen_by_f_ = false; }
            } else if (f_unseen_by_m_) { join(stream_, f_stream_); f_unseen_by_m_ = false; }       // (the fbank launch of this flight)
            if (m_unseen_by_s_) { join(s_stream_, stream_); m_unseen_by_s_ = false; }
            StreamTrace *tr = trace_slot();
            if (tr) HIP_CHECK(hipEventRecord(tr->ev[0], fe));
            HIP_CHECK(hipGraphLaunch(p.g3[0], fe));
            if (tr) HIP_CHECK(hipEventRecord(tr->ev[1], fe));
            if (fe == f_stream_) { join(stream_, f_stream_); f_unseen_by_m_ = false; }
            if (tr) HIP_CHECK(hipEventRecord(tr->ev[2], stream_));
            static const bool chains = getenv("APRIL_CHAIN_STREAMS") && atoi(getenv("APRIL_CHAIN_STREAMS")) != 0;
            if (chains && cfg_.precision == 0) run_sw_layers_chains(m, T);
            else HIP_CHECK(hipGraphLaunch(p.g3[1], stream_));
            if (tr) HIP_CHECK(hipEventRecord(tr->ev[3], stream_));
            join(s_stream_, stream_);
            if (tr) HIP_CHECK(hipEventRecord(tr->ev[4], s_stream_));
            HIP_CHECK(hipGraphLaunch(p.g3[2], s_stream_));
            if (tr) { HIP_CHECK(hipEventRecord(tr->ev[5], s_stream_)); tr->m = m; tr->T = T; tr->used = true; }
            s_unseen_by_m_ = true; flight_tail_s_ = true;
            if (fe != f_stream_) m_unseen_by_f_ = true;          // (front-end kernels on M read ring rows: the next fbank waits for them)
            return k;
        }
        general_prologue();
        if (graphs) {
            HIP_CHECK(hipGraphLaunch(p.graph, stream_));
            return k;
        }
        launch_count_ = 0;
        run_sw_chain(m, T, logits_out != nullptr, p, -1, stream_);
        kernels_per_step_ = (launch_count_ + 1 + T - 1) / T;          // per chunk
    } else {
        general_prologue();
        const int tk = f16_tile_ ? 2 : tile_ok();
        const bool wavefront = lm_wavefront_on() && !profiling_ && T > lm_wavefront_min_chunks() &&
                               gemm_fullk(m, d.d_model, kz_hr(), true, 1, tk) && gemm_fullk(m, d.d_model, kz_ff2(), true, 1, tk);
        if (wavefront) {                 // long feed: all layers of a wavefront per launch (run_lm_wavefront)
            run_lm_wavefront(m, T, logits_out != nullptr);
            if (!logits_out) return k;
        } else if (use_graphs_ && !profiling_ && !logits_out) {
            const std::pair<int, int> key(m * 2 + flight_parity_, T);
            auto it = lm_graphs_.find(key);
            if (it == lm_graphs_.end()) {
                if (lm_graphs_.size() >= 32) { sync(); for (auto &g : lm_graphs_) (void)hipGraphExecDestroy(g.second); lm_graphs_.clear(); }
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                HipLegacyLock capture_guard_4;      // (no legacy-stream call of any thread during the capture: see engine.h)
                HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed));
                run_lm_chain(m, T, false);
                HIP_CHECK(hipStreamEndCapture(stream_, &graph));
                HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                HIP_CHECK(hipGraphDestroy(graph));
                it = lm_graphs_.emplace(key, exec).first;
            }
            HIP_CHECK(hipGraphLaunch(it->second, stream_));
            return k;
        } else {
            launch_count_ = 0;
            run_lm_chain(m, T, logits_out != nullptr);
            kernels_per_step_ = launch_count_ + 1;
        }
    }
    if (logits_out) {
        HIP_CHECK(hipMemcpyAsync(logits_h_, logits_, (size_t)3 * rows * d.vocab * 4, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipMemcpyAsync(rec_h_ + rec_off_h_[k], rec_d_ + rec_off_h_[k], (size_t)3 * rows * sizeof(StepRecord), hipMemcpyDeviceToHost, stream_));
        sync();
        memcpy(logits_out, logits_h_, (size_t)3 * rows * d.vocab * 4);
    }
    return k;
}


// Flights alternate between the two halves of the per-flight rings, so that the host can enqueue flight k + 1 while the GPU
// still runs flight k and the records of flight k are still being read (Scheduler::loop).  The device step counter starts at
// the half's first step index: the step tables are indexed by it, so the launch chains (graphs) do not depend on the parity.
// the flight's own copies of what the next flight's front end overwrites while this flight's layers / search still read it
void Engine::select_parity(int p)
{
    flight_parity_ = p;
    y_ = y_buf_[p]; ssq_ = ssq_buf_[p]; y16_ = y16_buf_[p]; step_d_ = step_buf_[p]; flags_d_ = flags_buf_[p]; rec_off_d_ = rec_off_buf_[p]; eout_lm_ = eout_lm_buf_[p];
}

void Engine::begin_flight()
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    const int p = flight_parity_ = next_parity_; next_parity_ ^= 1;
    // the buffers of this parity (index ring half, record ring half, y / ssq / step tables, eout_lm) were lent to the flight before the
    // previous one: it must have completed -- the scheduler keeps at most two flights open and completes them in order, so the event
    // is done long ago and this costs nothing; any other caller would silently overwrite an in-flight search
    if (flight_open_[p]) { HIP_CHECK(hipEventSynchronize(flight_done_[p])); flight_open_[p] = false; }
    ring_base_ = (size_t)p * ring_cap_; rec_base_ = (size_t)p * rec_cap_;
    ring_pos_ = ring_base_; rec_pos_ = rec_base_; flight_steps_ = 0;
    select_parity(p);
    flight_tail_s_ = false;
    { std::lock_guard<std::mutex> g(slot_mu_); zero_pending_slots(); }
}

bool Engine::flight_has_room(int rows, int nsteps) const
{
    // + max_slots: decoder refreshes stage their slot lists in the same index ring
    return flight_steps_ + nsteps <= step_cap_ && ring_pos_ + (size_t)3 * rows + (size_t)cfg_.max_slots <= ring_base_ + ring_cap_ && rec_pos_ + (size_t)3 * rows <= rec_base_ + rec_cap_;
}

int Engine::step(int m, const int *slots, const int *ring_tails, const int *now_ms, float *logits_out)
{
    if (m <= 0 || m > cfg_.max_batch || !flight_has_room(m, 1)) { LOGE("engine: step of %d rows does not fit (max batch %d)", m, cfg_.max_batch); abort(); }
    const int k = next_step_index();
    int *blk = ring_h_ + ring_pos_;
    memcpy(blk, slots, (size_t)m * 4); memcpy(blk + m, ring_tails, (size_t)m * 4); memcpy(blk + 2 * m, now_ms, (size_t)m * 4);
    step_off_h_[k] = (int)ring_pos_; rec_off_h_[k] = (int)rec_pos_;
    ring_pos_ += (size_t)3 * m; rec_pos_ += (size_t)3 * m;
    // The whole per-chunk chain (index fetch + 58 kernels) is replayed from a hipGraph captured once per batch size (and flight
    // parity: the parity selects buffers): at small batches the chain is launch-bound on the host (~3.5 us per launch), the
    // replay is not.  Kernel arguments depend only on m; the step's indices and its record offset reach the kernels through
    // the device step counter.
    std::lock_guard<std::mutex> cg(capture_mu_);               // aas_free on another thread resets slots through this stream
    general_prologue();
    const int gkey = m * 2 + flight_parity_;
    if (use_graphs_ && !profiling_ && !logits_out && (step_graphs_.count(gkey) || ++step_seen_[m] >= 2)) {      // (captured at the second use, see lm_step)
        auto it = step_graphs_.find(gkey);
        if (it == step_graphs_.end()) {
            if (step_graphs_.size() >= 256) { sync(); for (auto &g : step_graphs_) (void)hipGraphExecDestroy(g.second); step_graphs_.clear(); step_seen_.clear(); }
            const int par = flight_parity_;
            for (int k2 = 0; k2 < 2; ++k2) {               // both parities at once (see lm_step), this flight's last
                const int q = k2 == 0 ? 1 - par : par;
                if (step_graphs_.count(m * 2 + q)) continue;
                select_parity(q);
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                HipLegacyLock capture_guard_5;      // (no legacy-stream call of any thread during the capture: see engine.h)
                HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed));
                run_chain(m, false);
                HIP_CHECK(hipStreamEndCapture(stream_, &graph));
                HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                HIP_CHECK(hipGraphDestroy(graph));
                step_graphs_.emplace(m * 2 + q, exec);
            }
            select_parity(par);
            it = step_graphs_.find(gkey);
        }
        HIP_CHECK(hipGraphLaunch(it->second, stream_));
        return k;
    }
    launch_count_ = 0;
    run_chain(m, logits_out != nullptr);
    kernels_per_step_ = launch_count_ + 1;        // + the index fetch
    if (logits_out) {
        const NetDims &d = L_.dims;
        HIP_CHECK(hipMemcpyAsync(logits_h_, logits_, (size_t)3 * m * d.vocab * 4, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipMemcpyAsync(rec_h_ + rec_off_h_[k], rec_d_ + rec_off_h_[k], (size_t)3 * m * sizeof(StepRecord), hipMemcpyDeviceToHost, stream_));
        sync();
        memcpy(logits_out, logits_h_, (size_t)3 * m * d.vocab * 4);
    }
    return k;
}

void Engine::decode_rows(int n, const int *slots, int op)
{
    if (n <= 0) return;
    if (dec_table_ && op == 0) return;          // the joiner reads the table row of the slot's context: nothing to refresh
    HIP_CHECK(hipSetDevice(cfg_.device));
    const NetDims &d = L_.dims;
    const int MB = cfg_.max_batch;
    std::lock_guard<std::mutex> cg(capture_mu_);
    general_prologue();
    for (int o = 0; o < n; o += MB) {
        const int m = std::min(MB, n - o);
        if (ring_pos_ + (size_t)m > ring_base_ + ring_cap_) { LOGE("engine: index ring exhausted"); abort(); }
        int *blk = ring_h_ + ring_pos_;                  // staged in the flight's pinned ring: never rewritten before the copy ran
        memcpy(blk, slots + o, (size_t)m * 4);
        ring_pos_ += (size_t)m;
        HIP_CHECK(hipMemcpyAsync(dec_slots_d_, blk, (size_t)m * 4, hipMemcpyHostToDevice, stream_));
        DecRowsArgs a; a.slot_idx = dec_slots_d_; a.M = m; a.op = op; a.blank = P_.blank_id; a.state = gstate_; a.dec = dec_params(); a.de_out = dec_table_ ? nullptr : de_; a.ld_de = d.d_model;
        timed_begin(T_DEC); launch_dec_rows(a, stream_); timed_end(T_DEC);
        if (!dec_table_) run_decproj(m, dec_slots_d_, nullptr, nullptr, 1);
    }
}

int Engine::close_flight()
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    std::lock_guard<std::mutex> cg(capture_mu_);
    // where the flight ends: on S when its last step was a split feed (layers on M, search on S: the record copy must not sit
    // in M's order, where it would hold the next flight's layers back until this search is through), else on M
    hipStream_t tail = s_stream_;
    if (!(flight_tail_s_ && s_unseen_by_m_)) {
        tail = stream_;
        if (f_unseen_by_m_) { join(stream_, f_stream_); f_unseen_by_m_ = false; }
        if (s_unseen_by_m_) { join(stream_, s_stream_); s_unseen_by_m_ = false; }
    }
    if (rec_pos_ > rec_base_) HIP_CHECK(hipMemcpyAsync(rec_h_ + rec_base_, rec_d_ + rec_base_, (rec_pos_ - rec_base_) * sizeof(StepRecord), hipMemcpyDeviceToHost, tail));
    HIP_CHECK(hipEventRecord(flight_done_[flight_parity_], tail));
    flight_open_[flight_parity_] = true;
    return flight_parity_;
}

bool Engine::flight_done(int parity)
{
    const hipError_t e = hipEventQuery(flight_done_[parity]);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) HIP_CHECK(e);
    (void)hipGetLastError();
    return false;
}

void Engine::wait_flight(int parity)
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    HIP_CHECK(hipEventSynchronize(flight_done_[parity]));
    if (profiling_) sync();            // (profiled flights are completed one by one: every launch's event pair has run, collect them)
}

void Engine::end_flight() { wait_flight(close_flight()); }

// ---------------------------------------------------------------- debug / parity entry points
// The debug calls borrow slots 0..n-1; give them back reset (what a new session expects).
void Engine::zero_slots(int n)
{
    const NetDims &d = L_.dims;
    for (int i = 0; i < n; ++i) {
        ZeroSlotArgs z;
        z.h = h_; z.c = c_; z.n_layers = d.n_layers; z.slots = (size_t)cfg_.max_slots; z.d_model = d.d_model; z.hidden = d.hidden;
        z.eout = eout_; z.dout = dout_; z.joiner = d.joiner; z.state = gstate_; z.blank = P_.blank_id; z.slot = i; z.h16 = h16_;
        launch_zero_slot(z, stream_);
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::debug_encoder(int n, const float *x, const float *h, const float *c, float *eout, float *h2, float *c2)
{
    std::lock_guard<std::mutex> cg(capture_mu_);      // (lock order everywhere: capture_mu_, then the process-wide legacy-stream lock -- step() takes them in that order)
    HipLegacyLock legacy;
    HIP_CHECK(hipSetDevice(cfg_.device));
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    if (n > cfg_.max_batch || n > cfg_.max_slots) { LOGE("debug_encoder: n too large"); abort(); }
    // uses slots 0..n-1 directly (callers must not have live sessions); state layout is [n][L][*] on the host
    std::vector<int> slots((size_t)n);
    for (int i = 0; i < n; ++i) slots[(size_t)i] = i;
    float *xd = dmalloc<float>((size_t)n * d.seg * d.mel);
    HIP_CHECK(hipMemcpy(xd, x, (size_t)n * d.seg * d.mel * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < n; ++i)
        for (int l = 0; l < d.n_layers; ++l) {
            HIP_CHECK(hipMemcpy(h_ + ((size_t)l * S + i) * d.d_model, h + ((size_t)i * d.n_layers + l) * d.d_model, (size_t)d.d_model * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(c_ + ((size_t)l * S + i) * d.hidden, c + ((size_t)i * d.n_layers + l) * d.hidden, (size_t)d.hidden * 4, hipMemcpyHostToDevice));
        }
    HIP_CHECK(hipMemcpy(step_d_, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    if (f16_tile_)                              // the binary16 copy of the uploaded h rows (slots 0..n-1 of every layer)
        for (int l = 0; l < d.n_layers; ++l) launch_cvt_f16(h_ + (size_t)l * S * d.d_model, h16_ + (size_t)l * S * d.d_model, (size_t)n * d.d_model, stream_);
    run_encoder_rows(n, step_d_, step_d_, xd);
    sync();
    for (int i = 0; i < n; ++i) {
        HIP_CHECK(hipMemcpy(eout + (size_t)i * d.joiner, eout_ + (size_t)i * d.joiner, (size_t)d.joiner * 4, hipMemcpyDeviceToHost));
        for (int l = 0; l < d.n_layers; ++l) {
            HIP_CHECK(hipMemcpy(h2 + ((size_t)i * d.n_layers + l) * d.d_model, h_ + ((size_t)l * S + i) * d.d_model, (size_t)d.d_model * 4, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(c2 + ((size_t)i * d.n_layers + l) * d.hidden, c_ + ((size_t)l * S + i) * d.hidden, (size_t)d.hidden * 4, hipMemcpyDeviceToHost));
        }
    }
    (void)hipFree(xd);
    zero_slots(n);
}

void Engine::debug_decoder(int n, const int64_t *ctx, float *dout)
{
    std::lock_guard<std::mutex> cg(capture_mu_);      // (lock order everywhere: capture_mu_, then the process-wide legacy-stream lock -- step() takes them in that order)
    HipLegacyLock legacy;
    HIP_CHECK(hipSetDevice(cfg_.device));
    const NetDims &d = L_.dims;
    if (n > cfg_.max_batch) { LOGE("debug_decoder: n too large"); abort(); }
    if (dec_table_) {                           // what the sessions read: rows of the table (built by the kernels below, build_dec_table)
        bool in_range = true;
        for (int i = 0; i < n * 2; ++i) if (ctx[i] < 0 || ctx[i] >= d.vocab) in_range = false;
        if (in_range) {
            sync();
            for (int i = 0; i < n; ++i)
                HIP_CHECK(hipMemcpy(dout + (size_t)i * d.joiner, dec_table_ + ((size_t)ctx[2 * i] * d.vocab + (size_t)ctx[2 * i + 1]) * d.joiner, (size_t)d.joiner * 4, hipMemcpyDeviceToHost));
            return;
        }
    }
    std::vector<int> slots((size_t)n), c32((size_t)n * d.context);
    for (int i = 0; i < n; ++i) { slots[(size_t)i] = i; for (int t = 0; t < d.context; ++t) c32[(size_t)i * d.context + t] = (int)ctx[(size_t)i * d.context + t]; }
    int *ctx_d = dmalloc<int>(c32.size());
    HIP_CHECK(hipMemcpy(ctx_d, c32.data(), c32.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dec_slots_d_, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    DecEmbedArgs a; a.dec = dec_params(); a.ctx = ctx_d; a.M = n; a.out = de_; a.ldo = d.d_model;
    launch_dec_embed(a, stream_);
    run_decproj(n, dec_slots_d_, nullptr, nullptr, 1);
    sync();
    HIP_CHECK(hipMemcpy(dout, dout_, (size_t)n * d.joiner * 4, hipMemcpyDeviceToHost));
    (void)hipFree(ctx_d);
    zero_slots(n);
}

void Engine::debug_joiner(int n, const float *eout, const float *dout, float *logits)
{
    std::lock_guard<std::mutex> cg(capture_mu_);      // (lock order everywhere: capture_mu_, then the process-wide legacy-stream lock -- step() takes them in that order)
    HipLegacyLock legacy;
    const NetDims &d = L_.dims;
    HIP_CHECK(hipSetDevice(cfg_.device));
    if (n > cfg_.max_batch) { LOGE("debug_joiner: n too large"); abort(); }
    HIP_CHECK(hipMemcpy(eout_, eout, (size_t)n * d.joiner * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dout_, dout, (size_t)n * d.joiner * 4, hipMemcpyHostToDevice));
    std::vector<int> slots((size_t)n);
    for (int i = 0; i < n; ++i) slots[(size_t)i] = i;
    HIP_CHECK(hipMemcpy(step_d_, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    GemmArgs g; g.a0 = eout_; g.a0b = dout_; g.lda0 = d.joiner; g.aidx0 = step_d_; g.K0 = d.joiner; g.a_op = AOP_TANH_ADD;
    lin(g, L_.w_out); g.M = n; g.N = L_.vocab_pad; g.K = d.joiner; g.kz = kz_out_; g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
    launch_gemm(g, stream_);
    // logits = tree + bias for all padded columns; the host keeps the first `vocab` of each row
    float *lg = dmalloc<float>((size_t)n * L_.vocab_pad);
    RowArgs r; r.mode = ROW_SLOT_STORE; r.ws = ws_; r.parts = gemm_partials(n, L_.vocab_pad, kz_out_); r.m_stride = ws_mstride_; r.N = L_.vocab_pad; r.M = n;
    r.bias = w_ + L_.b_out; r.out = lg; r.ldo = L_.vocab_pad;
    launch_row(r, stream_);
    sync();
    std::vector<float> host((size_t)n * L_.vocab_pad);
    HIP_CHECK(hipMemcpy(host.data(), lg, host.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) memcpy(logits + (size_t)i * d.vocab, host.data() + (size_t)i * L_.vocab_pad, (size_t)d.vocab * 4);
    (void)hipFree(lg);
    zero_slots(n);
}

void Engine::debug_decide(int n, int op, const float *logits, float early_emit, const int *now_ms, int round, int32_t *state_io, StepRecord *rec_out)
{
    std::lock_guard<std::mutex> cg(capture_mu_);      // (lock order everywhere: capture_mu_, then the process-wide legacy-stream lock -- step() takes them in that order)
    HipLegacyLock legacy;
    const NetDims &d = L_.dims;
    HIP_CHECK(hipSetDevice(cfg_.device));
    if (n > cfg_.max_batch || n > cfg_.max_slots) { LOGE("debug_decide: n too large"); abort(); }
    static_assert(sizeof(GreedyState) == 16, "GreedyState is 4 x int32 in the ABI");
    std::vector<int> slots((size_t)n);
    for (int i = 0; i < n; ++i) slots[(size_t)i] = i;
    int *slots_d = dmalloc<int>((size_t)n), *now_d = dmalloc<int>((size_t)n), *active_d = dmalloc<int>((size_t)n), *dirty_d = dmalloc<int>((size_t)n);
    StepRecord *rec_d = dmalloc<StepRecord>((size_t)n);
    HIP_CHECK(hipMemcpy(slots_d, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(gstate_, state_io, (size_t)n * sizeof(GreedyState), hipMemcpyHostToDevice));
    if (op == 1) {
        DecRowsArgs a; a.slot_idx = slots_d; a.M = n; a.op = 1; a.blank = P_.blank_id; a.state = gstate_; a.dec = dec_params(); a.de_out = nullptr;
        launch_dec_rows(a, stream_);
    } else {
        // the logits enter as ONE partial plane with a zero bias: decide_kernel's tree sum + bias returns them unchanged
        std::vector<float> plane((size_t)n * L_.vocab_pad, -1000.0f);
        for (int i = 0; i < n; ++i) memcpy(plane.data() + (size_t)i * L_.vocab_pad, logits + (size_t)i * d.vocab, (size_t)d.vocab * 4);
        float *plane_d = dmalloc<float>(plane.size()), *zero_d = dmalloc<float>((size_t)L_.vocab_pad);
        HIP_CHECK(hipMemcpy(plane_d, plane.data(), plane.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(zero_d, 0, (size_t)L_.vocab_pad * 4));
        std::vector<int> gen((size_t)n, 1);
        HIP_CHECK(hipMemcpy(active_d, gen.data(), (size_t)n * 4, hipMemcpyHostToDevice));      // every row still searches
        HIP_CHECK(hipMemcpy(now_d, now_ms, (size_t)n * 4, hipMemcpyHostToDevice));
        DecideArgs a;
        a.ws = plane_d; a.parts = 1; a.m_stride = n; a.N = L_.vocab_pad; a.M = n; a.n_valid = d.vocab;
        a.bias = zero_d; a.blank = P_.blank_id; a.early_emit = early_emit;
        a.slot_idx = slots_d; a.now_ms = now_d; a.active = active_d; a.dirty = dirty_d; a.tok_class = cls_; a.state = gstate_;
        a.rec = rec_d; a.round = round; a.gen = 1; a.dec = dec_params(); a.de_out = nullptr;
        launch_decide(a, stream_);
        sync();
        HIP_CHECK(hipMemcpy(rec_out, rec_d, (size_t)n * sizeof(StepRecord), hipMemcpyDeviceToHost));
        (void)hipFree(plane_d); (void)hipFree(zero_d);
    }
    sync();
    HIP_CHECK(hipMemcpy(state_io, gstate_, (size_t)n * sizeof(GreedyState), hipMemcpyDeviceToHost));
    (void)hipFree(slots_d); (void)hipFree(now_d); (void)hipFree(active_d); (void)hipFree(dirty_d); (void)hipFree(rec_d);
    zero_slots(n);
}

void Engine::debug_fbank(int n_frames, const int16_t *pcm_frames, float *out)
{
    // every frame goes to slot 0, consecutive ring rows (n_frames <= ring_frames)
    const int padded = ft_.padded;
    std::vector<FbankFrameDesc> desc((size_t)n_frames);
    for (int i = 0; i < n_frames; ++i) { desc[(size_t)i].slot = 0; desc[(size_t)i].ring_row = i; desc[(size_t)i].pcm_off = i * padded; }
    std::pair<const int16_t *, size_t> part(pcm_frames, (size_t)n_frames * padded);
    fbank(n_frames, desc.data(), &part, 1, (size_t)n_frames * padded);      // (takes capture_mu_ itself: the legacy lock only afterwards, in the same order as step())
    std::lock_guard<std::mutex> cg(capture_mu_);
    HipLegacyLock legacy;
    sync();
    HIP_CHECK(hipMemcpy(out, ring_, (size_t)n_frames * ft_.nbins * 4, hipMemcpyDeviceToHost));
}

void Engine::read_greedy_state(int slot, GreedyState *out)
{
    HipLegacyLock legacy;
    HIP_CHECK(hipSetDevice(cfg_.device));
    sync_streams();
    HIP_CHECK(hipMemcpy(out, gstate_ + slot, sizeof(GreedyState), hipMemcpyDeviceToHost));
}

void Engine::read_ring(int slot, int row, int n_rows, float *out)
{
    HipLegacyLock legacy;
    HIP_CHECK(hipSetDevice(cfg_.device));
    sync_streams();
    const int nb = ft_.nbins;
    for (int i = 0; i < n_rows; ++i)
        HIP_CHECK(hipMemcpy(out + (size_t)i * nb, ring_ + ((size_t)slot * ring_frames_ + (row + i) % ring_frames_) * nb, (size_t)nb * 4, hipMemcpyDeviceToHost));
}

}  // namespace aprilx
