// See engine.h.
#include "engine.h"
#include <algorithm>
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
    L.vocab_pad = (d.vocab + 15) & ~15;
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
template <typename T> static T *dmalloc(size_t n) { T *p = nullptr; HIP_CHECK(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T))); return p; }
template <typename T> static T *hmalloc(size_t n) { T *p = nullptr; HIP_CHECK(hipHostMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault)); return p; }

static int pick_kz(int K, int N)
{
    int kz = 256 / std::max(1, N / 16);
    kz = std::max(1, std::min(kz, 8));
    while (kz > 1 && ((K / 16) / kz < 4 || (K / 16) % (4 * kz) != 0)) kz >>= 1;   // every wave gets whole blocks of every slab
    return kz;
}

Engine::Engine(const EngineConfig &cfg, const PackedLayout &layout, const float *blob_host, const float *blob_device,
               const ModelParams &params, const FbankHostTables &ft)
    : cfg_(cfg), L_(layout), P_(params)
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    const NetDims &d = L_.dims;
    w_ = dmalloc<float>(L_.total);
    if (blob_device) HIP_CHECK(hipMemcpy(w_, blob_device, L_.total * 4, hipMemcpyDeviceToDevice));
    else HIP_CHECK(hipMemcpy(w_, blob_host, L_.total * 4, hipMemcpyHostToDevice));
    if (cfg_.precision == 1) {
        // fp16 operand mode (BASELINE configs[4]): every Linear / LSTM weight matrix gets an fp16 copy in the same
        // packed element order (round-to-nearest-even, on the device); convolutions, biases, embeddings stay fp32
        wh_ = dmalloc<uint16_t>(L_.total);
        const NetDims &d0 = L_.dims;
        auto cv = [&](size_t off, size_t n) { launch_cvt_f16(w_ + off, wh_ + off, n, nullptr); };
        cv(L_.w_embed, (size_t)d0.embed_in * d0.d_model);
        for (const PackedLayout::Layer &o : L_.layers) {
            cv(o.wg, (size_t)2 * d0.d_model * 4 * d0.hidden); cv(o.whr, (size_t)d0.hidden * d0.d_model);
            cv(o.wff1, (size_t)d0.d_model * d0.ffn); cv(o.wff2, (size_t)d0.ffn * d0.d_model);
        }
        cv(L_.w_encproj, (size_t)d0.d_model * d0.joiner); cv(L_.w_decproj, (size_t)d0.d_model * d0.joiner);
        cv(L_.w_out, (size_t)d0.joiner * L_.vocab_pad);
        HIP_CHECK(hipDeviceSynchronize());
    }

    const size_t S = (size_t)cfg_.max_slots, MB = (size_t)cfg_.max_batch;
    ring_frames_ = P_.segment_size * 32;                   // reference src/fbank.c:147
    h_ = dmalloc<float>((size_t)d.n_layers * S * d.d_model);
    c_ = dmalloc<float>((size_t)d.n_layers * S * d.hidden);
    ring_ = dmalloc<float>(S * ring_frames_ * d.mel);
    eout_ = dmalloc<float>(S * d.joiner);
    dout_ = dmalloc<float>(S * d.joiner);
    HIP_CHECK(hipMemset(h_, 0, (size_t)d.n_layers * S * d.d_model * 4));
    HIP_CHECK(hipMemset(c_, 0, (size_t)d.n_layers * S * d.hidden * 4));
    HIP_CHECK(hipMemset(ring_, 0, S * ring_frames_ * d.mel * 4));
    HIP_CHECK(hipMemset(eout_, 0, S * d.joiner * 4));
    HIP_CHECK(hipMemset(dout_, 0, S * d.joiner * 4));

    kz_embed_ = pick_kz(d.embed_in, d.d_model);
    kz_hr_ = pick_kz(d.hidden, d.d_model);
    kz_ff2_ = pick_kz(d.ffn, d.d_model);
    kz_proj_ = pick_kz(d.d_model, d.joiner);
    kz_out_ = pick_kz(d.joiner, L_.vocab_pad);
    ws_mstride_ = cfg_.max_batch;
    const size_t ws_n = (size_t)std::max({kz_embed_ * d.d_model, kz_hr_ * d.d_model, kz_ff2_ * d.d_model, kz_proj_ * d.joiner, kz_out_ * L_.vocab_pad});
    ws_ = dmalloc<float>(ws_n * MB);
    xin_ = dmalloc<float>(MB * d.embed_in);
    a3_ = dmalloc<float>(MB * d.f_out * L_.k3);
    HIP_CHECK(hipMemset(a3_, 0, MB * d.f_out * L_.k3 * 4));      // padded k columns (if any) stay zero
    xa_ = dmalloc<float>(MB * d.d_model);
    xb_ = dmalloc<float>(MB * d.d_model);
    u_ = dmalloc<float>(MB * d.hidden);
    ff_ = dmalloc<float>(MB * d.ffn);
    de_ = dmalloc<float>(MB * d.d_model);
    logits_ = dmalloc<float>(MB * d.vocab);
    joint_d_ = dmalloc<JointResult>(MB);
    hs_enc_ = hmalloc<int>(2 * MB); ds_enc_ = dmalloc<int>(2 * MB);
    hs_dec_ = hmalloc<int>(MB * (1 + d.context)); ds_dec_ = dmalloc<int>(MB * (1 + d.context));
    hs_joi_ = hmalloc<int>(MB); ds_joi_ = dmalloc<int>(MB);
    joint_h_ = hmalloc<JointResult>(MB);
    logits_h_ = hmalloc<float>(MB * d.vocab);

    upload_tables(ft);
    use_graphs_ = !(getenv("APRIL_NO_GRAPHS") && atoi(getenv("APRIL_NO_GRAPHS")));
    free_.reserve(S);
    for (int i = cfg_.max_slots - 1; i >= 0; --i) free_.push_back(i);
    LOGI("engine: device %d, %d slots, max batch %d, weights %.1f MB, kz(embed,hr,ff2,proj,out)=%d,%d,%d,%d,%d",
         cfg_.device, cfg_.max_slots, cfg_.max_batch, L_.total * 4.0 / 1e6, kz_embed_, kz_hr_, kz_ff2_, kz_proj_, kz_out_);
}

Engine::~Engine()
{
    (void)hipSetDevice(cfg_.device);
    (void)hipStreamSynchronize(stream_);
    for (auto &e : ev_pool_) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &g : enc_graphs_) (void)hipGraphExecDestroy(g.second);
    for (void *p : {(void *)w_, (void *)wh_, (void *)h_, (void *)c_, (void *)ring_, (void *)eout_, (void *)dout_, (void *)ws_, (void *)xin_, (void *)a3_, (void *)xa_,
                    (void *)xb_, (void *)u_, (void *)ff_, (void *)de_, (void *)logits_, (void *)joint_d_, (void *)ds_enc_, (void *)ds_dec_,
                    (void *)ds_joi_, (void *)ds_desc_[0], (void *)ds_desc_[1], (void *)ds_pcm_[0], (void *)ds_pcm_[1]})
        if (p) (void)hipFree(p);
    for (void *p : {(void *)hs_enc_, (void *)hs_dec_, (void *)hs_joi_, (void *)joint_h_, (void *)logits_h_, (void *)hs_desc_[0], (void *)hs_desc_[1], (void *)hs_pcm_[0], (void *)hs_pcm_[1]})
        if (p) (void)hipHostFree(p);
    for (int b = 0; b < 2; ++b) if (fb_done_[b]) (void)hipEventDestroy(fb_done_[b]);
    if (dec_done_) (void)hipEventDestroy(dec_done_);
    for (void *p : table_allocs_) (void)hipFree(p);
    (void)hipStreamDestroy(stream_);
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
    }
    ft_.padded = ft.padded; ft_.nbins = ft.nbins;
    pad_value_ = ft.pad_value;
}

int Engine::alloc_slot()
{
    std::lock_guard<std::mutex> g(slot_mu_);
    if (free_.empty()) return -1;
    const int s = free_.back();
    free_.pop_back();
    ++live_;
    return s;
}

void Engine::free_slot(int slot)
{
    // zero the slot's state so the next owner starts from the reference's calloc'd tensors (april_session.c:40-58).
    // Two strided 2-D memsets cover all layers; they are stream-ordered ahead of any later use of the slot.
    // Teardown-tolerant: sessions may be freed while the process is exiting and the HIP runtime is already gone.
    if (hipSetDevice(cfg_.device) == hipSuccess) {
        std::lock_guard<std::mutex> cg(capture_mu_);           // never enqueue into a stream that is being captured (encode())
        const NetDims &d = L_.dims;
        const size_t S = (size_t)cfg_.max_slots;
        (void)hipMemset2DAsync(h_ + (size_t)slot * d.d_model, S * d.d_model * 4, 0, (size_t)d.d_model * 4, (size_t)d.n_layers, stream_);
        (void)hipMemset2DAsync(c_ + (size_t)slot * d.hidden, S * d.hidden * 4, 0, (size_t)d.hidden * 4, (size_t)d.n_layers, stream_);
        (void)hipMemsetAsync(eout_ + (size_t)slot * d.joiner, 0, (size_t)d.joiner * 4, stream_);
        (void)hipMemsetAsync(dout_ + (size_t)slot * d.joiner, 0, (size_t)d.joiner * 4, stream_);
    }
    std::lock_guard<std::mutex> g(slot_mu_);
    free_.push_back(slot);
    --live_;
}

void Engine::sync() { HIP_CHECK(hipStreamSynchronize(stream_)); if (profiling_) collect_timing(); }

// ---------------------------------------------------------------- profiling
void Engine::set_profiling(bool on) { sync(); profiling_ = on; }
void Engine::reset_timing() { for (auto &t : timing_) t = KernelTiming(); }
void Engine::timed_begin(int cls)
{
    if (!profiling_) return;
    if (ev_used_ == ev_pool_.size()) { Ev e; HIP_CHECK(hipEventCreate(&e.a)); HIP_CHECK(hipEventCreate(&e.b)); e.cls = cls; ev_pool_.push_back(e); }
    ev_pool_[ev_used_].cls = cls;
    HIP_CHECK(hipEventRecord(ev_pool_[ev_used_].a, stream_));
}
void Engine::timed_end(int cls)
{
    if (!profiling_) return;
    (void)cls;
    HIP_CHECK(hipEventRecord(ev_pool_[ev_used_].b, stream_));
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
        for (int b = 0; b < 2; ++b) {
            if (hs_desc_[b]) { (void)hipHostFree(hs_desc_[b]); (void)hipFree(ds_desc_[b]); }
            if (hs_pcm_[b]) { (void)hipHostFree(hs_pcm_[b]); (void)hipFree(ds_pcm_[b]); }
        }
        desc_cap_ = std::max({n_frames * 2, desc_cap_, 1024});
        pcm_cap_ = std::max({n_pcm * 2, pcm_cap_, (size_t)1 << 16});
        for (int b = 0; b < 2; ++b) {
            hs_desc_[b] = hmalloc<FbankFrameDesc>((size_t)desc_cap_); ds_desc_[b] = dmalloc<FbankFrameDesc>((size_t)desc_cap_);
            hs_pcm_[b] = hmalloc<int16_t>(pcm_cap_); ds_pcm_[b] = dmalloc<int16_t>(pcm_cap_);
            if (!fb_done_[b]) HIP_CHECK(hipEventCreateWithFlags(&fb_done_[b], hipEventDisableTiming));
        }
    }
    const int b = fb_flip_;
    fb_flip_ ^= 1;
    HIP_CHECK(hipEventSynchronize(fb_done_[b]));          // the launch that used this pair two calls ago has consumed it
    memcpy(hs_desc_[b], desc, (size_t)n_frames * sizeof(FbankFrameDesc));
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
    HIP_CHECK(hipMemcpyAsync(ds_desc_[b], hs_desc_[b], (size_t)n_frames * sizeof(FbankFrameDesc), hipMemcpyHostToDevice, stream_));
    if (n_pcm) HIP_CHECK(hipMemcpyAsync(ds_pcm_[b], hs_pcm_[b], n_pcm * sizeof(int16_t), hipMemcpyHostToDevice, stream_));
    FbankArgs a;
    a.t = ft_; a.pcm = ds_pcm_[b]; a.desc = ds_desc_[b]; a.n_frames = n_frames; a.ring = ring_; a.ring_frames = ring_frames_; a.pad_value = pad_value_;
    timed_begin(T_FBANK);
    launch_fbank(a, stream_);
    timed_end(T_FBANK);
    HIP_CHECK(hipEventRecord(fb_done_[b], stream_));       // no host wait here: the encoder launches queue right behind
}

// ---------------------------------------------------------------- encoder
void Engine::run_encoder_rows(int n, const int *d_slots, const int *d_tails, const float *x_direct)
{
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    // conv front end
    ConvEmbedArgs ca;
    ca.ring = ring_; ca.ring_frames = ring_frames_; ca.mel = d.mel; ca.seg = d.seg;
    ca.slot_idx = d_slots; ca.ring_tail = d_tails; ca.x_direct = x_direct;
    for (int i = 0; i < 3; ++i) { ca.w[i] = w_ + L_.conv_w[i]; ca.b[i] = w_ + L_.conv_b[i]; ca.ch[i] = d.conv_ch[i]; ca.stride[i] = d.conv_stride[i]; }
    ca.ch1_per_group = (d.conv_ch[1] % 8 == 0) ? 8 : d.conv_ch[1];
    ca.out = a3_; ca.ldo = L_.k3; ca.M = n;
    timed_begin(T_CONV); launch_conv_embed(ca, stream_); timed_end(T_CONV);
    {   // third conv: [n*f_out, k3] x [k3, c2] + bias, DoubleSwish -> xin[n][f_out*c2]
        GemmArgs g; g.a0 = a3_; g.lda0 = L_.k3; g.K0 = L_.k3; g.wp = w_ + L_.conv_w[2];
        g.M = n * d.f_out; g.N = d.conv_ch[2]; g.K = L_.k3; g.kz = 1; g.epi = EPI_BIAS_DSWISH; g.out = xin_; g.ldo = d.conv_ch[2]; g.bias = w_ + L_.conv_b[2];
        timed_begin(T_CONV); launch_gemm(g, stream_); timed_end(T_CONV);
    }

    // embed linear + bias + BasicNorm
    {
        GemmArgs g; g.a0 = xin_; g.lda0 = d.embed_in; g.K0 = d.embed_in; lin(g, L_.w_embed);
        g.M = n; g.N = d.d_model; g.K = d.embed_in; g.kz = kz_embed_; g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
        timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
        RowArgs r; r.mode = ROW_NORM; r.ws = ws_; r.kz = gemm_partials(n, d.d_model, kz_embed_); r.m_stride = ws_mstride_; r.N = d.d_model; r.M = n;
        r.bias = w_ + L_.b_embed; r.out = xa_; r.ldo = d.d_model; r.eps = L_.embed_eps;
        timed_begin(T_ROW); launch_row(r, stream_); timed_end(T_ROW);
    }
    for (int l = 0; l < d.n_layers; ++l) {
        const PackedLayout::Layer &o = L_.layers[(size_t)l];
        float *h_l = h_ + (size_t)l * S * d.d_model;
        float *c_l = c_ + (size_t)l * S * d.hidden;
        {   // gates = [x | h_prev] x Wg ; fused LSTM cell
            GemmArgs g; g.a0 = xa_; g.lda0 = d.d_model; g.K0 = d.d_model;
            g.a1 = h_l; g.lda1 = d.d_model; g.aidx1 = d_slots; g.K1 = d.d_model;
            lin(g, o.wg); g.M = n; g.N = 4 * d.hidden; g.K = 2 * d.d_model; g.kz = 1; g.epi = EPI_LSTM;
            g.out = u_; g.ldo = d.hidden; g.bias = w_ + o.bg; g.c_state = c_l; g.slot_idx = d_slots; g.hidden = d.hidden;
            timed_begin(T_GATES); launch_gemm(g, stream_); timed_end(T_GATES);
        }
        {   // h' = u x Whr ; state write + residual
            GemmArgs g; g.a0 = u_; g.lda0 = d.hidden; g.K0 = d.hidden; lin(g, o.whr);
            g.M = n; g.N = d.d_model; g.K = d.hidden; g.kz = kz_hr_; g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
            timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
            RowArgs r; r.mode = ROW_HR; r.ws = ws_; r.kz = gemm_partials(n, d.d_model, kz_hr_); r.m_stride = ws_mstride_; r.N = d.d_model; r.M = n;
            r.resid = xa_; r.ldr = d.d_model; r.out = xb_; r.ldo = d.d_model; r.slot_idx = d_slots; r.state = h_l; r.ld_state = d.d_model;
            timed_begin(T_ROW); launch_row(r, stream_); timed_end(T_ROW);
        }
        {   // FFN up + DoubleSwish
            GemmArgs g; g.a0 = xb_; g.lda0 = d.d_model; g.K0 = d.d_model; lin(g, o.wff1);
            g.M = n; g.N = d.ffn; g.K = d.d_model; g.kz = 1; g.epi = EPI_BIAS_DSWISH; g.out = ff_; g.ldo = d.ffn; g.bias = w_ + o.bff1;
            timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
        }
        {   // FFN down + bias + residual + BasicNorm
            GemmArgs g; g.a0 = ff_; g.lda0 = d.ffn; g.K0 = d.ffn; lin(g, o.wff2);
            g.M = n; g.N = d.d_model; g.K = d.ffn; g.kz = kz_ff2_; g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
            timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
            RowArgs r; r.mode = ROW_NORM; r.ws = ws_; r.kz = gemm_partials(n, d.d_model, kz_ff2_); r.m_stride = ws_mstride_; r.N = d.d_model; r.M = n;
            r.bias = w_ + o.bff2; r.resid = xb_; r.ldr = d.d_model; r.out = xa_; r.ldo = d.d_model; r.eps = L_.norm_eps[(size_t)l];
            timed_begin(T_ROW); launch_row(r, stream_); timed_end(T_ROW);
        }
    }
    {   // encoder_proj -> eout[slot]
        GemmArgs g; g.a0 = xa_; g.lda0 = d.d_model; g.K0 = d.d_model; lin(g, L_.w_encproj);
        g.M = n; g.N = d.joiner; g.K = d.d_model; g.kz = kz_proj_; g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
        timed_begin(T_GEMM_OTHER); launch_gemm(g, stream_); timed_end(T_GEMM_OTHER);
        RowArgs r; r.mode = ROW_BIAS_STORE; r.ws = ws_; r.kz = gemm_partials(n, d.joiner, kz_proj_); r.m_stride = ws_mstride_; r.N = d.joiner; r.M = n;
        r.bias = w_ + L_.b_encproj; r.out = eout_; r.ldo = d.joiner; r.slot_idx = d_slots;
        timed_begin(T_ROW); launch_row(r, stream_); timed_end(T_ROW);
    }
}

void Engine::encode(int n, const int *slots, const int *ring_tails)
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    const int MB = cfg_.max_batch;
    for (int o = 0; o < n; o += MB) {
        const int m = std::min(MB, n - o);
        if (o > 0) sync();                               // staging region reuse
        memcpy(hs_enc_, slots + o, (size_t)m * 4);
        memcpy(hs_enc_ + MB, ring_tails + o, (size_t)m * 4);
        // The whole per-chunk chain (2 index uploads + ~90 kernels) is replayed from a hipGraph captured once per
        // batch size: at small batches the chain is launch-bound on the host (~3.5 us per launch), the replay is not.
        // Kernel arguments depend only on m; the slot/tail indices travel through the fixed pinned buffer.
        if (use_graphs_ && !profiling_) {
            auto it = enc_graphs_.find(m);
            if (it == enc_graphs_.end()) {
                if (enc_graphs_.size() >= 128) { for (auto &g : enc_graphs_) (void)hipGraphExecDestroy(g.second); enc_graphs_.clear(); }
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                std::lock_guard<std::mutex> cg(capture_mu_);       // aas_free on another thread zeroes slots through this stream
                HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
                HIP_CHECK(hipMemcpyAsync(ds_enc_, hs_enc_, (size_t)m * 4, hipMemcpyHostToDevice, stream_));
                HIP_CHECK(hipMemcpyAsync(ds_enc_ + MB, hs_enc_ + MB, (size_t)m * 4, hipMemcpyHostToDevice, stream_));
                run_encoder_rows(m, ds_enc_, ds_enc_ + MB, nullptr);
                HIP_CHECK(hipStreamEndCapture(stream_, &graph));
                HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                HIP_CHECK(hipGraphDestroy(graph));
                it = enc_graphs_.emplace(m, exec).first;
            }
            HIP_CHECK(hipGraphLaunch(it->second, stream_));
            continue;
        }
        HIP_CHECK(hipMemcpyAsync(ds_enc_, hs_enc_, (size_t)m * 4, hipMemcpyHostToDevice, stream_));
        HIP_CHECK(hipMemcpyAsync(ds_enc_ + MB, hs_enc_ + MB, (size_t)m * 4, hipMemcpyHostToDevice, stream_));
        run_encoder_rows(m, ds_enc_, ds_enc_ + MB, nullptr);
    }
}

// ---------------------------------------------------------------- decoder
void Engine::decode(int n, const int *slots, const int *ctx)
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    const NetDims &d = L_.dims;
    const int MB = cfg_.max_batch;
    for (int o = 0; o < n; o += MB) {
        const int m = std::min(MB, n - o);
        // two decoder launches can follow each other without a host wait in between (context reset after a flush, then
        // the first step of a new session): the pinned staging and the device index buffer may not be rewritten
        // before the previous launch has consumed them
        if (!dec_done_) HIP_CHECK(hipEventCreateWithFlags(&dec_done_, hipEventDisableTiming));
        else HIP_CHECK(hipEventSynchronize(dec_done_));
        memcpy(hs_dec_, slots + o, (size_t)m * 4);
        memcpy(hs_dec_ + MB, ctx + (size_t)o * d.context, (size_t)m * d.context * 4);
        HIP_CHECK(hipMemcpyAsync(ds_dec_, hs_dec_, (size_t)m * 4, hipMemcpyHostToDevice, stream_));
        HIP_CHECK(hipMemcpyAsync(ds_dec_ + MB, hs_dec_ + MB, (size_t)m * d.context * 4, hipMemcpyHostToDevice, stream_));
        DecEmbedArgs a; a.emb = w_ + L_.emb; a.conv_w = w_ + L_.dec_conv; a.conv_b = L_.has_dec_conv_b ? w_ + L_.dec_conv_b : nullptr;
        a.ctx = ds_dec_ + MB; a.d = d.d_model; a.groups = d.dec_groups; a.context = d.context; a.vocab = d.vocab; a.M = m; a.out = de_; a.ldo = d.d_model;
        timed_begin(T_DEC); launch_dec_embed(a, stream_); timed_end(T_DEC);
        GemmArgs g; g.a0 = de_; g.lda0 = d.d_model; g.K0 = d.d_model; lin(g, L_.w_decproj);
        g.M = m; g.N = d.joiner; g.K = d.d_model; g.kz = kz_proj_; g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
        timed_begin(T_DEC); launch_gemm(g, stream_); timed_end(T_DEC);
        RowArgs r; r.mode = ROW_BIAS_STORE; r.ws = ws_; r.kz = gemm_partials(m, d.joiner, kz_proj_); r.m_stride = ws_mstride_; r.N = d.joiner; r.M = m;
        r.bias = w_ + L_.b_decproj; r.out = dout_; r.ldo = d.joiner; r.slot_idx = ds_dec_;
        timed_begin(T_DEC); launch_row(r, stream_); timed_end(T_DEC);
        HIP_CHECK(hipEventRecord(dec_done_, stream_));
    }
}

// ---------------------------------------------------------------- joiner
void Engine::joint(int n, const int *slots, JointResult *out, float *logits_out)
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    const NetDims &d = L_.dims;
    const int MB = cfg_.max_batch;
    for (int o = 0; o < n; o += MB) {
        const int m = std::min(MB, n - o);
        memcpy(hs_joi_, slots + o, (size_t)m * 4);
        HIP_CHECK(hipMemcpyAsync(ds_joi_, hs_joi_, (size_t)m * 4, hipMemcpyHostToDevice, stream_));
        GemmArgs g; g.a0 = eout_; g.a0b = dout_; g.lda0 = d.joiner; g.aidx0 = ds_joi_; g.K0 = d.joiner; g.a_op = AOP_TANH_ADD;
        lin(g, L_.w_out); g.M = m; g.N = L_.vocab_pad; g.K = d.joiner; g.kz = kz_out_; g.epi = EPI_PARTIAL; g.out = ws_; g.m_stride = ws_mstride_;
        timed_begin(T_DEC); launch_gemm(g, stream_); timed_end(T_DEC);
        RowArgs r; r.mode = ROW_ARGMAX; r.ws = ws_; r.kz = gemm_partials(m, L_.vocab_pad, kz_out_); r.m_stride = ws_mstride_; r.N = L_.vocab_pad; r.M = m; r.n_valid = d.vocab;
        r.bias = w_ + L_.b_out; r.blank = P_.blank_id; r.joint = joint_d_; r.logits_dump = logits_out ? logits_ : nullptr;
        timed_begin(T_DEC); launch_row(r, stream_); timed_end(T_DEC);
        HIP_CHECK(hipMemcpyAsync(joint_h_, joint_d_, (size_t)m * sizeof(JointResult), hipMemcpyDeviceToHost, stream_));
        if (logits_out) HIP_CHECK(hipMemcpyAsync(logits_h_, logits_, (size_t)m * d.vocab * 4, hipMemcpyDeviceToHost, stream_));
        sync();
        memcpy(out + o, joint_h_, (size_t)m * sizeof(JointResult));
        if (logits_out) memcpy(logits_out + (size_t)o * d.vocab, logits_h_, (size_t)m * d.vocab * 4);
    }
}

// ---------------------------------------------------------------- debug / parity entry points
// The debug calls borrow slots 0..n-1; give them back zeroed (what a new session expects).
void Engine::zero_slots(int n)
{
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    for (int l = 0; l < d.n_layers; ++l) {
        HIP_CHECK(hipMemsetAsync(h_ + (size_t)l * S * d.d_model, 0, (size_t)n * d.d_model * 4, stream_));
        HIP_CHECK(hipMemsetAsync(c_ + (size_t)l * S * d.hidden, 0, (size_t)n * d.hidden * 4, stream_));
    }
    HIP_CHECK(hipMemsetAsync(eout_, 0, (size_t)n * d.joiner * 4, stream_));
    HIP_CHECK(hipMemsetAsync(dout_, 0, (size_t)n * d.joiner * 4, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::debug_encoder(int n, const float *x, const float *h, const float *c, float *eout, float *h2, float *c2)
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    const NetDims &d = L_.dims;
    const size_t S = (size_t)cfg_.max_slots;
    if (n > cfg_.max_batch || n > cfg_.max_slots) { LOGE("debug_encoder: n too large"); abort(); }
    // uses slots 0..n-1 directly (callers must not have live sessions); state layout is [n][L][*] on the host
    std::vector<int> slots((size_t)n), tails((size_t)n, 0);
    for (int i = 0; i < n; ++i) slots[(size_t)i] = i;
    float *xd = dmalloc<float>((size_t)n * d.seg * d.mel);
    HIP_CHECK(hipMemcpy(xd, x, (size_t)n * d.seg * d.mel * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < n; ++i)
        for (int l = 0; l < d.n_layers; ++l) {
            HIP_CHECK(hipMemcpy(h_ + ((size_t)l * S + i) * d.d_model, h + ((size_t)i * d.n_layers + l) * d.d_model, (size_t)d.d_model * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(c_ + ((size_t)l * S + i) * d.hidden, c + ((size_t)i * d.n_layers + l) * d.hidden, (size_t)d.hidden * 4, hipMemcpyHostToDevice));
        }
    memcpy(hs_enc_, slots.data(), (size_t)n * 4);
    HIP_CHECK(hipMemcpyAsync(ds_enc_, hs_enc_, (size_t)n * 4, hipMemcpyHostToDevice, stream_));
    run_encoder_rows(n, ds_enc_, ds_enc_, xd);
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
    const NetDims &d = L_.dims;
    std::vector<int> slots((size_t)n), c32((size_t)n * d.context);
    for (int i = 0; i < n; ++i) { slots[(size_t)i] = i; for (int t = 0; t < d.context; ++t) c32[(size_t)i * d.context + t] = (int)ctx[(size_t)i * d.context + t]; }
    decode(n, slots.data(), c32.data());
    sync();
    HIP_CHECK(hipMemcpy(dout, dout_, (size_t)n * d.joiner * 4, hipMemcpyDeviceToHost));
    zero_slots(n);
}

void Engine::debug_joiner(int n, const float *eout, const float *dout, float *logits)
{
    const NetDims &d = L_.dims;
    HIP_CHECK(hipSetDevice(cfg_.device));
    HIP_CHECK(hipMemcpy(eout_, eout, (size_t)n * d.joiner * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dout_, dout, (size_t)n * d.joiner * 4, hipMemcpyHostToDevice));
    std::vector<int> slots((size_t)n);
    for (int i = 0; i < n; ++i) slots[(size_t)i] = i;
    std::vector<JointResult> jr((size_t)n);
    joint(n, slots.data(), jr.data(), logits);
    zero_slots(n);
}

void Engine::debug_fbank(int n_frames, const int16_t *pcm_frames, float *out)
{
    // every frame goes to slot 0, consecutive ring rows (n_frames <= ring_frames)
    const int padded = ft_.padded;
    std::vector<FbankFrameDesc> desc((size_t)n_frames);
    for (int i = 0; i < n_frames; ++i) { desc[(size_t)i].slot = 0; desc[(size_t)i].ring_row = i; desc[(size_t)i].pcm_off = i * padded; }
    std::pair<const int16_t *, size_t> part(pcm_frames, (size_t)n_frames * padded);
    fbank(n_frames, desc.data(), &part, 1, (size_t)n_frames * padded);
    sync();
    HIP_CHECK(hipMemcpy(out, ring_, (size_t)n_frames * ft_.nbins * 4, hipMemcpyDeviceToHost));
}

void Engine::read_ring(int slot, int row, int n_rows, float *out)
{
    HIP_CHECK(hipSetDevice(cfg_.device));
    sync();
    const int nb = ft_.nbins;
    for (int i = 0; i < n_rows; ++i)
        HIP_CHECK(hipMemcpy(out + (size_t)i * nb, ring_ + ((size_t)slot * ring_frames_ + (row + i) % ring_frames_) * nb, (size_t)nb * 4, hipMemcpyDeviceToHost));
}

}  // namespace aprilx
