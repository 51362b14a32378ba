// Per-GPU engine: packed weights + slot-indexed persistent session state in HBM
// and the batched step functions that replace the three ORT Run() calls of the
// reference (src/april_session.c:131-179) and its per-session fbank
// (src/fbank.c:174-306).
//
// HBM layout (fp32, row-major):
//   h      [L][slots][d_model]      LSTM projected hidden state      (reference h tensor (L,1,d))
//   c      [L][slots][hidden]       LSTM cell state                  (reference c tensor (L,1,H))
//   ring   [slots][ring_frames][mel]  log-mel feature ring           (reference OnlineFBank ring)
//   eout   [slots][joiner]          projected encoder output of the last chunk
//   dout   [slots][joiner]          projected decoder output of the current token context
// plus per-step work buffers sized for `max_batch` rows.  State never leaves HBM
// between feed calls; per round only 12 bytes per session come back (JointResult).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include "fbank_tables.h"
#include "kernels.h"
#include "host_pool.h"
#include "model_loader.h"

namespace aprilx {

// Packed, device-layout weights as ONE contiguous blob (so they can be broadcast with a
// single collective and uploaded with a single copy).  Offsets are in floats.
struct PackedLayout {
    NetDims dims;
    size_t conv_w[3], conv_b[3];
    int k3 = 0;                              // K of the third conv as a GEMM (conv_ch[1]*9 rounded up to 64)
    size_t w_embed, b_embed;
    struct Layer { size_t wg, bg, whr, wff1, bff1, wff2, bff2; };
    std::vector<Layer> layers;
    size_t w_encproj, b_encproj, emb, dec_conv, dec_conv_b, w_decproj, b_decproj, w_out, b_out;
    size_t total = 0;
    int vocab_pad = 0;                       // joiner N padded to 16
    bool has_dec_conv_b = false;
    std::vector<float> norm_eps;             // [L]
    float embed_eps = 0;
};
void plan_layout(const NetDims &d, bool has_dec_conv_b, PackedLayout &L);
// fills `blob` (L.total floats) from the neutral host weights
void pack_weights(const HostModel &m, PackedLayout &L, std::vector<float> &blob);

struct EngineConfig {
    int device = 0;
    int max_slots = 4096;
    int max_batch = 2048;
    int precision = 0;                       // 0: fp32 GEMMs; 1: fp16 weights + fp16-rounded activations, fp32 accumulate (APRIL_PRECISION=f16)
};

struct KernelTiming { double ms = 0; long launches = 0; };

class Engine {
public:
    Engine(const EngineConfig &cfg, const PackedLayout &layout, const float *blob_host, const float *blob_device,
           const ModelParams &params, const FbankHostTables &ft);
    ~Engine();

    int device() const { return cfg_.device; }
    int max_batch() const { return cfg_.max_batch; }
    int max_slots() const { return cfg_.max_slots; }
    int ring_frames() const { return ring_frames_; }
    const NetDims &dims() const { return L_.dims; }
    hipStream_t stream() const { return stream_; }
    const float *weights_device() const { return w_; }
    int precision() const { return cfg_.precision; }
    const PackedLayout &layout() const { return L_; }

    int alloc_slot();                 // -1 when full; state zeroed (reference calloc, april_session.c:40-58)
    void free_slot(int slot);
    int live_slots() const { return live_; }

    // ---- batched hot path; host arrays are copied to pinned staging, all launches go to stream()
    // pcm arrives as `n_parts` windows that are gathered straight into pinned staging (total n_pcm samples)
    void fbank(int n_frames, const FbankFrameDesc *desc, const std::pair<const int16_t *, size_t> *parts, size_t n_parts, size_t n_pcm,
               HostPool *pool = nullptr);
    void encode(int n, const int *slots, const int *ring_tails);
    void decode(int n, const int *slots, const int *ctx /*[n][context]*/);
    // runs the joiner for n sessions, waits, returns results; logits_out optional [n][vocab] (host)
    void joint(int n, const int *slots, JointResult *out, float *logits_out);
    void sync();

    // ---- debug / parity entry points (state passed explicitly, like the ORT tensors)
    void debug_encoder(int n, const float *x, const float *h, const float *c, float *eout, float *h2, float *c2);
    void debug_decoder(int n, const int64_t *ctx, float *dout);
    void debug_joiner(int n, const float *eout, const float *dout, float *logits);
    void debug_fbank(int n_frames, const int16_t *pcm_frames /*[n][padded]*/, float *out /*[n][nbins]*/);
    void read_ring(int slot, int row, int n_rows, float *out);

    // ---- profiling: when enabled every launch of the named classes is bracketed by hipEvents
    void set_profiling(bool on);
    KernelTiming timing(int cls) const { return timing_[cls]; }
    void reset_timing();
    enum { T_GATES = 0, T_GEMM_OTHER = 1, T_ROW = 2, T_CONV = 3, T_FBANK = 4, T_DEC = 5, T_COUNT = 6 };

private:
    void upload_tables(const FbankHostTables &ft);
    void zero_slots(int n);
    void run_encoder_rows(int n, const int *d_slots, const int *d_tails, const float *x_direct);
    void timed_begin(int cls);
    void timed_end(int cls);
    void collect_timing();

    EngineConfig cfg_;
    PackedLayout L_;
    ModelParams P_;
    hipStream_t stream_ = nullptr;
    float *w_ = nullptr;                       // packed weights
    uint16_t *wh_ = nullptr;                   // fp16 copies of the GEMM weight sections (same offsets, in elements), precision 1 only
    void lin(GemmArgs &g, size_t off) const { if (wh_) { g.wp = wh_ + off; g.wt = 1; } else { g.wp = w_ + off; g.wt = 0; } }
    float *h_ = nullptr, *c_ = nullptr, *ring_ = nullptr, *eout_ = nullptr, *dout_ = nullptr;
    int ring_frames_ = 0;
    // work buffers
    float *xin_ = nullptr, *a3_ = nullptr, *xa_ = nullptr, *xb_ = nullptr, *u_ = nullptr, *ff_ = nullptr, *ws_ = nullptr, *de_ = nullptr;
    float *logits_ = nullptr;
    JointResult *joint_d_ = nullptr;
    // staging (pinned host + device mirrors), one region per call type
    int *hs_enc_ = nullptr, *ds_enc_ = nullptr;      // [2][max_batch]: slots, tails
    int *hs_dec_ = nullptr, *ds_dec_ = nullptr;      // [max_batch] slots + [max_batch*ctx] ctx
    int *hs_joi_ = nullptr, *ds_joi_ = nullptr;      // [max_batch]
    JointResult *joint_h_ = nullptr;
    float *logits_h_ = nullptr;
    // fbank staging is double-buffered so the next call can fill one pair while the previous copy is in flight
    FbankFrameDesc *hs_desc_[2] = {nullptr, nullptr}, *ds_desc_[2] = {nullptr, nullptr}; int desc_cap_ = 0;
    int16_t *hs_pcm_[2] = {nullptr, nullptr}, *ds_pcm_[2] = {nullptr, nullptr}; size_t pcm_cap_ = 0;
    int fb_flip_ = 0;
    hipEvent_t fb_done_[2] = {nullptr, nullptr};
    std::vector<size_t> part_off_;             // staging offsets of the PCM windows of one fbank call
    hipEvent_t dec_done_ = nullptr;            // last decoder launch has consumed its staged indices
    // fbank tables on device
    FbankTables ft_;
    std::vector<void *> table_allocs_;
    float pad_value_ = 0;
    // slots
    std::vector<int> free_;
    int live_ = 0;
    std::mutex slot_mu_;
    std::mutex capture_mu_;                    // held while stream_ is being captured into a graph, and by other threads' enqueues
    // gemm split factors (fixed per shape => batch-invariant numerics)
    int kz_embed_ = 1, kz_hr_ = 1, kz_ff2_ = 1, kz_proj_ = 1, kz_out_ = 1;
    int ws_mstride_ = 0;
    // encoder launch chains captured per batch size
    bool use_graphs_ = true;
    std::map<int, hipGraphExec_t> enc_graphs_;
    // profiling
    bool profiling_ = false;
    struct Ev { hipEvent_t a, b; int cls; };
    std::vector<Ev> ev_pool_; size_t ev_used_ = 0;
    KernelTiming timing_[T_COUNT];
};

}  // namespace aprilx
