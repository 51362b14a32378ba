// .april container + PARAMS reader and structural weight extraction.
// Load-time half of the boundary: replaces reference src/file/model_file.c,
// src/params.c and the ORT session creation in src/april_model.c:24-107.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace aprilx {

struct ModelParams {                       // reference src/params.h:26-46
    int batch_size = 0, segment_size = 0, segment_step = 0, mel_features = 0, sample_rate = 0;
    int frame_shift_ms = 0, frame_length_ms = 0, round_pow2 = 0, mel_low = 0, mel_high = 0, snip_edges = 0;
    int token_count = 0, blank_id = 0;
    size_t token_stride = 0;               // longest token + 1
    std::vector<char> tokens;              // token_count x token_stride, NUL padded
    const char *token(size_t i) const { return tokens.data() + token_stride * i; }
};

struct NetDims {
    int n_layers = 0, d_model = 0, hidden = 0, ffn = 0, joiner = 0, vocab = 0;
    int mel = 0, seg = 0, context = 0, dec_groups = 0;
    int conv_ch[3] = {0, 0, 0};
    int conv_stride[3] = {1, 2, 2};
    int f_out = 0;                         // frequency bins after the conv stack
    int embed_in = 0;                      // conv_ch[2] * f_out
    int d_norm = 0;                        // pad_host_model: the file's d_model (the BasicNorm mean runs over it); 0 = d_model, nothing was padded
};

// Weights in a neutral host layout: every linear map is stored K x N row-major
// ("input-major"), i.e. out[n] = sum_k in[k] * W[k*N + n].
struct LayerWeights {
    std::vector<float> w_gates;            // [2*d_model][4*hidden]  rows: x part then h part; cols: gate-major i,f,g,o
    std::vector<float> b_gates;            // [4*hidden] = b_ih + b_hh
    std::vector<float> w_hr;               // [hidden][d_model]
    std::vector<float> w_ff1, b_ff1;       // [d_model][ffn], [ffn]
    std::vector<float> w_ff2, b_ff2;       // [ffn][d_model], [d_model]
    float norm_eps = 0;                    // exp(eps): added to mean(x^2)
};

struct HostModel {
    std::string language, name, description;
    ModelParams params;
    NetDims dims;
    std::vector<float> conv_w[3], conv_b[3];   // OIHW, [O]
    std::vector<float> w_embed, b_embed;       // [embed_in][d_model]
    float embed_norm_eps = 0;
    std::vector<LayerWeights> layers;
    std::vector<float> w_encproj, b_encproj;   // [d_model][joiner]
    std::vector<float> emb;                    // [vocab][d_model]
    std::vector<float> dec_conv;               // [d_model][d_model/groups][context]
    std::vector<float> dec_conv_b;             // optional [d_model]
    std::vector<float> w_decproj, b_decproj;   // [d_model][joiner]
    std::vector<float> w_out, b_out;           // [joiner][vocab]
    size_t param_count() const;
};

// Returns false and fills err on any validation failure (the C API then returns NULL,
// reference src/april_model.c:25-40,65-72,99-102).
bool load_april_file(const char *path, HostModel &out, std::string &err);

// Layer widths that are not multiples of 64 (the MFMA kernels' tile: 4 waves x 16 columns, 64-k stages), multiples of 16 or not: every width --
// d_model, cell, ffn, joiner, the third conv's channels -- is rounded up to the next multiple of 64 with zero weights and biases.
// Padded units compute exact zeros (LSTM cell: sigma(0) c + sigma(0) tanh(0); DoubleSwish(0) = 0; tanh(0) = 0 in the joiner) and feed
// zero rows of the next matrix, so the real outputs are the file's network; the one place the true width shows is the BasicNorm mean
// (dims.d_norm).  The decoder's grouped convolution grows by whole groups.  False (with err) when the padding is not a whole number
// of groups.  The reference runs any width through ONNXRuntime (src/april_session.c:131-180); nothing there corresponds to this.
bool pad_host_model(HostModel &m, std::string &err);

// Container-only parse (no network interpretation); used by tests of rejection cases.
struct ContainerInfo {
    std::string language, name, description;
    uint32_t model_type = 0;
    uint64_t params_off = 0, params_size = 0;
    std::vector<uint64_t> net_off, net_size;
    ModelParams params;
};
bool parse_container(const std::vector<uint8_t> &blob, ContainerInfo &info, std::string &err);
bool validate_params(const ModelParams &p, std::string &err);     // range checks of reference src/params.c:71-82 (+ frame shift >= 1 sample)

}  // namespace aprilx
