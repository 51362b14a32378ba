/*
 * ORACLE (test infrastructure, NOT product code) -- public declarations.
 *
 * A plain-C CPU restatement of april-asr's streaming hot path:
 *   PCM16 -> online fbank -> encoder / joiner / decoder graphs -> greedy
 *   transducer search -> result callbacks.
 * Every function cites the reference file:line it follows in its .c file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library; the product (april_asr_amd/) never links it.
 *
 * Parity status:
 *   fbank + container/params parser : PINNED against the reference's own C
 *       sources compiled into oracle/_ref/libaprilref.so.
 *   session state machine           : restated from src/april_session.c; the
 *       reference file cannot be compiled here (needs onnxruntime_c_api.h,
 *       which the image lacks) -> PARITY UNPINNED for this part.
 *   network arithmetic              : ONNX operator semantics executed over
 *       the graphs embedded in the .april file; the reference delegates this
 *       to ONNXRuntime 1.13.1 (absent) -> PARITY UNPINNED vs ORT.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- fbank ---------------- */
#define ORC_FFT_MAX 8192
typedef struct OrcRfftPlan {
    size_t n, nfct;
    size_t fct[16];
    double *tw[16];
    double *tws[16];      /* factors above 5 (generic pass): (cos, sin)(2 pi i / ip), i in [0, ip) */
    double *tw_store;
} OrcRfftPlan;

int  orc_rfft_plan_init(OrcRfftPlan *p, size_t n);
void orc_rfft_plan_free(OrcRfftPlan *p);
void orc_rfft_forward(const OrcRfftPlan *p, double *c, double *scratch);

typedef struct OrcFbank {
    int shift, padded, nfft_bins, nbins, seg_count, seg_step, shift_ms;
    OrcRfftPlan plan;
    float *window, *mel;
    float *ring; int ring_frames; int head, tail; size_t avail; long avail_shadow;
    float *fifo; size_t fifo_len; int fifo_cap;
    double *work;
    int dropped;
} OrcFbank;

void orc_make_window(float *out, int n);
void orc_make_melbank(float *tab, int nbins, int nfft_bins, int padded, int rate, int lo_hz, int hi_hz);
OrcFbank *orc_fbank_new(int rate, int shift_ms, int len_ms, int nbins, int round_pow2,
                        int mel_lo, int mel_hi, int seg_count, int seg_step);
void orc_fbank_free(OrcFbank *fb);
void orc_fbank_frame(const OrcFbank *fb, const float *frame, float *out);
void orc_fbank_accept(OrcFbank *fb, const float *wave, size_t count);
int  orc_fbank_flush(OrcFbank *fb);
int  orc_fbank_pull(OrcFbank *fb, float *out);
int  orc_fbank_stride_ms(const OrcFbank *fb);
const float *orc_fbank_window_ptr(const OrcFbank *fb);
const float *orc_fbank_mel_ptr(const OrcFbank *fb);
int  orc_fbank_padded(const OrcFbank *fb);

/* ---------------- .april container + PARAMS ---------------- */
typedef struct OrcParams {
    int batch_size, segment_size, segment_step, mel_features, sample_rate;
    int frame_shift_ms, frame_length_ms, round_pow2, mel_low, mel_high, snip_edges;
    int token_count, blank_id;
    size_t token_stride;   /* longest token + 1 */
    char *tokens;          /* token_count x token_stride, NUL padded */
} OrcParams;

typedef struct OrcFile {
    char language[9];
    char *name, *description;
    uint32_t model_type;
    uint64_t params_off, params_size;
    uint64_t n_networks;
    uint64_t net_off[8], net_size[8];
    uint8_t *blob; size_t blob_size;    /* whole file */
    OrcParams params;
} OrcFile;

OrcFile *orc_file_open(const char *path);   /* NULL on any validation failure */
void     orc_file_free(OrcFile *f);
const char *orc_token(const OrcParams *p, size_t idx);

/* ---------------- mini ONNX interpreter ---------------- */
typedef struct OrcGraph OrcGraph;
OrcGraph *orc_graph_parse(const uint8_t *bytes, size_t n);
void      orc_graph_free(OrcGraph *g);
int  orc_graph_num_inputs(const OrcGraph *g);
int  orc_graph_num_outputs(const OrcGraph *g);
/* dims of graph input/output i; returns rank */
int  orc_graph_input_dims(const OrcGraph *g, int i, int64_t *dims, int max);
int  orc_graph_output_dims(const OrcGraph *g, int i, int64_t *dims, int max);
const char *orc_graph_input_name(const OrcGraph *g, int i);
const char *orc_graph_output_name(const OrcGraph *g, int i);
/* Run with inputs/outputs bound by NAME to caller buffers (float32 or int64
   according to the graph's declared types).  Returns 0 on success. */
int  orc_graph_run(OrcGraph *g, int n_in, const char *const *in_names, const void *const *in_bufs,
                   int n_out, const char *const *out_names, void *const *out_bufs);
const char *orc_graph_last_error(void);
/* checker mode for the product's fp16 path: MatMul/Gemm operands rounded to binary16, fp32 accumulate (process-wide) */
void orc_set_f16_linear(int on);
int  orc_get_f16_linear(void);
void orc_round_f16(const float *src, float *dst, size_t n);   /* the rounding it uses (nearest binary16, ties to even) */

/* ---------------- model + session ---------------- */
typedef struct OrcModel {
    OrcFile *file;
    OrcGraph *enc, *dec, *joi;
    int64_t x_dim[3], h_dim[3], c_dim[3], eout_dim[3], dout_dim[3], ctx_dim[2], logits_dim[3];
} OrcModel;

OrcModel *orc_model_load(const char *path);
void      orc_model_free(OrcModel *m);

typedef struct OrcToken {
    int32_t id;        /* token index into the model's table */
    float   logprob;
    int32_t flags;     /* bit0 word boundary, bit1 sentence end */
    uint64_t time_ms;
} OrcToken;

/* result types use the reference's numbering: 1 partial, 2 final, 3 cant-keep-up, 4 silence */
typedef void (*OrcHandler)(void *ud, int type, size_t count, const OrcToken *tokens);

/* Network provider hooks: default = the ONNX graphs; tests may script them. */
typedef struct OrcNets {
    void *ud;
    void (*encoder)(void *ud, const float *x, const float *h, const float *c,
                    float *eout, float *h2, float *c2);
    void (*decoder)(void *ud, const int64_t *ctx, float *dout);
    void (*joiner)(void *ud, const float *eout, const float *dout, float *logits);
} OrcNets;

typedef struct OrcSession OrcSession;
OrcSession *orc_session_new(OrcModel *m, OrcHandler h, void *ud);
/* state machine only: dims given explicitly, nets scripted */
OrcSession *orc_session_new_scripted(const OrcParams *p, const OrcNets *nets, int n_layers_h, int h_elems,
                                     int c_elems, int e_elems, int vocab, OrcHandler h, void *ud);
void orc_session_free(OrcSession *s);
void orc_session_feed_pcm16(OrcSession *s, const int16_t *pcm, size_t n);
void orc_session_flush(OrcSession *s);
/* optional tracing: every joiner call appends vocab floats here when set */
void orc_session_set_logit_trace(OrcSession *s, float *buf, size_t cap_floats, size_t *used_floats);
/* optional tracing of every pulled chunk x (seg_count*nbins floats each) */
void orc_session_set_chunk_trace(OrcSession *s, float *buf, size_t cap_floats, size_t *used_floats);
uint64_t orc_session_chunks(const OrcSession *s);

#ifdef __cplusplus
}
#endif
#endif
