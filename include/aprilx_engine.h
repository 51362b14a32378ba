/*
 * aprilx_engine.h -- engine-level C ABI of libaprilasr.so (MI355X build).
 *
 * The reference runs its networks through ONNXRuntime's C API behind
 * src/ort_util.{h,c}: one g_ort->Run per graph per session per chunk, batch 1
 * (src/april_session.c:145,160,176).  This header is the batched replacement of
 * that inner boundary plus the few knobs a multi-session / multi-GPU host needs.
 * Plain pointers and sizes only; every function cites what it replaces.
 *
 * All functions are safe to call from any thread; sessions must not be fed from
 * two threads at once (same rule as the reference).
 */
#ifndef APRILX_ENGINE_H
#define APRILX_ENGINE_H
#include "april_api.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Dimensions read from the model's graphs (reference src/april_model.h:35-41). */
typedef struct AprilxDims {
    int32_t n_layers, d_model, hidden, ffn, joiner, vocab, mel, seg, seg_step, context;
    int32_t fft_size, frame_shift, sample_rate, blank_id, n_devices;
    int32_t precision;          /* 0: fp32 GEMMs (default); 1: fp16 operands, fp32 accumulate (env APRIL_PRECISION=f16) */
    int32_t d_model_file;       /* 0, or the model file's d_model when its layer widths (any that are not multiples of 64) were rounded up to multiples of
                                   64 at load: d_model, hidden, ffn, joiner above are then the padded widths the engine runs */
    int64_t param_count;
} AprilxDims;
APRIL_EXPORT int aprilx_model_dims(AprilASRModel model, AprilxDims *out);
/* token text by id (reference src/params.c:31-33 get_token) */
APRIL_EXPORT const char *aprilx_model_token(AprilASRModel model, int32_t id);

/* ---- weight distribution (multi-GPU) -------------------------------------------------------
 * Sessions are independent, so a node's session pool is partitioned across its GPUs and the only
 * collective is the broadcast of the packed weights at model load (reference load site
 * src/april_model.c:57-61; the reference has no multi-device path).  The library does it with RCCL:
 *   - one process, several GPUs: aam_create_model with APRIL_GPU_DEVICES=0,1,... uploads the weights
 *     once and broadcasts them to the other devices (ncclCommInitAll + grouped ncclBroadcast);
 *   - one process per GPU: rank 0 parses the .april file (aam_create_model), calls
 *     aprilx_broadcast_get_id and hands the 128 bytes to the other ranks by any means (a launcher's
 *     store, a file, torch.distributed); then EVERY rank calls aprilx_model_broadcast.  Rank 0 passes
 *     its model and gets it back; the others pass NULL and receive a model built from the metadata and
 *     the weights that arrive in their GPU's memory over xGMI -- no file access, no host staging.
 * The blob functions below carry the same content through host memory (machines without RCCL peers,
 * the gloo test path, the on-disk cache).                                                      */
typedef struct AprilxLoadInfo {
    double broadcast_ms;        /* wall time of the weight broadcast (0 if none happened) */
    double comm_init_ms;        /* wall time of the RCCL communicator set-up */
    uint64_t broadcast_bytes;
    int32_t ranks;              /* devices / processes the weights were broadcast to, including the root */
    int32_t used_rccl;
} AprilxLoadInfo;
/* writes an RCCL unique id (128 bytes) to id_out; returns its size, -1 on failure */
APRIL_EXPORT int aprilx_broadcast_get_id(void *id_out, size_t cap);
APRIL_EXPORT AprilASRModel aprilx_model_broadcast(AprilASRModel root_model, int rank, int world, const void *id_bytes);
APRIL_EXPORT int aprilx_model_load_info(AprilASRModel model, AprilxLoadInfo *out);
APRIL_EXPORT size_t aprilx_model_blob_size(AprilASRModel model);
APRIL_EXPORT int aprilx_model_export_blob(AprilASRModel model, void *dst, size_t dst_size);
/* `blob` may be a host pointer or a device pointer on the calling process's GPU */
APRIL_EXPORT AprilASRModel aprilx_model_from_blob(const void *blob, size_t size, int blob_is_device_ptr);
/* The same blob as a file next to the model: a cache of the parsed + MFMA-packed weights, so that later loads skip the
 * ONNX parse and the packing (the reference re-parses the .april file at every
 * aam_create_model, april_model.c:24-107).  save returns 0 on success; load returns NULL on any failure. */
APRIL_EXPORT int aprilx_model_save_blob(AprilASRModel model, const char *path);
/* fp16 variant of the cache file for fp16-operand mode (BASELINE configs[4]): the MFMA-packed matrices as binary16, half the
 * size; aprilx_model_load_blob recognises it and requires APRIL_PRECISION=f16 on a GPU runtime */
APRIL_EXPORT int aprilx_model_save_blob_f16(AprilASRModel model, const char *path);
APRIL_EXPORT AprilASRModel aprilx_model_load_blob(const char *path);

/* ---- batched session driving ------------------------------------------------------------
 * aas_feed_pcm16 / aas_flush (reference april_api.h:183,186) for n sessions in ONE call, so
 * that all of them advance in the same GPU steps.  Blocks until the work is done; handlers
 * of synchronous sessions run on the calling thread before it returns.                    */
APRIL_EXPORT void aprilx_feed_many(size_t n, AprilASRSession *sessions, const short *const *pcm16, const size_t *short_counts);
APRIL_EXPORT void aprilx_flush_many(size_t n, AprilASRSession *sessions);
/* block until an asynchronous session has consumed everything queued so far */
APRIL_EXPORT void aprilx_session_drain(AprilASRSession session);
/* Pipelined group feed: queue one feed for each of the n sessions (the samples are COPIED, as for an asynchronous session,
 * reference src/april_session.c:479-500), then block only until every session has at most `depth - 1` feeds that are
 * queued or in progress.  depth = 2 is double buffering: the call for feed k + 1 returns when feed k is complete, so the
 * library prepares and launches feed k + 1 while the GPU still works on feed k.  depth = 1 waits for this very feed
 * (aprilx_feed_many without lending the buffers).  Results arrive in feed order: handlers of synchronous sessions run on the
 * calling thread before the call returns (for the feeds that have completed by then), those of asynchronous sessions on the
 * library thread.  aprilx_drain_many waits for everything queued and delivers what is left.                              */
APRIL_EXPORT void aprilx_feed_many_pipelined(size_t n, AprilASRSession *sessions, const short *const *pcm16, const size_t *short_counts, int depth);
APRIL_EXPORT void aprilx_drain_many(size_t n, AprilASRSession *sessions);

/* ---- direct network evaluation (parity tests) -------------------------------------------
 * Same tensors as the three ORT Run calls, with a leading batch of n independent sessions:
 *   encoder: x[n][seg][mel], h[n][L][d_model], c[n][L][hidden] -> eout[n][joiner], h2, c2
 *            (reference src/april_session.c:131-148)
 *   decoder: context[n][2] int64 -> dout[n][joiner]              (:151-163)
 *   joiner : eout[n][joiner], dout[n][joiner] -> logits[n][vocab] (:166-179)
 *   fbank  : n frames of fft_size PCM16 samples -> n rows of `mel` log energies
 *            (reference src/fbank.c:228-296 per frame)
 * These use state slots 0..n-1 directly and must not be mixed with live sessions.        */
APRIL_EXPORT int aprilx_run_encoder(AprilASRModel model, int n, const float *x, const float *h, const float *c,
                                    float *eout, float *h2, float *c2);
APRIL_EXPORT int aprilx_run_decoder(AprilASRModel model, int n, const int64_t *context, float *dout);
APRIL_EXPORT int aprilx_run_joiner(AprilASRModel model, int n, const float *eout, const float *dout, float *logits);
APRIL_EXPORT int aprilx_run_fbank(AprilASRModel model, int n_frames, const int16_t *pcm_frames, float *out);
/* The device's copy of the search decision (reference src/april_session.c:306-429, the part the next network call depends
 * on) in isolation: op 0 = one joiner round for n rows with GIVEN logits[n][vocab], session times now_ms[n] and search states
 * state_io[n][4] = {context[0], context[1], last active token or -1, time of the last emission in ms}; writes the 16-byte
 * records {idx, max, blank logit, flags: 1 valid | 2 blank | 4 context changed} to records_out[n][4 x 32 bit] and the new states
 * to state_io.  op 1 = the end-of-flush reset (:561-563: tokens forgotten, context cleared unless it starts with blank).
 * Tests only (uses slots 0..n-1; must not be mixed with live sessions). */
APRIL_EXPORT int aprilx_run_decide(AprilASRModel model, int n, int op, const float *logits, float early_emit, const int32_t *now_ms,
                                   int round, int32_t *state_io, void *records_out);

/* The host-side plan of a row-epilogue GEMM out[M, N] = A[M, K] x W (K in `kz` slabs; `zcount` same-shape problems per launch;
 * tile_ok 0 = round-2 schedules, 1 = GM_TILE by the occupancy rule, 2 = GM_TILE always (fp16 tile path); force = the caller
 * needs the fused form): out[0] = 1 when the plan keeps all of K in the workgroup (row epilogue fused into the GEMM), out[1] =
 * partial planes the split form writes (finished by the row kernel), out[2] = 1 when the occupancy rule picks GM_TILE.  No GPU
 * needed; tests only (the engine and the kernels consult the same functions, so their decisions cannot diverge). */
APRIL_EXPORT int aprilx_plan_gemm(int M, int N, int kz, int zcount, int tile_ok, int force, int32_t *out);

/* Which weight-stream kernel (csrc/kernels_recur.hip; layer GEMMs at <= 16 rows) takes a layer GEMM of this shape: kind 0 = the
 * one-launch gates GEMM of a chunk step, 1 = its recurrent half (long feeds), 2 = its input half (long feeds), 3 = FFN up,
 * 4 = LSTM projection, 5 = FFN down; K in `kz` slabs, `groups` sum-of-squares partials per row.  Returns 0 when the general GEMM
 * kernels run it, else the form number (3, 1, 4, 5, 2, 6 for the six kinds), -1 on bad arguments.  No GPU needed; tests only. */
APRIL_EXPORT int aprilx_stream_form(int kind, int M, int N, int K, int kz, int groups);

/* ---- tracing / statistics ---------------------------------------------------------------*/
/* every joiner evaluation of this session appends `vocab` floats to buf (tests only; chunk steps of a traced session are
   issued eagerly and waited for one by one) */
APRIL_EXPORT void aprilx_session_trace_logits(AprilASRSession session, float *buf, size_t cap_floats, size_t *used_floats);
APRIL_EXPORT uint64_t aprilx_session_chunks(AprilASRSession session);
/* parity tests of the online fbank (reference src/fbank.c:174-349): copies log-mel rows [first, first + n) of everything the
 * session's feature ring has received so far -- real frames and flush padding, in the order the reference's ring sees them --
 * into out[n][mel] and returns the number of rows written so far.  Nothing is copied when the range is not available: rows that have
 * not been written yet (first + n > rows written) return the count as usual -- the caller sees count < first + n -- and rows that
 * have already left the ring (more than ring_frames rows written since `first`) return UINT64_MAX.  n = 0 / out = NULL: only the
 * count.  Waits for the session to be idle.  Chunk j of the session is rows [j * segment_step, j * segment_step + segment_size).  */
APRIL_EXPORT uint64_t aprilx_session_read_frames(AprilASRSession session, uint64_t first, int n, float *out);
/* The token context as the host's result state machine holds it (host_ctx[2]) and the search state the device keeps for the
   session's slot (device_state[4]: context[0], context[1], last active token or -1, time of the last emission in ms).  The two
   contexts are derived independently from the same joiner results (reference context tensor, src/april_session.c:181-196)
   and must agree; tests only. */
APRIL_EXPORT void aprilx_session_context(AprilASRSession session, int32_t *host_ctx, int32_t *device_state);

typedef struct AprilxStats {
    uint64_t ticks, steps, chunks, rounds, frames, max_batch_seen;
    /* per kernel class: accumulated ms and launch counts while profiling is enabled
       0 gates GEMM+LSTM cell, 1 other encoder GEMMs, 2 row epilogues, 3 conv front end, 4 fbank, 5 decoder+joiner */
    double kernel_ms[6];
    uint64_t kernel_launches[6];
    /* host wall time of the GPU's stepping thread by phase (ms): 0 collect work, 1 frame bookkeeping, 2 fbank call,
       3 chunk-step enqueue, 4 end of flight (the one wait for the GPU), 5 replay of the device's per-round records through
       the result state machine, 6 decoder refresh enqueue, 7 completion */
    double host_ms[8];
    uint64_t flights;            /* host waits for the GPU (one per flight = per batch of queued chunk steps) */
    uint64_t replay_mismatch;    /* rounds where the host state machine and the device decision disagreed (must stay 0) */
    uint64_t kernels_per_step;   /* launches of the last eagerly issued chunk chain (profiling / APRIL_NO_GRAPHS runs) */
    uint64_t lm_steps, lm_chunks;/* layer-major steps (long feeds) and the session-chunks they covered (included in steps / chunks) */
    uint64_t wave_steps, wave_chunks;/* feeds whose 2..7 chunk steps ran as one wavefront over the layers, and the session-chunks they covered (included in steps / chunks) */
    /* the gates clock (aprilx_model_profile(model, 2) ... (model, 0)): every gates launch of the feed wavefronts timed ITSELF under graph
       replay (first workgroup's start to last workgroup's end, s_memrealtime) -- accumulated ms, launches and rows (sessions x layers
       sharing the launch) of the last clocked interval */
    double gates_clock_ms; uint64_t gates_clock_launches, gates_clock_rows;
    double gates_clock_ms_by_n[4]; uint64_t gates_clock_launches_by_n[4];   /* the same, split by the problems sharing the launch (1, 2, 3, >= 4) */
} AprilxStats;
APRIL_EXPORT void aprilx_model_stats(AprilASRModel model, int device_index, AprilxStats *out);
/* Hand-over -> delivery latency of the last (up to 8192) completed ticks of one GPU's stepping thread, in ms, oldest first: from the
 * feed call (aas_feed_pcm16 / aprilx_feed_many / aprilx_feed_many_pipelined / flush) that queued the oldest work a flight served to
 * the moment that flight's results were delivered (asynchronous and pipelined sessions: their handlers have run; synchronous callers:
 * released).  With the pipelined group feed the duration of the feed CALL is only the hand-over; this is the latency a client sees
 * (reference: the time aas_feed_pcm16 blocks, src/april_session.c:479-538).  out_ms = NULL: returns the number available; reset != 0
 * empties the ring afterwards.                                                                                                    */
APRIL_EXPORT int aprilx_model_feed_latency(AprilASRModel model, int device_index, double *out_ms, int cap, int reset);
/* enable 1: launches go out one by one with hipEvents around them on the engine's stream (gates launches: their own dispatch time
   stamps), per-class times in AprilxStats.kernel_ms -- measurement runs only.  enable 2: the gates clock -- nothing changes in how feeds
   run (graphs replay, flights overlap) except that the feed wavefronts' gates kernels stamp their own start and end; switching back to 0
   publishes AprilxStats.gates_clock_*.  0: off. */
APRIL_EXPORT void aprilx_model_profile(AprilASRModel model, int enable);

/* state machine alone, for host-logic tests: feed (idx, max, blank) triples, receive events */
typedef struct AprilxGreedy_i *AprilxGreedy;
APRIL_EXPORT AprilxGreedy aprilx_greedy_create(AprilASRModel model, AprilRecognitionResultHandler handler, void *userdata);
/* returns 1 when the round resolved to blank; ctx_out receives the 2-token context */
APRIL_EXPORT int aprilx_greedy_step(AprilxGreedy g, int32_t idx, float max_val, float blank_val, float early_emit,
                                    size_t now_ms, int32_t *ctx_out);
APRIL_EXPORT void aprilx_greedy_finish(AprilxGreedy g);
APRIL_EXPORT void aprilx_greedy_free(AprilxGreedy g);

/* A result handler implemented in C, for load generators and benchmarks (a Python or JNI callback costs more than
   the GPU step at thousands of sessions).  userdata -> uint64_t[6]: calls, partial, final, cant_keep_up, silence, tokens */
APRIL_EXPORT void aprilx_counting_handler(void *userdata, AprilResultType type, size_t count, const AprilToken *tokens);

/* parse + weight extraction + packing without creating any GPU object (loader tests; no sessions) */
APRIL_EXPORT AprilASRModel aprilx_model_load_host(const char *model_path);
/* fbank tables as the device sees them (window[fft_size], mel[mel][fft_size/2]); returns fft_size */
APRIL_EXPORT int aprilx_model_fbank_tables(AprilASRModel model, float *window, float *mel);

/* container-only parse (no GPU): 0 on success, otherwise -1 and a message in err */
APRIL_EXPORT int aprilx_probe_file(const char *path, char *err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif
