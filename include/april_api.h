/*
 * april_api.h -- public C ABI of libaprilasr.so (MI355X-native build).
 *
 * This header is written for the MI355X engine; it declares the SAME eleven
 * entry points, struct layouts and enum values as the reference library so that
 * ./main-style programs and the Python / C# / Java bindings bind unchanged.
 * Each declaration cites the reference interface it replaces
 * (/root/reference/april_api.h:LINE).  Layouts (LP64): AprilToken 32 bytes
 * {0,8,12,16,24}, AprilConfig 40 bytes {0,16,24,32}; enums are 4 bytes.
 * tests/test_abi.py pins those numbers.
 *
 * Behavioural contract (reference src/april_session.c):
 *   - flags == 0  : synchronous session. aas_feed_pcm16 / aas_flush return after
 *                   all work is done; the handler runs on the calling thread.
 *   - ASYNC_RT(1) / ASYNC_NO_RT(2): feed copies into a bounded per-session ring
 *                   and returns; the handler runs on a library thread;
 *                   CANT_KEEP_UP is delivered on the calling thread on overflow.
 *                   The GPU engine never time-compresses audio, so ASYNC_RT
 *                   behaves like ASYNC_NO_RT and aas_realtime_get_speedup()
 *                   reports 1.0.
 */
#ifndef APRIL_API_MI355X_H
#define APRIL_API_MI355X_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define APRIL_EXPORT __attribute__((visibility("default")))
#else
#define APRIL_EXPORT
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define APRIL_VERSION 1                                     /* ref :55 */

typedef struct AprilASRModel_i *AprilASRModel;              /* ref :48,51 */
typedef struct AprilASRSession_i *AprilASRSession;          /* ref :49,52 */

/* ref :82-84 -- carried for ABI compatibility, never interpreted (the reference
   does not implement it either). */
typedef struct AprilSpeakerID { uint8_t data[16]; } AprilSpeakerID;

/* ref :86-106 */
typedef enum AprilResultType {
    APRIL_RESULT_UNKNOWN = 0,
    APRIL_RESULT_RECOGNITION_PARTIAL = 1,  /* text so far; later calls repeat it, updated      */
    APRIL_RESULT_RECOGNITION_FINAL = 2,    /* text is final; later calls start from empty       */
    APRIL_RESULT_ERROR_CANT_KEEP_UP = 3,   /* async only: ingest ring overflowed; count=0       */
    APRIL_RESULT_SILENCE = 4               /* emitted once per silence; count=0, tokens=NULL    */
} AprilResultType;

/* ref :108-116 */
typedef enum AprilTokenFlagBits {
    APRIL_TOKEN_FLAG_WORD_BOUNDARY_BIT = 0x00000001,  /* token text starts with ' '             */
    APRIL_TOKEN_FLAG_SENTENCE_END_BIT = 0x00000002    /* token is ".", "!" or "?"              */
} AprilTokenFlagBits;

/* ref :118-137 */
typedef struct AprilToken {
    const char *token;          /* NUL-terminated; valid for the model's lifetime               */
    float logprob;              /* raw joiner logit of the token (the reference does no softmax) */
    AprilTokenFlagBits flags;
    size_t time_ms;             /* audio time at which the token was emitted                    */
    void *reserved;
} AprilToken;

/* ref :142 -- (userdata, result type, token count, tokens); tokens valid during the call only */
typedef void (*AprilRecognitionResultHandler)(void *, AprilResultType, size_t, const AprilToken *);

/* ref :145-162 */
typedef enum AprilConfigFlagBits {
    APRIL_CONFIG_FLAG_ZERO_BIT = 0x00000000,
    APRIL_CONFIG_FLAG_ASYNC_RT_BIT = 0x00000001,
    APRIL_CONFIG_FLAG_ASYNC_NO_RT_BIT = 0x00000002
} AprilConfigFlagBits;

/* ref :164-174 -- passed BY VALUE to aas_create_session */
typedef struct AprilConfig {
    AprilSpeakerID speaker;
    AprilRecognitionResultHandler handler;   /* required; NULL makes aas_create_session fail     */
    void *userdata;
    AprilConfigFlagBits flags;
} AprilConfig;

/* ref :58  must be called once first; picks the GPU(s) (APRIL_GPU_DEVICES) and log level (APRIL_LOG_LEVEL). */
APRIL_EXPORT void aam_api_init(int version);
/* ref :61  loads a .april file, uploads packed weights to HBM; NULL on any failure. */
APRIL_EXPORT AprilASRModel aam_create_model(const char *model_path);
/* ref :65-67  pointers owned by the model */
APRIL_EXPORT const char *aam_get_name(AprilASRModel model);
APRIL_EXPORT const char *aam_get_description(AprilASRModel model);
APRIL_EXPORT const char *aam_get_language(AprilASRModel model);
/* ref :70 */
APRIL_EXPORT size_t aam_get_sample_rate(AprilASRModel model);
/* ref :74  all sessions of the model must be freed first; NULL is a no-op */
APRIL_EXPORT void aam_free(AprilASRModel model);

/* ref :178  allocates a state slot in HBM on the least-loaded GPU; NULL if handler is NULL */
APRIL_EXPORT AprilASRSession aas_create_session(AprilASRModel model, AprilConfig config);
/* ref :183  short_count is a count of int16 samples, mono, at aam_get_sample_rate() */
APRIL_EXPORT void aas_feed_pcm16(AprilASRSession session, short *pcm16, size_t short_count);
/* ref :186 */
APRIL_EXPORT void aas_flush(AprilASRSession session);
/* ref :192 */
APRIL_EXPORT float aas_realtime_get_speedup(AprilASRSession session);
/* ref :196  NULL is a no-op; does not flush */
APRIL_EXPORT void aas_free(AprilASRSession session);

#ifdef __cplusplus
}
#endif
#endif
