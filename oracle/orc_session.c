/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of the reference's per-session runtime:
 *   src/april_session.c:25-93    session creation (state zeroed, emitted_silence starts true)
 *   src/april_session.c:131-196  encoder / decoder / joiner invocation, context shift
 *   src/april_session.c:199-294  finalize / finalize_previous_words / emit_silence / emit_token
 *   src/april_session.c:296-301  clear_context (tests context[0], quirk kept)
 *   src/april_session.c:306-429  greedy decision per joiner output
 *   src/april_session.c:431-476  chunk loop (early-emit schedule 1,0,0)
 *   src/april_session.c:501-538  PCM16 -> float, 3200-sample segments
 *   src/april_session.c:547-564  flush = pad-drain, 2 x 3200 zeros, pad-drain, FINAL, clear, SILENCE
 *   src/april_model.c:24-107     model load (3 networks, dims from graph I/O)
 *
 * The reference's own april_session.c cannot be compiled in this image (it
 * includes onnxruntime_c_api.h) so this part is PARITY UNPINNED; it is checked
 * by hand-derived expectations in tests/test_state_machine.py.
 *
 * Tokens are carried as vocabulary indices; the reference carries char*
 * into the model's token table (src/params.c:31-33) -- same identity.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define ORC_MAX_ACTIVE 72   /* april_session.h:30 */
#define ORC_SEG 3200        /* april_session.c:500 */

struct OrcSession {
    const OrcParams *P;
    OrcModel *model;
    OrcNets nets;
    OrcFbank *fb;
    size_t x_elems, h_elems, c_elems, e_elems, vocab;
    float *x, *h[2], *c[2], *eout, *dout, *logits;
    int64_t ctx[8]; int ctx_n;
    int flip, dout_ready;
    OrcToken active[ORC_MAX_ACTIVE];
    size_t head, last_call_head;
    int emitted_silence, flushed;
    uint64_t now_ms, last_emit_ms, chunks;
    OrcHandler handler; void *ud;
    float *ltrace; size_t ltrace_cap, *ltrace_used;
    float *ctrace; size_t ctrace_cap, *ctrace_used;
};

/* ---- default network provider: the three graphs ---- */
static void graph_encoder(void *ud, const float *x, const float *h, const float *c, float *eout, float *h2, float *c2)
{
    OrcModel *m = ud;
    const char *in[] = {"x", "h", "c"}; const void *ib[] = {x, h, c};
    const char *on[] = {"encoder_out", "next_h", "next_c"}; void *ob[] = {eout, h2, c2};
    if (orc_graph_run(m->enc, 3, in, ib, 3, on, ob)) { fprintf(stderr, "oracle encoder: %s\n", orc_graph_last_error()); abort(); }
}
static void graph_decoder(void *ud, const int64_t *ctx, float *dout)
{
    OrcModel *m = ud;
    const char *in[] = {"context"}; const void *ib[] = {ctx};
    const char *on[] = {"decoder_out"}; void *ob[] = {dout};
    if (orc_graph_run(m->dec, 1, in, ib, 1, on, ob)) { fprintf(stderr, "oracle decoder: %s\n", orc_graph_last_error()); abort(); }
}
static void graph_joiner(void *ud, const float *e, const float *d, float *logits)
{
    OrcModel *m = ud;
    const char *in[] = {"encoder_out", "decoder_out"}; const void *ib[] = {e, d};
    const char *on[] = {"logits"}; void *ob[] = {logits};
    if (orc_graph_run(m->joi, 2, in, ib, 1, on, ob)) { fprintf(stderr, "oracle joiner: %s\n", orc_graph_last_error()); abort(); }
}

/* april_model.c:24-107 */
OrcModel *orc_model_load(const char *path)
{
    OrcFile *f = orc_file_open(path);
    if (!f) return NULL;
    if (f->model_type != 1 || f->n_networks != 3) { orc_file_free(f); return NULL; }
    OrcModel *m = calloc(1, sizeof *m);
    m->file = f;
    m->enc = orc_graph_parse(f->blob + f->net_off[0], f->net_size[0]);
    m->dec = orc_graph_parse(f->blob + f->net_off[1], f->net_size[1]);
    m->joi = orc_graph_parse(f->blob + f->net_off[2], f->net_size[2]);
    if (!m->enc || !m->dec || !m->joi) goto bad;
    if (orc_graph_num_inputs(m->enc) != 3 || orc_graph_num_outputs(m->enc) != 3) goto bad;
    if (orc_graph_num_inputs(m->dec) != 1 || orc_graph_num_outputs(m->dec) != 1) goto bad;
    if (orc_graph_num_inputs(m->joi) != 2 || orc_graph_num_outputs(m->joi) != 1) goto bad;
    if (orc_graph_input_dims(m->enc, 0, m->x_dim, 3) != 3) goto bad;
    if (orc_graph_input_dims(m->enc, 1, m->h_dim, 3) != 3) goto bad;
    if (orc_graph_input_dims(m->enc, 2, m->c_dim, 3) != 3) goto bad;
    if (orc_graph_output_dims(m->enc, 0, m->eout_dim, 3) != 3) goto bad;
    if (orc_graph_input_dims(m->dec, 0, m->ctx_dim, 2) != 2) goto bad;
    if (orc_graph_output_dims(m->dec, 0, m->dout_dim, 3) != 3) goto bad;
    if (orc_graph_output_dims(m->joi, 0, m->logits_dim, 3) != 3) goto bad;
    /* april_model.c:99-102 */
    if (m->x_dim[0] != f->params.batch_size || m->x_dim[1] != f->params.segment_size ||
        m->x_dim[2] != f->params.mel_features || m->logits_dim[2] != f->params.token_count) goto bad;
    return m;
bad:
    orc_model_free(m);
    return NULL;
}

void orc_model_free(OrcModel *m)
{
    if (!m) return;
    orc_graph_free(m->enc); orc_graph_free(m->dec); orc_graph_free(m->joi);
    orc_file_free(m->file);
    free(m);
}

static OrcSession *session_alloc(const OrcParams *P, size_t h_elems, size_t c_elems, size_t e_elems, size_t vocab,
                                 int ctx_n, OrcHandler handler, void *ud)
{
    if (!handler) return NULL;                   /* april_session.c:81-85 */
    OrcSession *s = calloc(1, sizeof *s);
    s->P = P;
    s->fb = orc_fbank_new(P->sample_rate, P->frame_shift_ms, P->frame_length_ms, P->mel_features, P->round_pow2,
                          P->mel_low, P->mel_high, P->segment_size, P->segment_step);
    s->x_elems = (size_t)P->segment_size * (size_t)P->mel_features;
    s->h_elems = h_elems; s->c_elems = c_elems; s->e_elems = e_elems; s->vocab = vocab;
    s->x = calloc(s->x_elems, 4);
    for (int i = 0; i < 2; ++i) { s->h[i] = calloc(h_elems, 4); s->c[i] = calloc(c_elems, 4); }
    s->eout = calloc(e_elems, 4); s->dout = calloc(e_elems, 4); s->logits = calloc(vocab, 4);
    s->ctx_n = ctx_n;
    s->emitted_silence = 1;                      /* april_session.c:64 */
    s->handler = handler; s->ud = ud;
    for (int i = 0; i < ORC_MAX_ACTIVE; ++i) s->active[i].id = -1;
    return s;
}

OrcSession *orc_session_new(OrcModel *m, OrcHandler handler, void *ud)
{
    if (m->ctx_dim[0] != 1) return NULL;         /* april_session.c:51-55 */
    OrcSession *s = session_alloc(&m->file->params,
                                  (size_t)(m->h_dim[0] * m->h_dim[1] * m->h_dim[2]),
                                  (size_t)(m->c_dim[0] * m->c_dim[1] * m->c_dim[2]),
                                  (size_t)(m->eout_dim[0] * m->eout_dim[1] * m->eout_dim[2]),
                                  (size_t)m->logits_dim[2], (int)m->ctx_dim[1], handler, ud);
    if (!s) return NULL;
    s->model = m;
    s->nets.ud = m; s->nets.encoder = graph_encoder; s->nets.decoder = graph_decoder; s->nets.joiner = graph_joiner;
    return s;
}

OrcSession *orc_session_new_scripted(const OrcParams *p, const OrcNets *nets, int n_layers_h, int h_elems, int c_elems,
                                     int e_elems, int vocab, OrcHandler h, void *ud)
{
    (void)n_layers_h;
    OrcSession *s = session_alloc(p, (size_t)h_elems, (size_t)c_elems, (size_t)e_elems, (size_t)vocab, 2, h, ud);
    if (s) s->nets = *nets;
    return s;
}

void orc_session_free(OrcSession *s)
{
    if (!s) return;
    orc_fbank_free(s->fb);
    free(s->x); for (int i = 0; i < 2; ++i) { free(s->h[i]); free(s->c[i]); }
    free(s->eout); free(s->dout); free(s->logits);
    free(s);
}

void orc_session_set_logit_trace(OrcSession *s, float *buf, size_t cap, size_t *used) { s->ltrace = buf; s->ltrace_cap = cap; s->ltrace_used = used; }
void orc_session_set_chunk_trace(OrcSession *s, float *buf, size_t cap, size_t *used) { s->ctrace = buf; s->ctrace_cap = cap; s->ctrace_used = used; }
uint64_t orc_session_chunks(const OrcSession *s) { return s->chunks; }

/* april_session.c:131-148: ping-pong h/c */
static void run_encoder(OrcSession *s)
{
    s->flip = !s->flip;
    int src = s->flip ? 0 : 1, dst = s->flip ? 1 : 0;
    s->nets.encoder(s->nets.ud, s->x, s->h[src], s->c[src], s->eout, s->h[dst], s->c[dst]);
}

/* april_session.c:181-196 */
static void push_context(OrcSession *s, int64_t tok)
{
    for (int i = 0; i + 1 < s->ctx_n; ++i) s->ctx[i] = s->ctx[i + 1];
    s->ctx[s->ctx_n - 1] = tok;
    s->nets.decoder(s->nets.ud, s->ctx, s->dout);
}

/* april_session.c:199-211 */
static void finalize_all(OrcSession *s)
{
    if (s->head == 0) return;
    s->handler(s->ud, 2, s->head, s->active);
    s->last_call_head = s->head;
    s->head = 0;
}

/* april_session.c:213-255 */
static void finalize_before_word(OrcSession *s, const OrcToken *incoming)
{
    if (s->head == 0) return;
    if (incoming->flags & 1) { finalize_all(s); return; }
    size_t start = ORC_MAX_ACTIVE;
    for (size_t i = s->head - 1; i > 2; --i)
        if (s->active[i].flags & 1) { start = i; break; }
    if (start == ORC_MAX_ACTIVE) { finalize_all(s); return; }
    s->handler(s->ud, 2, start, s->active);
    memmove(s->active, &s->active[start], sizeof(OrcToken) * (s->head - start));
    s->head -= start;
}

/* april_session.c:257-268 */
static void emit_silence(OrcSession *s)
{
    if (s->emitted_silence) return;
    s->emitted_silence = 1;
    s->handler(s->ud, 4, 0, NULL);
}

/* april_session.c:270-294 */
static int emit_partial(OrcSession *s, const OrcToken *tok, int force)
{
    if (tok) {
        if (!force && s->last_call_head == s->head + 1 && s->active[s->head].id == tok->id) return 0;
        s->active[s->head++] = *tok;
    } else {
        if (!force && s->last_call_head == s->head) return 0;
    }
    s->handler(s->ud, 1, s->head, s->active);
    s->last_call_head = s->head;
    return 1;
}

/* april_session.c:296-301 */
static void clear_context(OrcSession *s)
{
    if (s->ctx[0] == s->P->blank_id) return;
    for (int i = 0; i < s->ctx_n; ++i) push_context(s, s->P->blank_id);
}

static int is_sentence_end_text(const char *t) { return t[1] == 0 && (t[0] == '.' || t[0] == '!' || t[0] == '?'); }

/* april_session.c:306-429; returns 1 when the round resolved to blank */
static int decide(OrcSession *s, float early_emit)
{
    const OrcParams *P = s->P;
    const size_t blank = (size_t)P->blank_id;
    const float *lg = s->logits;

    int best = -1; float best_v = -9999999999.0f;          /* :311-320 */
    for (size_t i = 0; i < (size_t)P->token_count; ++i) {
        if (i == blank) continue;
        if (lg[i] > best_v) { best = (int)i; best_v = lg[i]; }
    }
    if (best < 0) best = blank == 0 ? 1 : 0;                /* NaN guard; reference would index -1 */

    const int cleared = s->ctx[1] == (int64_t)P->blank_id;  /* :322 */
    const int same = s->ctx[1] == (int64_t)best;            /* :326 */
    if (same) early_emit = 0.0f;
    const float blank_v = lg[blank];
    int is_blank = (blank_v - early_emit) > best_v;         /* :329-330 */

    const char *txt = orc_token(P, (size_t)best);
    OrcToken tok; memset(&tok, 0, sizeof tok);
    tok.id = best; tok.logprob = best_v; tok.time_ms = s->now_ms;
    if (txt[0] == ' ') tok.flags |= 1;                      /* :338 */
    int single = txt[1] == 0;
    int eos = single && (txt[0] == '.' || txt[0] == '!' || txt[0] == '?');
    int punct = eos || (single && txt[0] == ',');
    if (punct && s->head > 0) {                             /* :345-351 */
        const char *last = orc_token(P, (size_t)s->active[s->head - 1].id);
        if (last[0] >= '0' && last[0] <= '9' && txt[0] == '.') { eos = 0; punct = 0; }
    }
    if (eos) tok.flags |= 2;
    if (!cleared && punct && !same && best_v > (blank_v - 3.5f)) is_blank = 0;   /* :356-358 */

    if (!is_blank) {                                        /* :361-400 */
        s->last_emit_ms = s->now_ms;
        push_context(s, (int64_t)best);
        int fin = s->head >= (ORC_MAX_ACTIVE - 1);
        if (s->head > 0 && (tok.flags & 1)) {
            OrcToken *prev = &s->active[s->head - 1];
            int prev_eos = is_sentence_end_text(orc_token(P, (size_t)prev->id));
            if (prev_eos && !(prev->flags & 2)) prev->flags |= 2;
            if (prev_eos) fin = 1;
        }
        if (fin) finalize_before_word(s, &tok);
        if (s->head >= (ORC_MAX_ACTIVE - 1)) s->head = 0;   /* "No room left" :391-394 */
        emit_partial(s, &tok, 1);
        s->emitted_silence = 0;
    } else {                                                /* :401-426 */
        uint64_t gap = s->now_ms - s->last_emit_ms;
        float decayed = best_v - (float)gap / 3000.0f;
        int confident = !same && decayed > (blank_v - 4.0f);
        if (gap >= 2200) {
            finalize_all(s);
            clear_context(s);
            emit_silence(s);
        } else if (confident) {
            tok.logprob -= 8.0f;
            if (emit_partial(s, &tok, 0)) s->head--;
        } else {
            emit_partial(s, NULL, 0);
        }
    }
    return is_blank;
}

/* april_session.c:431-476 */
static void drain_chunks(OrcSession *s)
{
    if (!s->dout_ready) {
        for (int i = 0; i < s->ctx_n; ++i) push_context(s, s->P->blank_id);
        s->dout_ready = 1;
    }
    while (orc_fbank_pull(s->fb, s->x)) {
        s->now_ms += (uint64_t)orc_fbank_stride_ms(s->fb);
        s->chunks++;
        if (s->ctrace && *s->ctrace_used + s->x_elems <= s->ctrace_cap) {
            memcpy(s->ctrace + *s->ctrace_used, s->x, s->x_elems * 4);
            *s->ctrace_used += s->x_elems;
        }
        run_encoder(s);
        for (int round = 0; round < 3; ++round) {
            s->nets.joiner(s->nets.ud, s->eout, s->dout, s->logits);
            if (s->ltrace && *s->ltrace_used + s->vocab <= s->ltrace_cap) {
                memcpy(s->ltrace + *s->ltrace_used, s->logits, s->vocab * 4);
                *s->ltrace_used += s->vocab;
            }
            if (decide(s, round == 0 ? 1.0f : 0.0f)) break;
        }
    }
}

/* april_session.c:501-538 */
void orc_session_feed_pcm16(OrcSession *s, const int16_t *pcm, size_t n)
{
    float wave[ORC_SEG];
    s->flushed = 0;
    size_t head = 0;
    while (head < n) {
        size_t take = n - head > ORC_SEG ? ORC_SEG : n - head;
        for (size_t i = 0; i < take; ++i) wave[i] = (float)pcm[head + i] / 32768.0f;
        orc_fbank_accept(s->fb, wave, take);
        drain_chunks(s);
        head += take;
    }
}

/* april_session.c:547-564 */
void orc_session_flush(OrcSession *s)
{
    if (s->flushed) return;
    s->flushed = 1;
    while (orc_fbank_flush(s->fb)) drain_chunks(s);
    for (int i = 0; i < 2; ++i) orc_fbank_accept(s->fb, NULL, ORC_SEG);
    while (orc_fbank_flush(s->fb)) drain_chunks(s);
    finalize_all(s);
    clear_context(s);
    emit_silence(s);
}
