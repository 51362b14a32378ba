/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * .april container + PARAMS block reader, restated from
 *   src/file/model_file.c:58-86   (magic, version == 1, header_size)
 *   src/file/model_file.c:88-129  (language[8], name, description, type in (0,2),
 *                                  params entry, <= 8 network entries, bounds)
 *   src/params.c:46-112           (13 int32 + token table, range checks)
 *   extra/file-format.md          (layout)
 *
 * Difference from the reference: the whole file is read into memory and the
 * params block is located through params_off (the reference reads it from the
 * current FILE position, model_file.c:164-166, which only works because the
 * exporter writes params right after network 2; same bytes either way).
 *
 * Pinned against the compiled reference parser (oracle/_ref) in
 * tests/test_oracle_file.py, including the rejection cases.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

typedef struct { const uint8_t *p; size_t n, pos; int bad; } Rd;

static uint64_t rd_le(Rd *r, int bytes)
{
    uint64_t v = 0;
    if (r->pos + (size_t)bytes > r->n) { r->bad = 1; r->pos = r->n; return 0; }
    for (int i = 0; i < bytes; ++i) v |= (uint64_t)r->p[r->pos + i] << (8 * i);
    r->pos += (size_t)bytes;
    return v;
}
static int32_t rd_i32(Rd *r) { return (int32_t)(uint32_t)rd_le(r, 4); }

static char *rd_string(Rd *r)
{
    uint64_t len = rd_le(r, 8);
    if (r->bad || len > r->n - r->pos) { r->bad = 1; return NULL; }
    char *s = (char *)malloc(len + 1);
    memcpy(s, r->p + r->pos, len);
    s[len] = 0;
    r->pos += len;
    return s;
}

const char *orc_token(const OrcParams *p, size_t idx) { return p->tokens + p->token_stride * idx; }

/* params.c:46-112 */
static int parse_params(Rd *r, OrcParams *o)
{
    static const char magic[8] = {'P', 'A', 'R', 'A', 'M', 'S', 0, 0};
    if (r->pos + 8 > r->n || memcmp(r->p + r->pos, magic, 8) != 0) return 0;
    r->pos += 8;
    o->batch_size = rd_i32(r);
    o->segment_size = rd_i32(r);
    o->segment_step = rd_i32(r);
    o->mel_features = rd_i32(r);
    o->sample_rate = rd_i32(r);
    o->frame_shift_ms = rd_i32(r);
    o->frame_length_ms = rd_i32(r);
    o->round_pow2 = rd_i32(r) != 0;
    o->mel_low = rd_i32(r);
    o->mel_high = rd_i32(r);
    o->snip_edges = rd_i32(r) != 0;
    o->token_count = rd_i32(r);
    o->blank_id = rd_i32(r);
    if (r->bad) return 0;
    /* params.c:71-82 */
    if (o->batch_size != 1) return 0;
    if (!(o->segment_size > 0 && o->segment_size < 100)) return 0;
    if (!(o->segment_step > 0 && o->segment_step < 100 && o->segment_step <= o->segment_size)) return 0;
    if (!(o->mel_features > 0 && o->mel_features < 256)) return 0;
    if (!(o->sample_rate > 0 && o->sample_rate < 144000)) return 0;
    if (!(o->token_count > 0 && o->token_count < 16384)) return 0;
    if (!(o->blank_id >= 0 && o->blank_id < o->token_count)) return 0;
    if (!(o->frame_shift_ms > 0 && o->frame_shift_ms <= o->frame_length_ms)) return 0;
    if (!(o->frame_length_ms > 0 && o->frame_length_ms <= 5000)) return 0;
    if (!(o->mel_low > 0 && o->mel_low < o->sample_rate)) return 0;
    if (!(o->mel_high == 0 || o->mel_high > o->mel_low)) return 0;
    /* two passes over the token list: longest, then copy (params.c:86-109) */
    size_t start = r->pos, longest = 0;
    for (int i = 0; i < o->token_count; ++i) {
        int32_t len = rd_i32(r);
        if (r->bad || len < 0 || (size_t)len > r->n - r->pos) return 0;
        if ((size_t)len > longest) longest = (size_t)len;
        r->pos += (size_t)len;
    }
    o->token_stride = longest + 1;
    o->tokens = (char *)calloc((size_t)o->token_count, o->token_stride);
    r->pos = start;
    for (int i = 0; i < o->token_count; ++i) {
        int32_t len = rd_i32(r);
        memcpy(o->tokens + o->token_stride * (size_t)i, r->p + r->pos, (size_t)len);
        r->pos += (size_t)len;
    }
    return 1;
}

void orc_file_free(OrcFile *f)
{
    if (!f) return;
    free(f->name); free(f->description); free(f->blob); free(f->params.tokens);
    free(f);
}

OrcFile *orc_file_open(const char *path)
{
    FILE *fd = fopen(path, "rb");
    if (!fd) return NULL;
    OrcFile *f = (OrcFile *)calloc(1, sizeof(*f));
    fseek(fd, 0, SEEK_END);
    long sz = ftell(fd);
    fseek(fd, 0, SEEK_SET);
    f->blob_size = sz > 0 ? (size_t)sz : 0;
    f->blob = (uint8_t *)malloc(f->blob_size ? f->blob_size : 1);
    if (fread(f->blob, 1, f->blob_size, fd) != f->blob_size) { fclose(fd); orc_file_free(f); return NULL; }
    fclose(fd);

    Rd r = {f->blob, f->blob_size, 0, 0};
    if (r.n < 20 || memcmp(r.p, "APRILMDL", 8) != 0) goto fail;      /* model_file.c:68-71 */
    r.pos = 8;
    if ((uint32_t)rd_le(&r, 4) != 1) goto fail;                          /* :74-78 */
    (void)rd_le(&r, 8);                                                  /* header_size, unused */
    if (r.pos + 8 > r.n) goto fail;
    memcpy(f->language, r.p + r.pos, 8);                                 /* :94-95 */
    f->language[8] = 0;
    r.pos += 8;
    f->name = rd_string(&r);
    f->description = rd_string(&r);
    if (r.bad) goto fail;
    f->model_type = (uint32_t)rd_le(&r, 4);
    if (!(f->model_type > 0 && f->model_type < 2)) goto fail;            /* :100-104 */
    f->params_off = rd_le(&r, 8);
    f->params_size = rd_le(&r, 8);
    if (r.bad || f->params_off + f->params_size > f->blob_size) goto fail;   /* :108-111 */
    f->n_networks = rd_le(&r, 8);
    if (r.bad || f->n_networks > 8) goto fail;                           /* :114-117 */
    for (uint64_t i = 0; i < f->n_networks; ++i) {
        f->net_off[i] = rd_le(&r, 8);
        f->net_size[i] = rd_le(&r, 8);
        if (r.bad || f->net_off[i] + f->net_size[i] > f->blob_size) goto fail;   /* :122-125 */
    }
    {
        Rd pr = {f->blob, f->blob_size, (size_t)f->params_off, 0};
        if (!parse_params(&pr, &f->params)) goto fail;
    }
    return f;
fail:
    orc_file_free(f);
    return NULL;
}
