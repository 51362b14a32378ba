/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Minimal ONNX reader + fp32 interpreter: executes the three graphs embedded
 * in a .april file (encoder / decoder / joiner) following ONNX operator
 * semantics (opset 11 as written by extra/export-april.py:226-331).
 *
 * The reference delegates this arithmetic to ONNXRuntime 1.13.1's CPU
 * provider (download_onnx_linux_x64.sh:8; call sites src/april_session.c:145,
 * 160,176; graph load src/ort_util.h:127-134).  ORT is not vendored under
 * /root/reference and not installed in this image, so this interpreter is a
 * restatement of the published ONNX operator definitions, anchored on the
 * reference's call sites (tensor names/shapes, src/april_session.c:121-128).
 * PARITY UNPINNED versus ORT itself.  Its operator semantics are cross-checked on a
 * code-disjoint implementation: a PyTorch fp32 statement of the same architecture
 * (tests/torch_ref.py, tests/test_oracle_vs_torch.py; agreement within 2e-5).
 *
 * Summation order: every dot product is a plain left-to-right float chain in
 * k (MatMul / Conv) or 8 interleaved partial chains (Gemm with transB=1), one
 * rounding per product and per add (no FMA).  MLAS uses a different blocking,
 * so agreement with ORT is expected at fp32 round-off level, not bitwise.
 *
 * Supported ops: Conv MatMul Gemm Add Sub Mul Div Pow Sqrt Exp Neg ReduceMean
 * Sigmoid Tanh Relu Split Slice Concat Gather Squeeze Unsqueeze Transpose
 * Reshape Shape Cast Constant ConstantOfShape Identity.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define MAXR 6
enum { DT_F32 = 1, DT_I32 = 6, DT_I64 = 7 };

typedef struct {
    int dtype, rank, owns, is_const, live;
    int64_t dims[MAXR];
    size_t n;
    void *data;
    float *h16;          /* fp16-rounded shadow of a constant MatMul/Gemm operand (orc_set_f16_linear), built on first use */
} Ten;

typedef struct { char *name; int kind; float f; int64_t i; int64_t *ints; int n_ints; Ten t; int has_t; } Attr;

typedef struct {
    char *op;
    int n_in, n_out;
    int in[64], out[8];
    Attr *attrs; int n_attrs;
    int is_const;
} Node;

struct OrcGraph {
    char **vname; Ten *val; int n_val, cap_val;
    Node *nodes; int n_nodes, cap_nodes;
    int in_ids[8], n_in; int out_ids[8], n_out;
    int64_t in_dims[8][MAXR]; int in_rank[8]; int in_dt[8];
    int64_t out_dims[8][MAXR]; int out_rank[8]; int out_dt[8];
    int folded;
};

static char g_err[256];
const char *orc_graph_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* ---------------- protobuf wire ---------------- */
typedef struct { const uint8_t *p, *end; } PB;
static uint64_t pb_varint(PB *b)
{
    uint64_t v = 0; int s = 0;
    while (b->p < b->end) { uint8_t c = *b->p++; if (s < 64) v |= (uint64_t)(c & 0x7f) << s; if (!(c & 0x80)) break; s += 7; }
    return v;
}
/* returns field number (0 at the end or on malformed input), sets wire type; for LEN fields sets sub, otherwise sub is
 * empty.  Never reads or points beyond b->end: model files are untrusted input (tests/test_loader.py fuzzes them). */
static int pb_next(PB *b, int *wt, PB *sub, uint64_t *v)
{
    sub->p = sub->end = b->end;
    *v = 0;
    if (b->p >= b->end) return 0;
    uint64_t key = pb_varint(b);
    *wt = (int)(key & 7);
    int field = (int)(key >> 3);
    const size_t left = (size_t)(b->end - b->p);
    switch (*wt) {
    case 0: *v = pb_varint(b); break;
    case 1: if (left < 8) { b->p = b->end; return 0; } memcpy(v, b->p, 8); b->p += 8; break;
    case 5: { if (left < 4) { b->p = b->end; return 0; } uint32_t t; memcpy(&t, b->p, 4); *v = t; b->p += 4; break; }
    case 2: { uint64_t len = pb_varint(b); if (len > (uint64_t)(b->end - b->p)) { b->p = b->end; return 0; }
              sub->p = b->p; sub->end = b->p + len; b->p += len; break; }
    default: b->p = b->end; return 0;
    }
    return field > 0 ? field : 0;
}
static char *pb_str(const PB *s) { size_t n = (size_t)(s->end - s->p); char *r = malloc(n + 1); memcpy(r, s->p, n); r[n] = 0; return r; }

static size_t numel(const Ten *t) { size_t n = 1; for (int i = 0; i < t->rank; ++i) n *= (size_t)t->dims[i]; return n; }

static void ten_alloc(Ten *t, int dtype, int rank, const int64_t *dims)
{
    t->dtype = dtype; t->rank = rank;
    for (int i = 0; i < rank; ++i) t->dims[i] = dims[i];
    t->n = numel(t);
    t->data = calloc(t->n ? t->n : 1, dtype == DT_F32 ? 4 : 8);
    t->owns = 1; t->live = 1;
}
static void ten_release(Ten *t) { if (t->owns && t->data) free(t->data); if (t->h16) free(t->h16); t->h16 = NULL; t->data = NULL; t->owns = 0; t->live = 0; }

/* TensorProto -> Ten (float32 / int64; int32 widened) */
static int parse_tensor(PB b, Ten *t, char **name_out)
{
    memset(t, 0, sizeof *t);
    int wt; PB sub; uint64_t v;
    int64_t dims[MAXR]; int rank = 0, dtype = 0;
    PB raw = {0, 0}; int have_raw = 0;
    float *fdata = NULL; size_t nf = 0, capf = 0;
    int64_t *idata = NULL; size_t ni = 0, capi = 0;
    int f;
    while ((f = pb_next(&b, &wt, &sub, &v))) {
        if (f == 1) {
            if (wt == 2) { PB q = sub; while (q.p < q.end && rank < MAXR) dims[rank++] = (int64_t)pb_varint(&q); }
            else if (rank < MAXR) dims[rank++] = (int64_t)v;
        } else if (f == 2) dtype = (int)v;
        else if (f == 4) {
            if (wt == 2) { size_t k = (size_t)(sub.end - sub.p) / 4; if (nf + k > capf) { capf = (nf + k) * 2; fdata = realloc(fdata, capf * 4); } memcpy(fdata + nf, sub.p, k * 4); nf += k; }
            else { if (nf + 1 > capf) { capf = capf ? capf * 2 : 16; fdata = realloc(fdata, capf * 4); } uint32_t u = (uint32_t)v; memcpy(fdata + nf, &u, 4); nf++; }
        } else if (f == 5 || f == 7) {
            if (wt == 2) { PB q = sub; while (q.p < q.end) { if (ni + 1 > capi) { capi = capi ? capi * 2 : 16; idata = realloc(idata, capi * 8); } idata[ni++] = (int64_t)pb_varint(&q); } }
            else { if (ni + 1 > capi) { capi = capi ? capi * 2 : 16; idata = realloc(idata, capi * 8); } idata[ni++] = (int64_t)v; }
        } else if (f == 8 && name_out) *name_out = pb_str(&sub);
        else if (f == 9) { raw = sub; have_raw = 1; }
    }
    if (dtype != DT_F32 && dtype != DT_I64 && dtype != DT_I32) { free(fdata); free(idata); FAIL("tensor dtype %d unsupported", dtype); }
    {   /* the element count must be what the payload holds BEFORE anything is allocated from it */
        const size_t esz = dtype == DT_I64 ? 8 : 4;
        const size_t have = have_raw ? (size_t)(raw.end - raw.p) / esz : (dtype == DT_F32 ? nf : ni);
        size_t want = 1;
        for (int i = 0; i < rank; ++i) {
            if (dims[i] < 0 || (dims[i] > 0 && want > ((size_t)1 << 40) / (size_t)dims[i])) { free(fdata); free(idata); FAIL("tensor dims"); }
            want *= (size_t)dims[i];
        }
        if (want != have || (have_raw && (size_t)(raw.end - raw.p) != want * esz)) { free(fdata); free(idata); FAIL("tensor payload size"); }
    }
    ten_alloc(t, dtype == DT_F32 ? DT_F32 : DT_I64, rank, dims);
    if (dtype == DT_F32) {
        if (have_raw) { if ((size_t)(raw.end - raw.p) != t->n * 4) FAIL("raw size"); memcpy(t->data, raw.p, t->n * 4); }
        else { if (nf != t->n) FAIL("float_data size"); memcpy(t->data, fdata, nf * 4); }
    } else if (dtype == DT_I64) {
        if (have_raw) { if ((size_t)(raw.end - raw.p) != t->n * 8) FAIL("raw size"); memcpy(t->data, raw.p, t->n * 8); }
        else { if (ni != t->n) FAIL("int64_data size"); memcpy(t->data, idata, ni * 8); }
    } else {
        int64_t *d = t->data;
        if (have_raw) { for (size_t i = 0; i < t->n; ++i) { int32_t x; memcpy(&x, raw.p + 4 * i, 4); d[i] = x; } }
        else { if (ni != t->n) FAIL("int32_data size"); for (size_t i = 0; i < ni; ++i) d[i] = (int32_t)idata[i]; }
    }
    t->is_const = 1;
    free(fdata); free(idata);
    return 0;
}

static int val_id(OrcGraph *g, const char *name)
{
    for (int i = 0; i < g->n_val; ++i) if (strcmp(g->vname[i], name) == 0) return i;
    if (g->n_val == g->cap_val) {
        g->cap_val = g->cap_val ? g->cap_val * 2 : 256;
        g->vname = realloc(g->vname, sizeof(char *) * (size_t)g->cap_val);
        g->val = realloc(g->val, sizeof(Ten) * (size_t)g->cap_val);
    }
    g->vname[g->n_val] = strdup(name);
    memset(&g->val[g->n_val], 0, sizeof(Ten));
    return g->n_val++;
}

static int parse_attr(PB b, Attr *a)
{
    memset(a, 0, sizeof *a);
    int wt, f; PB sub; uint64_t v;
    int64_t *ints = NULL; int n = 0, cap = 0;
    while ((f = pb_next(&b, &wt, &sub, &v))) {
        if (f == 1) { free(a->name); a->name = pb_str(&sub); }
        else if (f == 2) { uint32_t u = (uint32_t)v; memcpy(&a->f, &u, 4); }
        else if (f == 3) a->i = (int64_t)v;
        else if (f == 5) { if (a->has_t) { free(a->t.data); a->has_t = 0; } if (parse_tensor(sub, &a->t, NULL)) { free(a->name); free(ints); a->name = NULL; return -1; } a->has_t = 1; }
        else if (f == 8) {
            if (wt == 2) { PB q = sub; while (q.p < q.end) { if (n == cap) { cap = cap ? cap * 2 : 8; ints = realloc(ints, 8 * (size_t)cap); } ints[n++] = (int64_t)pb_varint(&q); } }
            else { if (n == cap) { cap = cap ? cap * 2 : 8; ints = realloc(ints, 8 * (size_t)cap); } ints[n++] = (int64_t)v; }
        }
    }
    a->ints = ints; a->n_ints = n;
    return 0;
}

static int parse_node(OrcGraph *g, PB b)
{
    if (g->n_nodes == g->cap_nodes) { g->cap_nodes = g->cap_nodes ? g->cap_nodes * 2 : 256; g->nodes = realloc(g->nodes, sizeof(Node) * (size_t)g->cap_nodes); }
    Node *nd = &g->nodes[g->n_nodes];
    memset(nd, 0, sizeof *nd);
    int wt, f; PB sub; uint64_t v;
    int cap_a = 0;
    while ((f = pb_next(&b, &wt, &sub, &v))) {
        if (f == 1) { char *s = pb_str(&sub); if (nd->n_in >= 64) { free(s); FAIL("too many inputs"); } nd->in[nd->n_in++] = s[0] ? val_id(g, s) : -1; free(s); }
        else if (f == 2) { char *s = pb_str(&sub); if (nd->n_out >= 8) { free(s); FAIL("too many outputs"); } nd->out[nd->n_out++] = val_id(g, s); free(s); }
        else if (f == 4) { free(nd->op); nd->op = pb_str(&sub); }
        else if (f == 5) {
            if (nd->n_attrs == cap_a) { cap_a = cap_a ? cap_a * 2 : 4; nd->attrs = realloc(nd->attrs, sizeof(Attr) * (size_t)cap_a); }
            if (parse_attr(sub, &nd->attrs[nd->n_attrs])) return -1;
            nd->n_attrs++;
        }
    }
    if (!nd->op) FAIL("node without op_type");
    g->n_nodes++;
    return 0;
}

static int parse_value_info(PB b, char **name, int *dt, int64_t *dims, int *rank)
{
    int wt, f; PB sub, s2, s3, s4; uint64_t v;
    *rank = 0; *dt = 0; *name = NULL;
    while ((f = pb_next(&b, &wt, &sub, &v))) {
        if (f == 1) *name = pb_str(&sub);
        else if (f == 2) {           /* TypeProto */
            PB t = sub; int f2;
            while ((f2 = pb_next(&t, &wt, &s2, &v))) if (f2 == 1) {   /* tensor_type */
                PB tt = s2; int f3;
                while ((f3 = pb_next(&tt, &wt, &s3, &v))) {
                    if (f3 == 1) *dt = (int)v;
                    else if (f3 == 2) {  /* shape */
                        PB sh = s3; int f4;
                        while ((f4 = pb_next(&sh, &wt, &s4, &v))) if (f4 == 1) {
                            PB d = s4; int f5; PB s5; int64_t dv = -1;
                            while ((f5 = pb_next(&d, &wt, &s5, &v))) if (f5 == 1) dv = (int64_t)v;
                            if (*rank < MAXR) dims[(*rank)++] = dv;
                        }
                    }
                }
            }
        }
    }
    return 0;
}

static const Attr *attr_find(const Node *n, const char *name)
{
    for (int i = 0; i < n->n_attrs; ++i) if (n->attrs[i].name && strcmp(n->attrs[i].name, name) == 0) return &n->attrs[i];
    return NULL;
}
static int64_t attr_i(const Node *n, const char *name, int64_t def) { const Attr *a = attr_find(n, name); return a ? a->i : def; }
static float attr_f(const Node *n, const char *name, float def) { const Attr *a = attr_find(n, name); return a ? a->f : def; }

OrcGraph *orc_graph_parse(const uint8_t *bytes, size_t n)
{
    OrcGraph *g = calloc(1, sizeof *g);
    PB m = {bytes, bytes + n}, sub; int wt, f; uint64_t v;
    PB graph = {0, 0};
    while ((f = pb_next(&m, &wt, &sub, &v))) if (f == 7 && wt == 2) graph = sub;
    if (!graph.p) { snprintf(g_err, sizeof g_err, "no graph in model"); free(g); return NULL; }
    /* first pass: initializers (so their ids exist as constants) */
    PB b = graph;
    char *init_names[1] = {0}; (void)init_names;
    while ((f = pb_next(&b, &wt, &sub, &v))) {
        if (f == 5) {
            Ten t; char *nm = NULL;
            if (parse_tensor(sub, &t, &nm)) { free(nm); orc_graph_free(g); return NULL; }
            int id = val_id(g, nm ? nm : "");
            g->val[id] = t;
            free(nm);
        }
    }
    b = graph;
    while ((f = pb_next(&b, &wt, &sub, &v))) {
        if (f == 1) { if (parse_node(g, sub)) { orc_graph_free(g); return NULL; } }
        else if (f == 11 || f == 12) {
            char *nm; int dt, rank; int64_t dims[MAXR];
            parse_value_info(sub, &nm, &dt, dims, &rank);
            if (!nm) continue;
            int id = val_id(g, nm);
            if (f == 11) {
                if (!g->val[id].is_const && g->n_in < 8) {   /* initializers may be re-listed as inputs */
                    int k = g->n_in++;
                    g->in_ids[k] = id; g->in_rank[k] = rank; g->in_dt[k] = dt;
                    memcpy(g->in_dims[k], dims, sizeof dims);
                }
            } else if (g->n_out < 8) {
                int k = g->n_out++;
                g->out_ids[k] = id; g->out_rank[k] = rank; g->out_dt[k] = dt;
                memcpy(g->out_dims[k], dims, sizeof dims);
            }
            free(nm);
        }
    }
    return g;
}

void orc_graph_free(OrcGraph *g)
{
    if (!g) return;
    for (int i = 0; i < g->n_val; ++i) { free(g->vname[i]); if (g->val[i].owns) free(g->val[i].data); }
    for (int i = 0; i < g->n_nodes; ++i) {
        Node *n = &g->nodes[i];
        for (int a = 0; a < n->n_attrs; ++a) { free(n->attrs[a].name); free(n->attrs[a].ints); if (n->attrs[a].has_t) free(n->attrs[a].t.data); }
        free(n->attrs); free(n->op);
    }
    free(g->vname); free(g->val); free(g->nodes); free(g);
}

int orc_graph_num_inputs(const OrcGraph *g) { return g->n_in; }
int orc_graph_num_outputs(const OrcGraph *g) { return g->n_out; }
int orc_graph_input_dims(const OrcGraph *g, int i, int64_t *d, int max) { for (int k = 0; k < g->in_rank[i] && k < max; ++k) d[k] = g->in_dims[i][k]; return g->in_rank[i]; }
int orc_graph_output_dims(const OrcGraph *g, int i, int64_t *d, int max) { for (int k = 0; k < g->out_rank[i] && k < max; ++k) d[k] = g->out_dims[i][k]; return g->out_rank[i]; }
const char *orc_graph_input_name(const OrcGraph *g, int i) { return g->vname[g->in_ids[i]]; }
const char *orc_graph_output_name(const OrcGraph *g, int i) { return g->vname[g->out_ids[i]]; }

/* ---------------- kernels ---------------- */

/* out[n] = sum_k x[k] * W[k][n]; each output is one left-to-right chain in k */
__attribute__((target_clones("avx2", "default")))
static void mv_kn(const float *x, const float *W, float *out, size_t K, size_t N)
{
    for (size_t n = 0; n < N; ++n) out[n] = 0.0f;
    for (size_t k = 0; k < K; ++k) {
        const float xv = x[k];
        const float *w = W + k * N;
        for (size_t n = 0; n < N; ++n) out[n] += xv * w[n];
    }
}

/* out[n] = dot(x, W[n][:]) with 8 interleaved partial chains, combined pairwise */
__attribute__((target_clones("avx2", "default")))
static void mv_nk(const float *x, const float *W, float *out, size_t K, size_t N)
{
    for (size_t n = 0; n < N; ++n) {
        const float *w = W + n * K;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        size_t k = 0;
        for (; k + 8 <= K; k += 8)
            for (int j = 0; j < 8; ++j) acc[j] += x[k + j] * w[k + j];
        float tail = 0.0f;
        for (; k < K; ++k) tail += x[k] * w[k];
        out[n] = (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))) + tail;
    }
}

static int bcast_shape(const Ten *a, const Ten *b, int64_t *od, int *orank)
{
    int r = a->rank > b->rank ? a->rank : b->rank;
    for (int i = 0; i < r; ++i) {
        int ia = i - (r - a->rank), ib = i - (r - b->rank);
        int64_t da = ia >= 0 ? a->dims[ia] : 1, db = ib >= 0 ? b->dims[ib] : 1;
        if (da != db && da != 1 && db != 1) return -1;
        od[i] = da > db ? da : db;
    }
    *orank = r;
    return 0;
}

static int binary_op(const char *op, const Ten *a, const Ten *b, Ten *o)
{
    int64_t od[MAXR] = {0}; int r;
    if (bcast_shape(a, b, od, &r)) FAIL("%s: shapes not broadcastable", op);
    if (a->dtype != b->dtype) FAIL("%s: dtype mismatch", op);
    ten_alloc(o, a->dtype, r, od);
    size_t sa[MAXR], sb[MAXR];
    { size_t s = 1; for (int i = r - 1; i >= 0; --i) { int ia = i - (r - a->rank); int64_t d = ia >= 0 ? a->dims[ia] : 1; sa[i] = d == 1 ? 0 : s; s *= (size_t)d; } }
    { size_t s = 1; for (int i = r - 1; i >= 0; --i) { int ib = i - (r - b->rank); int64_t d = ib >= 0 ? b->dims[ib] : 1; sb[i] = d == 1 ? 0 : s; s *= (size_t)d; } }
    int64_t idx[MAXR] = {0};
    size_t oa = 0, ob = 0;
    const char c = op[0];
    for (size_t lin = 0; lin < o->n; ++lin) {
        if (a->dtype == DT_F32) {
            float x = ((float *)a->data)[oa], y = ((float *)b->data)[ob], z;
            switch (c) { case 'A': z = x + y; break; case 'S': z = x - y; break; case 'M': z = x * y; break; case 'D': z = x / y; break;
                         default: z = (y == 2.0f) ? x * x : (y == 0.5f ? sqrtf(x) : powf(x, y)); }
            ((float *)o->data)[lin] = z;
        } else {
            int64_t x = ((int64_t *)a->data)[oa], y = ((int64_t *)b->data)[ob], z;
            switch (c) { case 'A': z = x + y; break; case 'S': z = x - y; break; case 'M': z = x * y; break; case 'D': z = y ? x / y : 0; break; default: z = 0; }
            ((int64_t *)o->data)[lin] = z;
        }
        for (int i = r - 1; i >= 0; --i) {
            idx[i]++; oa += sa[i]; ob += sb[i];
            if (idx[i] < od[i]) break;
            oa -= sa[i] * (size_t)od[i]; ob -= sb[i] * (size_t)od[i]; idx[i] = 0;
        }
    }
    return 0;
}

static int unary_op(const char *op, const Ten *a, Ten *o)
{
    if (a->dtype != DT_F32) FAIL("%s on non-float", op);
    ten_alloc(o, DT_F32, a->rank, a->dims);
    const float *x = a->data; float *y = o->data;
    if (!strcmp(op, "Sigmoid")) for (size_t i = 0; i < o->n; ++i) y[i] = 1.0f / (1.0f + expf(-x[i]));
    else if (!strcmp(op, "Tanh")) for (size_t i = 0; i < o->n; ++i) y[i] = tanhf(x[i]);
    else if (!strcmp(op, "Relu")) for (size_t i = 0; i < o->n; ++i) y[i] = x[i] > 0 ? x[i] : 0;
    else if (!strcmp(op, "Exp")) for (size_t i = 0; i < o->n; ++i) y[i] = expf(x[i]);
    else if (!strcmp(op, "Sqrt")) for (size_t i = 0; i < o->n; ++i) y[i] = sqrtf(x[i]);
    else if (!strcmp(op, "Neg")) for (size_t i = 0; i < o->n; ++i) y[i] = -x[i];
    else FAIL("unary %s", op);
    return 0;
}

static int norm_axis(int64_t ax, int rank) { return (int)(ax < 0 ? ax + rank : ax); }

static int op_conv(const Node *nd, const Ten *x, const Ten *w, const Ten *bias, Ten *o)
{
    int sp = x->rank - 2;      /* spatial rank 1 or 2 */
    if (sp < 1 || sp > 2 || x->dims[0] != 1) FAIL("Conv: unsupported input rank/batch");
    int64_t group = attr_i(nd, "group", 1);
    int64_t st[2] = {1, 1}, pad[4] = {0, 0, 0, 0}, dil[2] = {1, 1};
    const Attr *a;
    if ((a = attr_find(nd, "strides"))) for (int i = 0; i < sp; ++i) st[i] = a->ints[i];
    if ((a = attr_find(nd, "dilations"))) for (int i = 0; i < sp; ++i) dil[i] = a->ints[i];
    if ((a = attr_find(nd, "pads"))) for (int i = 0; i < 2 * sp; ++i) pad[i] = a->ints[i];
    int64_t C = x->dims[1], O = w->dims[0], Cg = w->dims[1];
    int64_t H = sp == 2 ? x->dims[2] : 1, W = sp == 2 ? x->dims[3] : x->dims[2];
    int64_t kh = sp == 2 ? w->dims[2] : 1, kw = sp == 2 ? w->dims[3] : w->dims[2];
    int64_t sh = sp == 2 ? st[0] : 1, sw = sp == 2 ? st[1] : st[0];
    int64_t dh = sp == 2 ? dil[0] : 1, dw = sp == 2 ? dil[1] : dil[0];
    int64_t ph0 = sp == 2 ? pad[0] : 0, pw0 = sp == 2 ? pad[1] : pad[0];
    int64_t ph1 = sp == 2 ? pad[2] : 0, pw1 = sp == 2 ? pad[3] : pad[1];
    if (Cg * group != C || O % group) FAIL("Conv: group mismatch");
    int64_t OH = (H + ph0 + ph1 - dh * (kh - 1) - 1) / sh + 1;
    int64_t OW = (W + pw0 + pw1 - dw * (kw - 1) - 1) / sw + 1;
    int64_t od[4] = {1, O, OH, OW};
    if (sp == 1) { od[2] = OW; ten_alloc(o, DT_F32, 3, od); } else ten_alloc(o, DT_F32, 4, od);
    const float *X = x->data, *Wt = w->data; float *Y = o->data;
    const float *B = bias ? bias->data : NULL;
    int64_t Og = O / group;
    for (int64_t oc = 0; oc < O; ++oc) {
        int64_t g0 = (oc / Og) * Cg;
        for (int64_t oh = 0; oh < OH; ++oh) for (int64_t ow = 0; ow < OW; ++ow) {
            float acc = 0.0f;
            for (int64_t c = 0; c < Cg; ++c) for (int64_t i = 0; i < kh; ++i) for (int64_t j = 0; j < kw; ++j) {
                int64_t ih = oh * sh - ph0 + i * dh, iw = ow * sw - pw0 + j * dw;
                if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                acc += X[((g0 + c) * H + ih) * W + iw] * Wt[((oc * Cg + c) * kh + i) * kw + j];
            }
            if (B) acc += B[oc];
            Y[(oc * OH + oh) * OW + ow] = acc;
        }
    }
    return 0;
}

/* ---- fp16-operand mode (checker for the product's APRIL_PRECISION=f16, BASELINE configs[4]) ----
 * Every MatMul / Gemm rounds BOTH operands to IEEE binary16 (round to nearest even) and accumulates in fp32, which is
 * what v_mfma_f32_16x16x16_f16 computes up to summation order.  Conv nodes, biases and everything else stay fp32. */
static int g_f16_linear = 0;
void orc_set_f16_linear(int on) { g_f16_linear = on; }
int orc_get_f16_linear(void) { return g_f16_linear; }

/* float -> nearest binary16 (ties to even) -> float, by bit arithmetic (this gcc has no _Float16) */
static float f16_round1(float x)
{
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u; uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return x;                                  /* inf / nan */
    if (a < 0x38800000u) {                                           /* below 2^-14: binary16 subnormal, quantum 2^-24 */
        float m; memcpy(&m, &a, 4);
        m = rintf(m * 16777216.0f) * (1.0f / 16777216.0f);           /* default rounding mode: nearest even */
        memcpy(&a, &m, 4);
    } else {
        a += 0xfffu + ((a >> 13) & 1u);                              /* keep 10 mantissa bits */
        a &= ~0x1fffu;
        if (a >= 0x47800000u) a = 0x7f800000u;                       /* >= 65520 rounds to infinity */
    }
    a |= sign;
    float r; memcpy(&r, &a, 4);
    return r;
}
void orc_round_f16(const float *src, float *dst, size_t n)
{
    for (size_t i = 0; i < n; ++i) dst[i] = f16_round1(src[i]);
}
static const float *f16_operand(const Ten *t, float **tmp)
{
    *tmp = NULL;
    if (!g_f16_linear) return (const float *)t->data;
    Ten *m = (Ten *)t;
    if (t->is_const) {
        if (!m->h16) { m->h16 = (float *)malloc((t->n ? t->n : 1) * 4); orc_round_f16((const float *)t->data, m->h16, t->n); }
        return m->h16;
    }
    *tmp = (float *)malloc((t->n ? t->n : 1) * 4);
    orc_round_f16((const float *)t->data, *tmp, t->n);
    return *tmp;
}

static int op_matmul(const Ten *a, const Ten *b, Ten *o)
{
    if (b->rank != 2 || a->rank < 1) FAIL("MatMul: B must be 2-D");
    int64_t K = b->dims[0], N = b->dims[1];
    if (a->dims[a->rank - 1] != K) FAIL("MatMul: K mismatch");
    int64_t od[MAXR]; int r = a->rank;
    for (int i = 0; i < r - 1; ++i) od[i] = a->dims[i];
    od[r - 1] = N;
    ten_alloc(o, DT_F32, r, od);
    size_t M = a->n / (size_t)K;
    float *ta, *tb;
    const float *ad = f16_operand(a, &ta), *bd = f16_operand(b, &tb);
    for (size_t m = 0; m < M; ++m) mv_kn(ad + m * (size_t)K, bd, (float *)o->data + m * (size_t)N, (size_t)K, (size_t)N);
    free(ta); free(tb);
    return 0;
}

static int op_gemm(const Node *nd, const Ten *a, const Ten *b, const Ten *c, Ten *o)
{
    int tA = (int)attr_i(nd, "transA", 0), tB = (int)attr_i(nd, "transB", 0);
    float alpha = attr_f(nd, "alpha", 1.0f), beta = attr_f(nd, "beta", 1.0f);
    if (tA || a->rank != 2 || b->rank != 2) FAIL("Gemm: unsupported layout tA=%d ra=%d rb=%d", tA, a->rank, b->rank);
    int64_t M = a->dims[0], K = a->dims[1];
    int64_t N = tB ? b->dims[0] : b->dims[1];
    if ((tB ? b->dims[1] : b->dims[0]) != K) FAIL("Gemm: K mismatch");
    int64_t od[2] = {M, N};
    ten_alloc(o, DT_F32, 2, od);
    float *ta, *tb;
    const float *ad = f16_operand(a, &ta), *bd = f16_operand(b, &tb);
    for (int64_t m = 0; m < M; ++m) {
        float *y = (float *)o->data + m * N;
        if (tB) mv_nk(ad + m * K, bd, y, (size_t)K, (size_t)N);
        else    mv_kn(ad + m * K, bd, y, (size_t)K, (size_t)N);
        for (int64_t n = 0; n < N; ++n) {
            float v = alpha == 1.0f ? y[n] : alpha * y[n];
            if (c) {
                float cv = c->n == 1 ? ((float *)c->data)[0] : (c->n == (size_t)N ? ((float *)c->data)[n] : ((float *)c->data)[m * N + n]);
                v += beta == 1.0f ? cv : beta * cv;
            }
            y[n] = v;
        }
    }
    free(ta); free(tb);
    return 0;
}

static void copy_elems(Ten *o, size_t oi, const Ten *a, size_t ai, size_t cnt)
{
    size_t es = a->dtype == DT_F32 ? 4 : 8;
    memcpy((char *)o->data + oi * es, (const char *)a->data + ai * es, cnt * es);
}

static int exec_node(OrcGraph *g, Node *nd)
{
    Ten *in[64]; Ten *out[8];
    for (int i = 0; i < nd->n_in; ++i) {
        in[i] = nd->in[i] >= 0 ? &g->val[nd->in[i]] : NULL;
        if (in[i] && !in[i]->data) FAIL("%s: input '%s' not computed", nd->op, g->vname[nd->in[i]]);
    }
    for (int i = 0; i < nd->n_out; ++i) { out[i] = &g->val[nd->out[i]]; ten_release(out[i]); }
    const char *op = nd->op;
    Ten *o = out[0];

    if (!strcmp(op, "Add") || !strcmp(op, "Sub") || !strcmp(op, "Mul") || !strcmp(op, "Div") || !strcmp(op, "Pow"))
        return binary_op(op, in[0], in[1], o);
    if (!strcmp(op, "Sigmoid") || !strcmp(op, "Tanh") || !strcmp(op, "Relu") || !strcmp(op, "Exp") || !strcmp(op, "Sqrt") || !strcmp(op, "Neg"))
        return unary_op(op, in[0], o);
    if (!strcmp(op, "Conv")) return op_conv(nd, in[0], in[1], nd->n_in > 2 ? in[2] : NULL, o);
    if (!strcmp(op, "MatMul")) return op_matmul(in[0], in[1], o);
    if (!strcmp(op, "Gemm")) return op_gemm(nd, in[0], in[1], nd->n_in > 2 ? in[2] : NULL, o);
    if (!strcmp(op, "Identity") || !strcmp(op, "Dropout")) { ten_alloc(o, in[0]->dtype, in[0]->rank, in[0]->dims); copy_elems(o, 0, in[0], 0, o->n); return 0; }
    if (!strcmp(op, "Constant")) {
        const Attr *a = attr_find(nd, "value");
        if (a && a->has_t) { ten_alloc(o, a->t.dtype, a->t.rank, a->t.dims); copy_elems(o, 0, &a->t, 0, o->n); return 0; }
        if ((a = attr_find(nd, "value_float"))) { ten_alloc(o, DT_F32, 0, NULL); ((float *)o->data)[0] = a->f; return 0; }
        if ((a = attr_find(nd, "value_int"))) { ten_alloc(o, DT_I64, 0, NULL); ((int64_t *)o->data)[0] = a->i; return 0; }
        FAIL("Constant: unsupported value kind");
    }
    if (!strcmp(op, "ConstantOfShape")) {
        const Attr *a = attr_find(nd, "value");
        int64_t d[MAXR]; int r = (int)in[0]->n;
        for (int i = 0; i < r; ++i) d[i] = ((int64_t *)in[0]->data)[i];
        int dt = a && a->has_t ? a->t.dtype : DT_F32;
        ten_alloc(o, dt, r, d);
        if (a && a->has_t) for (size_t i = 0; i < o->n; ++i) copy_elems(o, i, &a->t, 0, 1);
        return 0;
    }
    if (!strcmp(op, "Shape")) { int64_t d[1] = {in[0]->rank}; ten_alloc(o, DT_I64, 1, d); for (int i = 0; i < in[0]->rank; ++i) ((int64_t *)o->data)[i] = in[0]->dims[i]; return 0; }
    if (!strcmp(op, "Cast")) {
        int64_t to = attr_i(nd, "to", 1);
        int dt = to == 1 ? DT_F32 : DT_I64;
        ten_alloc(o, dt, in[0]->rank, in[0]->dims);
        for (size_t i = 0; i < o->n; ++i) {
            double v = in[0]->dtype == DT_F32 ? ((float *)in[0]->data)[i] : (double)((int64_t *)in[0]->data)[i];
            if (dt == DT_F32) ((float *)o->data)[i] = (float)v; else ((int64_t *)o->data)[i] = (int64_t)v;
        }
        return 0;
    }
    if (!strcmp(op, "Reshape")) {
        int64_t d[MAXR]; int r = (int)in[1]->n; size_t known = 1; int neg = -1;
        if (r > MAXR) FAIL("Reshape rank");
        for (int i = 0; i < r; ++i) {
            d[i] = ((int64_t *)in[1]->data)[i];
            if (d[i] == 0) d[i] = in[0]->dims[i];
            if (d[i] < 0) neg = i; else known *= (size_t)d[i];
        }
        if (neg >= 0) d[neg] = (int64_t)(in[0]->n / (known ? known : 1));
        ten_alloc(o, in[0]->dtype, r, d);
        if (o->n != in[0]->n) FAIL("Reshape: element count");
        copy_elems(o, 0, in[0], 0, o->n);
        return 0;
    }
    if (!strcmp(op, "Squeeze") || !strcmp(op, "Unsqueeze")) {
        int64_t axes[MAXR]; int na = 0;
        const Attr *a = attr_find(nd, "axes");
        if (a) { na = a->n_ints; for (int i = 0; i < na; ++i) axes[i] = a->ints[i]; }
        else if (nd->n_in > 1 && in[1]) { na = (int)in[1]->n; for (int i = 0; i < na; ++i) axes[i] = ((int64_t *)in[1]->data)[i]; }
        int64_t d[MAXR]; int r = 0;
        if (op[0] == 'S') {
            for (int i = 0; i < in[0]->rank; ++i) {
                int drop = 0;
                if (na == 0) drop = in[0]->dims[i] == 1;
                for (int k = 0; k < na; ++k) if (norm_axis(axes[k], in[0]->rank) == i) drop = 1;
                if (!drop) d[r++] = in[0]->dims[i];
            }
        } else {
            int nr = in[0]->rank + na, src = 0;
            if (nr > MAXR) FAIL("Unsqueeze rank");
            for (int i = 0; i < nr; ++i) {
                int ins = 0;
                for (int k = 0; k < na; ++k) if (norm_axis(axes[k], nr) == i) ins = 1;
                d[i] = ins ? 1 : in[0]->dims[src++];
            }
            r = nr;
        }
        ten_alloc(o, in[0]->dtype, r, d);
        copy_elems(o, 0, in[0], 0, o->n);
        return 0;
    }
    if (!strcmp(op, "Transpose")) {
        const Attr *a = attr_find(nd, "perm");
        int r = in[0]->rank; int perm[MAXR];
        for (int i = 0; i < r; ++i) perm[i] = a ? (int)a->ints[i] : r - 1 - i;
        int64_t d[MAXR]; for (int i = 0; i < r; ++i) d[i] = in[0]->dims[perm[i]];
        ten_alloc(o, in[0]->dtype, r, d);
        size_t is[MAXR]; { size_t s = 1; for (int i = r - 1; i >= 0; --i) { is[i] = s; s *= (size_t)in[0]->dims[i]; } }
        int64_t idx[MAXR] = {0};
        for (size_t lin = 0; lin < o->n; ++lin) {
            size_t src = 0; for (int i = 0; i < r; ++i) src += (size_t)idx[i] * is[perm[i]];
            copy_elems(o, lin, in[0], src, 1);
            for (int i = r - 1; i >= 0; --i) { if (++idx[i] < d[i]) break; idx[i] = 0; }
        }
        return 0;
    }
    if (!strcmp(op, "Concat")) {
        int ax = norm_axis(attr_i(nd, "axis", 0), in[0]->rank);
        int64_t d[MAXR]; int r = in[0]->rank; memcpy(d, in[0]->dims, sizeof d);
        d[ax] = 0; for (int i = 0; i < nd->n_in; ++i) d[ax] += in[i]->dims[ax];
        ten_alloc(o, in[0]->dtype, r, d);
        size_t outer = 1, inner = 1;
        for (int i = 0; i < ax; ++i) outer *= (size_t)d[i];
        for (int i = ax + 1; i < r; ++i) inner *= (size_t)d[i];
        size_t off = 0;
        for (int i = 0; i < nd->n_in; ++i) {
            size_t span = (size_t)in[i]->dims[ax] * inner;
            for (size_t ot = 0; ot < outer; ++ot) copy_elems(o, ot * (size_t)d[ax] * inner + off, in[i], ot * span, span);
            off += span;
        }
        return 0;
    }
    if (!strcmp(op, "Split")) {
        int ax = norm_axis(attr_i(nd, "axis", 0), in[0]->rank);
        const Attr *a = attr_find(nd, "split");
        size_t outer = 1, inner = 1;
        for (int i = 0; i < ax; ++i) outer *= (size_t)in[0]->dims[i];
        for (int i = ax + 1; i < in[0]->rank; ++i) inner *= (size_t)in[0]->dims[i];
        int64_t pos = 0;
        for (int k = 0; k < nd->n_out; ++k) {
            int64_t len = a ? a->ints[k] : in[0]->dims[ax] / nd->n_out;
            int64_t d[MAXR]; memcpy(d, in[0]->dims, sizeof d); d[ax] = len;
            ten_alloc(out[k], in[0]->dtype, in[0]->rank, d);
            for (size_t ot = 0; ot < outer; ++ot)
                copy_elems(out[k], ot * (size_t)len * inner, in[0], (ot * (size_t)in[0]->dims[ax] + (size_t)pos) * inner, (size_t)len * inner);
            pos += len;
        }
        return 0;
    }
    if (!strcmp(op, "Slice")) {
        int r = in[0]->rank;
        int64_t st[MAXR], en[MAXR];
        for (int i = 0; i < r; ++i) { st[i] = 0; en[i] = in[0]->dims[i]; }
        int ns; const int64_t *S, *E, *A = NULL, *P = NULL;
        const Attr *as = attr_find(nd, "starts");
        if (as) { ns = as->n_ints; S = as->ints; E = attr_find(nd, "ends")->ints; const Attr *aa = attr_find(nd, "axes"); A = aa ? aa->ints : NULL; }
        else { ns = (int)in[1]->n; S = in[1]->data; E = in[2]->data; if (nd->n_in > 3 && in[3]) A = in[3]->data; if (nd->n_in > 4 && in[4]) P = in[4]->data; }
        for (int k = 0; k < ns; ++k) {
            int ax = A ? norm_axis(A[k], r) : k;
            int64_t dim = in[0]->dims[ax], s = S[k], e = E[k], p = P ? P[k] : 1;
            if (p != 1) FAIL("Slice: steps != 1");
            if (s < 0) s += dim;
            if (e < 0) e += dim;
            if (s < 0) s = 0;
            if (s > dim) s = dim;
            if (e < 0) e = 0;
            if (e > dim) e = dim;
            st[ax] = s; en[ax] = e > s ? e : s;
        }
        int64_t d[MAXR]; for (int i = 0; i < r; ++i) d[i] = en[i] - st[i];
        ten_alloc(o, in[0]->dtype, r, d);
        size_t is[MAXR]; { size_t s = 1; for (int i = r - 1; i >= 0; --i) { is[i] = s; s *= (size_t)in[0]->dims[i]; } }
        int64_t idx[MAXR] = {0};
        for (size_t lin = 0; lin < o->n; ++lin) {
            size_t src = 0; for (int i = 0; i < r; ++i) src += (size_t)(st[i] + idx[i]) * is[i];
            copy_elems(o, lin, in[0], src, 1);
            for (int i = r - 1; i >= 0; --i) { if (++idx[i] < d[i]) break; idx[i] = 0; }
        }
        return 0;
    }
    if (!strcmp(op, "Gather")) {
        int ax = norm_axis(attr_i(nd, "axis", 0), in[0]->rank);
        const Ten *dat = in[0], *ix = in[1];
        if (ix->dtype != DT_I64) FAIL("Gather: indices must be int64");
        int64_t d[MAXR]; int r = 0;
        for (int i = 0; i < ax; ++i) d[r++] = dat->dims[i];
        for (int i = 0; i < ix->rank; ++i) d[r++] = ix->dims[i];
        for (int i = ax + 1; i < dat->rank; ++i) d[r++] = dat->dims[i];
        if (r > MAXR) FAIL("Gather rank");
        ten_alloc(o, dat->dtype, r, d);
        size_t outer = 1, inner = 1;
        for (int i = 0; i < ax; ++i) outer *= (size_t)dat->dims[i];
        for (int i = ax + 1; i < dat->rank; ++i) inner *= (size_t)dat->dims[i];
        size_t ni = ix->n;
        for (size_t ot = 0; ot < outer; ++ot) for (size_t k = 0; k < ni; ++k) {
            int64_t j = ((int64_t *)ix->data)[k];
            if (j < 0) j += dat->dims[ax];
            if (j < 0 || j >= dat->dims[ax]) FAIL("Gather: index out of range");
            copy_elems(o, (ot * ni + k) * inner, dat, (ot * (size_t)dat->dims[ax] + (size_t)j) * inner, inner);
        }
        return 0;
    }
    if (!strcmp(op, "ReduceMean")) {
        const Attr *a = attr_find(nd, "axes");
        if (!a || a->n_ints != 1) FAIL("ReduceMean: exactly one axis supported");
        int ax = norm_axis(a->ints[0], in[0]->rank);
        int keep = (int)attr_i(nd, "keepdims", 1);
        int64_t d[MAXR]; int r = 0;
        for (int i = 0; i < in[0]->rank; ++i) { if (i == ax) { if (keep) d[r++] = 1; } else d[r++] = in[0]->dims[i]; }
        ten_alloc(o, DT_F32, r, d);
        size_t outer = 1, inner = 1, len = (size_t)in[0]->dims[ax];
        for (int i = 0; i < ax; ++i) outer *= (size_t)in[0]->dims[i];
        for (int i = ax + 1; i < in[0]->rank; ++i) inner *= (size_t)in[0]->dims[i];
        const float *x = in[0]->data; float *y = o->data;
        for (size_t ot = 0; ot < outer; ++ot) for (size_t q = 0; q < inner; ++q) {
            float s = 0.0f;
            for (size_t k = 0; k < len; ++k) s += x[(ot * len + k) * inner + q];
            y[ot * inner + q] = s / (float)len;
        }
        return 0;
    }
    FAIL("unsupported op '%s'", op);
}

static int fold_constants(OrcGraph *g)
{
    for (int i = 0; i < g->n_nodes; ++i) {
        Node *nd = &g->nodes[i];
        int all = 1;
        for (int k = 0; k < nd->n_in; ++k) if (nd->in[k] >= 0 && !g->val[nd->in[k]].is_const) all = 0;
        if (all) {
            if (exec_node(g, nd)) return -1;
            for (int k = 0; k < nd->n_out; ++k) g->val[nd->out[k]].is_const = 1;
            nd->is_const = 1;
        }
    }
    g->folded = 1;
    return 0;
}

int orc_graph_run(OrcGraph *g, int n_in, const char *const *in_names, const void *const *in_bufs,
                  int n_out, const char *const *out_names, void *const *out_bufs)
{
    if (!g->folded && fold_constants(g)) return -1;
    for (int i = 0; i < n_in; ++i) {
        int k = -1;
        for (int j = 0; j < g->n_in; ++j) if (!strcmp(g->vname[g->in_ids[j]], in_names[i])) k = j;
        if (k < 0) FAIL("no graph input named '%s'", in_names[i]);
        Ten *t = &g->val[g->in_ids[k]];
        ten_release(t);
        t->dtype = g->in_dt[k] == DT_F32 ? DT_F32 : DT_I64;
        t->rank = g->in_rank[k];
        memcpy(t->dims, g->in_dims[k], sizeof t->dims);
        t->n = numel(t);
        t->data = (void *)in_bufs[i]; t->owns = 0; t->live = 1;
    }
    for (int i = 0; i < g->n_nodes; ++i) {
        if (g->nodes[i].is_const) continue;
        if (exec_node(g, &g->nodes[i])) return -1;
    }
    for (int i = 0; i < n_out; ++i) {
        int k = -1;
        for (int j = 0; j < g->n_out; ++j) if (!strcmp(g->vname[g->out_ids[j]], out_names[i])) k = j;
        if (k < 0) FAIL("no graph output named '%s'", out_names[i]);
        Ten *t = &g->val[g->out_ids[k]];
        if (!t->data) FAIL("output '%s' not produced", out_names[i]);
        memcpy(out_bufs[i], t->data, t->n * (t->dtype == DT_F32 ? 4 : 8));
    }
    /* release the per-run intermediates (constants stay) */
    for (int i = 0; i < g->n_val; ++i) if (!g->val[i].is_const) ten_release(&g->val[i]);
    return 0;
}
