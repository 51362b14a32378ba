// See model_loader.h.
//
// Container layout: reference extra/file-format.md; validation rules restate
// src/file/model_file.c:58-129 and src/params.c:46-112.  Unlike the reference
// (model_file.c:164-166 reads PARAMS from the current FILE position) the params
// block is located through its header entry.
//
// Weight extraction is STRUCTURAL: torch.onnx names most MatMul weights
// "onnx::MatMul_###", so nothing is looked up by name.  The walker lists the
// weight-bearing nodes (Conv / MatMul / Gemm with a constant operand) in graph
// order and checks the surrounding operators (DoubleSwish constant, BasicNorm
// epsilon, the ROLE of every gate -- which product meets the previous cell state --,
// state lineage of the recurrent matmul) against what the HIP kernels implement;
// anything else is rejected with a message that names the node it stopped at.
// Spellings an exporter / constant folder may choose freely are accepted alike and give the
// same packed weights (tests/test_loader.py): Gemm(transB) or MatMul(+Add), biases on either
// or both gate products or added after their sum, Split or four Slices (any gate order),
// constants as initializers / Constant nodes / behind Identity, Cast, Transpose, and
// Identity / Cast / Dropout nodes anywhere on the activation path.
#include "model_loader.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include "common.h"
#include "onnx_reader.h"

namespace aprilx {

size_t HostModel::param_count() const
{
    size_t n = 0;
    for (int i = 0; i < 3; ++i) n += conv_w[i].size() + conv_b[i].size();
    n += w_embed.size() + b_embed.size();
    for (auto &l : layers) n += l.w_gates.size() + 2 * l.b_gates.size() + l.w_hr.size() + l.w_ff1.size() + l.b_ff1.size() + l.w_ff2.size() + l.b_ff2.size() + 1;
    n += w_encproj.size() + b_encproj.size() + emb.size() + dec_conv.size() + dec_conv_b.size() + w_decproj.size() + b_decproj.size() + w_out.size() + b_out.size();
    return n;
}

// ---------------------------------------------------------------- container
namespace {
struct Rd {
    const uint8_t *p; size_t n, pos = 0; bool bad = false;
    uint64_t le(int bytes) {
        if (pos + (size_t)bytes > n) { bad = true; pos = n; return 0; }
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v |= (uint64_t)p[pos + i] << (8 * i);
        pos += (size_t)bytes;
        return v;
    }
    int32_t i32() { return (int32_t)(uint32_t)le(4); }
    bool str(std::string &s) {
        uint64_t len = le(8);
        if (bad || len > n - pos) { bad = true; return false; }
        s.assign((const char *)p + pos, (size_t)len);
        pos += (size_t)len;
        return true;
    }
};

bool parse_params(Rd &r, ModelParams &o, std::string &err)
{
    static const char magic[8] = {'P', 'A', 'R', 'A', 'M', 'S', 0, 0};
    if (r.pos > r.n || r.n - r.pos < 8 || memcmp(r.p + r.pos, magic, 8) != 0) { err = "params: bad magic"; return false; }
    r.pos += 8;
    o.batch_size = r.i32(); o.segment_size = r.i32(); o.segment_step = r.i32(); o.mel_features = r.i32();
    o.sample_rate = r.i32(); o.frame_shift_ms = r.i32(); o.frame_length_ms = r.i32(); o.round_pow2 = r.i32() != 0;
    o.mel_low = r.i32(); o.mel_high = r.i32(); o.snip_edges = r.i32() != 0; o.token_count = r.i32(); o.blank_id = r.i32();
    if (r.bad) { err = "params: truncated"; return false; }
    if (!validate_params(o, err)) return false;
    const size_t start = r.pos;
    size_t longest = 0;
    for (int i = 0; i < o.token_count; ++i) {
        int32_t len = r.i32();
        if (r.bad || len < 0 || (size_t)len > r.n - r.pos) { err = "params: token table truncated"; return false; }
        if ((size_t)len > longest) longest = (size_t)len;
        r.pos += (size_t)len;
    }
    o.token_stride = longest + 1;
    o.tokens.assign((size_t)o.token_count * o.token_stride, 0);
    r.pos = start;
    for (int i = 0; i < o.token_count; ++i) {
        int32_t len = r.i32();
        memcpy(o.tokens.data() + o.token_stride * (size_t)i, r.p + r.pos, (size_t)len);
        r.pos += (size_t)len;
    }
    return true;
}
}  // namespace

// Range checks of src/params.c:71-82, shared by the .april reader and the packed-blob reader (a blob is as untrusted
// as a model file: these values size the feature ring, drive the chunk loop and index the token table).
bool validate_params(const ModelParams &o, std::string &err)
{
#define PCHECK(c) if (!(c)) { err = "params: check failed: " #c; return false; }
    PCHECK(o.batch_size == 1);
    PCHECK(o.segment_size > 0 && o.segment_size < 100);
    PCHECK(o.segment_step > 0 && o.segment_step < 100 && o.segment_step <= o.segment_size);
    PCHECK(o.mel_features > 0 && o.mel_features < 256);
    PCHECK(o.sample_rate > 0 && o.sample_rate < 144000);
    PCHECK(o.token_count > 0 && o.token_count < 16384);
    PCHECK(o.blank_id >= 0 && o.blank_id < o.token_count);
    PCHECK(o.frame_shift_ms > 0 && o.frame_shift_ms <= o.frame_length_ms);
    PCHECK(o.frame_length_ms > 0 && o.frame_length_ms <= 5000);
    PCHECK(o.mel_low > 0 && o.mel_low < o.sample_rate);
    PCHECK(o.mel_high == 0 || o.mel_high > o.mel_low);
    // the frame shift in samples must be at least one (sample_rate * shift_ms / 1000), else framing never advances
    PCHECK((long)o.sample_rate * o.frame_shift_ms >= 1000);
#undef PCHECK
    return true;
}

bool parse_container(const std::vector<uint8_t> &blob, ContainerInfo &info, std::string &err)
{
    Rd r{blob.data(), blob.size()};
    if (r.n < 20 || memcmp(r.p, "APRILMDL", 8) != 0) { err = "not an .april file (magic)"; return false; }
    r.pos = 8;
    uint32_t version = (uint32_t)r.le(4);
    if (version != 1) { err = "unsupported .april version " + std::to_string(version); return false; }
    (void)r.le(8);   // header_size
    if (r.pos + 8 > r.n) { err = "truncated header"; return false; }
    char lang[9]; memcpy(lang, r.p + r.pos, 8); lang[8] = 0; r.pos += 8;
    info.language = lang;
    if (!r.str(info.name) || !r.str(info.description)) { err = "truncated header strings"; return false; }
    info.model_type = (uint32_t)r.le(4);
    if (!(info.model_type > 0 && info.model_type < 2)) { err = "unexpected model type " + std::to_string(info.model_type); return false; }
    info.params_off = r.le(8); info.params_size = r.le(8);
    if (r.bad || info.params_off > r.n || info.params_size > r.n - info.params_off) { err = "params out of bounds of file"; return false; }
    uint64_t nn = r.le(8);
    if (r.bad || nn > 8) { err = "too many networks"; return false; }
    for (uint64_t i = 0; i < nn; ++i) {
        uint64_t off = r.le(8), sz = r.le(8);
        if (r.bad || off > r.n || sz > r.n - off) { err = "network " + std::to_string(i) + " out of bounds of file"; return false; }
        info.net_off.push_back(off); info.net_size.push_back(sz);
    }
    Rd pr{blob.data(), blob.size(), (size_t)info.params_off};
    return parse_params(pr, info.params, err);
}

// ---------------------------------------------------------------- graph view
namespace {

struct Linear {
    int K = 0, N = 0;
    std::vector<float> W;      // K x N
    std::vector<float> b;      // N or empty
    std::string in_value, out_value;
    int node = -1;
};

struct View {
    const OGraph &g;
    std::string err;
    explicit View(const OGraph &g_) : g(g_) {}
    // i-th input name of a node, or the empty string (no value is named "") when the node has fewer inputs:
    // a malformed graph is rejected by the pattern checks instead of indexing past the input list
    static const std::string &arg(const ONode &n, size_t i) { static const std::string none; return i < n.in.size() ? n.in[i] : none; }
    static const std::string &res(const ONode &n, size_t i) { static const std::string none; return i < n.out.size() ? n.out[i] : none; }

    const ONode *producer(const std::string &v) const {
        auto it = g.producer.find(v);
        return it == g.producer.end() ? nullptr : &g.nodes[it->second];
    }
    static bool pass_through(const ONode &n) {      // value-preserving at inference time
        return (n.op == "Identity" || n.op == "Dropout" || n.op == "Cast") && !n.in.empty() && !n.out.empty();
    }
    // consumers of a value, looking through Identity / Dropout / Cast nodes
    std::vector<const ONode *> consumers(const std::string &v, int depth = 0) const {
        std::vector<const ONode *> r;
        auto range = g.consumers.equal_range(v);
        for (auto it = range.first; it != range.second; ++it) {
            const ONode *n = &g.nodes[it->second];
            if (pass_through(*n) && depth < 8) { for (const ONode *c : consumers(n->out[0], depth + 1)) r.push_back(c); }
            else r.push_back(n);
        }
        return r;
    }
    // the value a node input really carries (producer chain of Identity / Dropout / Cast skipped)
    std::string source(std::string v) const {
        for (int guard = 0; guard < 16; ++guard) {
            const ONode *p = producer(v);
            if (!p || !pass_through(*p)) return v;
            v = p->in[0];
        }
        return v;
    }
    bool same(const std::string &a, const std::string &b) const { return !a.empty() && source(a) == source(b); }
    static std::string describe(const ONode &n) {
        std::string s = n.op + " '" + n.name + "' (inputs:";
        for (auto &i : n.in) s += " " + i;
        s += "; outputs:";
        for (auto &o : n.out) s += " " + o;
        return s + ")";
    }
    // a constant float tensor WITH its shape: initializer, Constant node, or one of those behind Identity / Cast /
    // a 2-D Transpose (constant folding may or may not have removed the transpose of a Linear weight)
    bool const_tensor(const std::string &v, OTensor &out, int depth = 0) const {
        if (depth > 8 || v.empty()) return false;
        auto it = g.inits.find(v);
        if (it != g.inits.end()) { if (it->second.dtype != 1) return false; out = it->second; return true; }
        const ONode *p = producer(v);
        if (!p) return false;
        if (p->op == "Constant") {
            const OAttr *a = p->attr("value");
            if (a && a->has_t && a->t.dtype == 1) { out = a->t; return true; }
            return false;
        }
        if (pass_through(*p)) return const_tensor(p->in[0], out, depth + 1);
        if (p->op == "Transpose" && !p->in.empty()) {
            OTensor t;
            if (!const_tensor(p->in[0], t, depth + 1) || t.dims.size() != 2) return false;
            const OAttr *pa = p->attr("perm");
            if (pa && !(pa->ints.size() == 2 && pa->ints[0] == 1 && pa->ints[1] == 0)) return false;
            out.dtype = 1; out.dims = {t.dims[1], t.dims[0]}; out.f.resize(t.f.size());
            for (int64_t r = 0; r < t.dims[0]; ++r) for (int64_t c = 0; c < t.dims[1]; ++c) out.f[(size_t)(c * t.dims[0] + r)] = t.f[(size_t)(r * t.dims[1] + c)];
            return true;
        }
        return false;
    }
    // constant tensor behind a value name (initializer, Constant node, or a constant pushed
    // through Identity/Cast/Unsqueeze/Squeeze/Reshape/Exp)
    bool const_floats(const std::string &v, std::vector<float> &out, std::vector<int64_t> *dims = nullptr, int depth = 0) const {
        if (depth > 8) return false;
        auto it = g.inits.find(v);
        if (it != g.inits.end()) {
            if (it->second.dtype == 1) out = it->second.f;
            else { out.clear(); for (auto x : it->second.i) out.push_back((float)x); }
            if (dims) *dims = it->second.dims;
            return true;
        }
        const ONode *p = producer(v);
        if (!p) return false;
        if (p->op == "Constant") {
            const OAttr *a = p->attr("value");
            if (a && a->has_t) {
                if (a->t.dtype == 1) out = a->t.f;
                else { out.clear(); for (auto x : a->t.i) out.push_back((float)x); }
                if (dims) *dims = a->t.dims;
                return true;
            }
            if ((a = p->attr("value_float"))) { out = {a->f}; if (dims) dims->clear(); return true; }
            return false;
        }
        if (p->op == "Identity" || p->op == "Cast" || p->op == "Dropout" || p->op == "Unsqueeze" || p->op == "Squeeze" || p->op == "Reshape")
            return const_floats(arg(*p, 0), out, dims, depth + 1);
        if (p->op == "Exp") {
            if (!const_floats(arg(*p, 0), out, dims, depth + 1)) return false;
            for (auto &x : out) x = expf(x);
            return true;
        }
        return false;
    }
    bool is_const(const std::string &v) const { std::vector<float> t; return const_floats(v, t); }

    // Value of a small integer expression: slice bounds as torch.onnx writes them for Tensor.chunk() with static shapes that it
    // does not fold -- Shape -> Gather -> (Add, Div, Mul with constants).  `dim_of(tensor, axis, out)` supplies the static
    // dimensions the caller knows (the width of the gate pre-activations, the number of layers of a state tensor).
    typedef std::function<bool(const std::string &, long, long &)> DimOf;
    bool const_index(const std::string &v, const DimOf &dim_of, long &out, int depth = 0) const {
        if (depth > 32) return false;
        std::vector<float> c;
        if (const_floats(v, c)) { if (c.size() != 1) return false; out = (long)c[0]; return true; }
        const ONode *p = producer(v);
        if (!p) return false;
        const std::string &op = p->op;
        if (op == "Identity" || op == "Cast" || op == "Unsqueeze" || op == "Squeeze" || op == "Reshape" || (op == "Concat" && p->in.size() == 1))
            return const_index(arg(*p, 0), dim_of, out, depth + 1);
        if (op == "Add" || op == "Sub" || op == "Mul" || op == "Div") {
            long a = 0, b = 0;
            if (p->in.size() != 2 || !const_index(p->in[0], dim_of, a, depth + 1) || !const_index(p->in[1], dim_of, b, depth + 1)) return false;
            if (op == "Add") out = a + b; else if (op == "Sub") out = a - b; else if (op == "Mul") out = a * b;
            else { if (b == 0) return false; out = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) --out; }      // ONNX integer Div truncates; chunk sizes are non-negative
            return true;
        }
        if (op == "Gather" && p->in.size() == 2) {
            const ONode *sh = producer(p->in[0]);
            long axis = 0;
            std::vector<float> idx;
            if (!sh || sh->op != "Shape" || sh->in.empty() || !const_floats(p->in[1], idx) || idx.size() != 1) return false;
            axis = (long)idx[0];
            return dim_of(sh->in[0], axis, out);
        }
        return false;
    }

    // follow layout-only operators upwards to the value that carries the data
    std::string lineage(std::string v) const {
        for (int guard = 0; guard < 64; ++guard) {
            const ONode *p = producer(v);
            if (!p) return v;
            const std::string &op = p->op;
            if (op == "Unsqueeze" || op == "Squeeze" || op == "Reshape" || op == "Transpose" || op == "Identity" || op == "Dropout" ||
                op == "Cast" || op == "Slice" || op == "Flatten" || (op == "Concat" && p->in.size() == 1) ||
                (op == "Gather" && p->in.size() == 2 && is_const(p->in[1])) || (op == "Split" && p->out.size() == 1))
                v = arg(*p, 0);
            else return v;
        }
        return v;
    }

    // follow layout-only operators DOWNWARDS while the value has exactly one consumer (unbind of a length-1 time axis, squeezes,
    // reshapes between a product and the operator that uses it)
    std::string downstream(std::string v) const {
        for (int guard = 0; guard < 16; ++guard) {
            auto cs = consumers(v);
            if (cs.size() != 1 || cs[0]->out.empty()) return v;
            const ONode &c = *cs[0];
            const std::string &op = c.op;
            const bool layout = op == "Reshape" || op == "Squeeze" || op == "Unsqueeze" || op == "Identity" || op == "Flatten" || op == "Cast" || op == "Dropout" ||
                                (op == "Split" && c.out.size() == 1) || (op == "Concat" && c.in.size() == 1) ||
                                (op == "Gather" && c.in.size() == 2 && same(c.in[0], v) && is_const(c.in[1]));
            if (!layout || !same(arg(c, 0), v)) return v;
            v = c.out[0];
        }
        return v;
    }

    bool weighted(const ONode &n) const {
        OTensor t;
        if (n.op == "Conv") return n.in.size() >= 2 && const_tensor(n.in[1], t);
        if (n.op == "MatMul") return n.in.size() == 2 && const_tensor(n.in[1], t) && t.dims.size() == 2;
        if (n.op == "Gemm") return n.in.size() >= 2 && const_tensor(n.in[1], t) && t.dims.size() == 2;
        return false;
    }

    bool linear_at(int idx, Linear &L) {
        const ONode &n = g.nodes[idx];
        L.node = idx;
        if (n.in.size() < 2 || n.out.empty()) { err = "linear node without operands (" + n.name + ")"; return false; }
        L.in_value = n.in[0];
        OTensor w;
        if (!const_tensor(n.in[1], w) || w.dims.size() != 2) { err = "linear weight must be a constant 2-D float tensor: " + describe(n); return false; }
        if (n.op == "MatMul") {
            L.K = (int)w.dims[0]; L.N = (int)w.dims[1];
            L.W = w.f;
            L.out_value = n.out[0];
            for (const ONode *c : consumers(n.out[0])) {
                if (c->op != "Add" || c->in.size() != 2) continue;
                const std::string &other = same(c->in[0], n.out[0]) ? c->in[1] : c->in[0];
                std::vector<float> b;
                if (const_floats(other, b) && (int)b.size() == L.N) { L.b = b; L.out_value = c->out[0]; break; }
            }
            return true;
        }
        if (n.op == "Gemm") {
            const bool tB = n.attr_i("transB", 0) != 0;
            if (n.attr_i("transA", 0) != 0) { err = "Gemm transA unsupported: " + describe(n); return false; }
            const OAttr *al = n.attr("alpha"), *be = n.attr("beta");
            if ((al && al->f != 1.0f) || (be && be->f != 1.0f)) { err = "Gemm alpha/beta != 1 unsupported: " + describe(n); return false; }
            L.K = (int)(tB ? w.dims[1] : w.dims[0]);
            L.N = (int)(tB ? w.dims[0] : w.dims[1]);
            L.W.resize((size_t)L.K * L.N);
            if (tB) { for (int nn = 0; nn < L.N; ++nn) for (int k = 0; k < L.K; ++k) L.W[(size_t)k * L.N + nn] = w.f[(size_t)nn * L.K + k]; }
            else L.W = w.f;
            if (n.in.size() > 2 && !n.in[2].empty()) {
                std::vector<float> b;
                if (!const_floats(n.in[2], b) || (int)b.size() != L.N) { err = "Gemm bias must be a constant of length N: " + describe(n); return false; }
                L.b = b;
            }
            L.out_value = n.out[0];
            if (L.b.empty()) {            // Gemm without C, bias added by a separate node
                for (const ONode *c : consumers(n.out[0])) {
                    if (c->op != "Add" || c->in.size() != 2) continue;
                    const std::string &other = same(c->in[0], n.out[0]) ? c->in[1] : c->in[0];
                    std::vector<float> b;
                    if (const_floats(other, b) && (int)b.size() == L.N) { L.b = b; L.out_value = c->out[0]; break; }
                }
            }
            return true;
        }
        err = "not a linear node";
        return false;
    }

    // y -> y * sigmoid(y - 1); returns the product's value name
    bool double_swish(const std::string &y, std::string &out) {
        for (const ONode *c : consumers(y)) {
            float shift = 0; bool ok = false;
            std::vector<float> k;
            if (c->op == "Sub" && c->in.size() == 2 && same(c->in[0], y) && const_floats(c->in[1], k) && k.size() == 1) { shift = k[0]; ok = true; }
            if (c->op == "Add" && c->in.size() == 2) {
                const std::string &o = same(c->in[0], y) ? c->in[1] : c->in[0];
                if (const_floats(o, k) && k.size() == 1) { shift = -k[0]; ok = true; }
            }
            if (!ok) continue;
            if (fabsf(shift - 1.0f) > 1e-6f) { err = "activation is x*sigmoid(x-c) with c != 1 at " + describe(*c); return false; }
            for (const ONode *s : consumers(res(*c, 0))) if (s->op == "Sigmoid")
                for (const ONode *m : consumers(res(*s, 0))) if (m->op == "Mul" && m->in.size() == 2 && !m->out.empty() && (same(m->in[0], y) || same(m->in[1], y))) { out = m->out[0]; return true; }
        }
        err = "expected DoubleSwish (x * sigmoid(x - 1)) after '" + y + "'";
        for (const ONode *c : consumers(y)) { err += "; found " + describe(*c); break; }
        return false;
    }

    // y -> y * (mean(y^2) + eps)^-0.5 ; returns eps and the output value.  Spellings of the square: Pow(y, 2) or Mul(y, y);
    // of the scale: Pow(v, -0.5), Reciprocal(Sqrt(v)), Div(1, Sqrt(v)) followed by Mul(y, .), or Div(y, Sqrt(v)).
    bool basic_norm(const std::string &y, float &eps, std::string &out) {
        for (const ONode *c : consumers(y)) {
            bool sq = false;
            std::vector<float> k;
            if (c->op == "Pow" && c->in.size() == 2 && same(c->in[0], y) && const_floats(c->in[1], k) && k.size() == 1 && k[0] == 2.0f) sq = true;
            if (c->op == "Mul" && c->in.size() == 2 && same(c->in[0], y) && same(c->in[1], y)) sq = true;
            if (!sq) continue;
            for (const ONode *rm : consumers(res(*c, 0))) if (rm->op == "ReduceMean")
                for (const ONode *ad : consumers(res(*rm, 0))) if (ad->op == "Add" && ad->in.size() == 2 && !ad->out.empty()) {
                    const std::string &o = same(ad->in[0], rm->out[0]) ? ad->in[1] : ad->in[0];
                    std::vector<float> e;
                    if (!const_floats(o, e) || e.size() != 1) continue;
                    auto mul_with_y = [&](const std::string &scale) {
                        for (const ONode *m : consumers(scale)) if (m->op == "Mul" && m->in.size() == 2 && !m->out.empty() && (same(m->in[0], y) || same(m->in[1], y))) { eps = e[0]; out = m->out[0]; return true; }
                        return false;
                    };
                    for (const ONode *pw : consumers(ad->out[0])) {
                        std::vector<float> ex;
                        if (pw->op == "Pow" && const_floats(arg(*pw, 1), ex) && ex.size() == 1 && ex[0] == -0.5f && mul_with_y(res(*pw, 0))) return true;
                        if (pw->op == "Sqrt") {
                            for (const ONode *r : consumers(res(*pw, 0))) {
                                if (r->op == "Reciprocal" && mul_with_y(res(*r, 0))) return true;
                                if (r->op == "Div" && r->in.size() == 2 && !r->out.empty() && same(r->in[1], pw->out[0])) {
                                    if (same(r->in[0], y)) { eps = e[0]; out = r->out[0]; return true; }
                                    std::vector<float> one;
                                    if (const_floats(r->in[0], one) && one.size() == 1 && one[0] == 1.0f && mul_with_y(r->out[0])) return true;
                                }
                            }
                        }
                    }
                }
        }
        err = "expected BasicNorm (x * (mean(x^2)+eps)^-0.5) after '" + y + "'";
        for (const ONode *c : consumers(y)) { err += "; found " + describe(*c); break; }
        return false;
    }
};

bool fail(std::string &err, const std::string &m) { err = m; return false; }

bool extract_encoder(const OGraph &g, const ModelParams &P, HostModel &M, std::string &err)
{
    View v(g);
    if (g.inputs.size() != 3 || g.outputs.size() != 3) return fail(err, "encoder must have 3 inputs and 3 outputs");
    const auto &xi = g.inputs[0], &hi = g.inputs[1], &ci = g.inputs[2];
    if (xi.dims.size() != 3 || hi.dims.size() != 3 || ci.dims.size() != 3) return fail(err, "encoder inputs must be rank 3");
    NetDims &D = M.dims;
    if (xi.dims[0] != P.batch_size || xi.dims[1] != P.segment_size || xi.dims[2] != P.mel_features)
        return fail(err, "encoder input x does not match PARAMS (batch, segment_size, mel_features)");
    D.seg = (int)xi.dims[1]; D.mel = (int)xi.dims[2];
    D.n_layers = (int)hi.dims[0]; D.d_model = (int)hi.dims[2]; D.hidden = (int)ci.dims[2];
    if (hi.dims[1] != 1 || ci.dims[1] != 1 || ci.dims[0] != hi.dims[0]) return fail(err, "state tensors must be (L,1,*)");
    if (g.outputs[0].dims.size() != 3) return fail(err, "encoder_out must be rank 3");
    D.joiner = (int)g.outputs[0].dims[2];

    std::vector<int> wn;
    for (size_t i = 0; i < g.nodes.size(); ++i) if (v.weighted(g.nodes[i])) wn.push_back((int)i);
    const int L = D.n_layers;
    if ((int)wn.size() != 3 + 1 + 5 * L + 1)
        return fail(err, "encoder: expected " + std::to_string(5 + 5 * L) + " weight-bearing nodes, found " + std::to_string(wn.size()));

    // --- conv stack
    int H = D.seg, W = D.mel, C = 1;
    std::string cur;
    for (int i = 0; i < 3; ++i) {
        const ONode &n = g.nodes[wn[i]];
        if (n.op != "Conv") return fail(err, "encoder: node " + std::to_string(i) + " of the embed stack is not Conv but " + View::describe(n));
        OTensor w;
        if (!v.const_tensor(n.in[1], w)) return fail(err, "embed conv weight is not a constant: " + View::describe(n));
        if (w.dims.size() != 4 || w.dims[2] != 3 || w.dims[3] != 3 || w.dims[1] != C) return fail(err, "embed conv must be 3x3 over " + std::to_string(C) + " channels");
        if (n.attr_i("group", 1) != 1) return fail(err, "embed conv group != 1");
        int st = 1;
        if (auto a = n.attr("strides")) { if (a->ints.size() != 2 || a->ints[0] != a->ints[1]) return fail(err, "embed conv strides"); st = (int)a->ints[0]; }
        if (auto a = n.attr("pads")) for (auto p : a->ints) if (p != 0) return fail(err, "embed conv padding unsupported");
        if (auto a = n.attr("dilations")) for (auto p : a->ints) if (p != 1) return fail(err, "embed conv dilation unsupported");
        D.conv_ch[i] = (int)w.dims[0]; D.conv_stride[i] = st;
        M.conv_w[i] = w.f;
        if (n.in.size() > 2 && !n.in[2].empty()) { if (!v.const_floats(n.in[2], M.conv_b[i]) || (int)M.conv_b[i].size() != D.conv_ch[i]) return fail(err, "embed conv bias"); }
        else M.conv_b[i].assign((size_t)D.conv_ch[i], 0.0f);
        H = (H - 3) / st + 1; W = (W - 3) / st + 1; C = D.conv_ch[i];
        if (!v.double_swish(n.out[0], cur)) return fail(err, "encoder embed: " + v.err);
    }
    if (H != 1) return fail(err, "embed stack must reduce the segment to one frame (got " + std::to_string(H) + ")");
    D.f_out = W; D.embed_in = C * W;

    Linear lin;
    if (!v.linear_at(wn[3], lin)) return fail(err, "encoder embed linear: " + v.err);
    if (lin.K != D.embed_in || lin.N != D.d_model) return fail(err, "encoder embed linear has unexpected shape");
    M.w_embed = lin.W; M.b_embed = lin.b.empty() ? std::vector<float>((size_t)lin.N, 0.0f) : lin.b;
    if (!v.basic_norm(lin.out_value, M.embed_norm_eps, cur)) return fail(err, "encoder embed: " + v.err);

    // --- layers
    M.layers.resize((size_t)L);
    const int d = D.d_model, Hh = D.hidden;
    for (int l = 0; l < L; ++l) {
        LayerWeights &lw = M.layers[(size_t)l];
        const int base = 4 + 5 * l;
        Linear a, b, hr, f1, f2;
        if (!v.linear_at(wn[base], a) || !v.linear_at(wn[base + 1], b) || !v.linear_at(wn[base + 2], hr) ||
            !v.linear_at(wn[base + 3], f1) || !v.linear_at(wn[base + 4], f2))
            return fail(err, "encoder layer " + std::to_string(l) + ": " + v.err);
        const bool a_is_h = v.lineage(a.in_value) == hi.name, b_is_h = v.lineage(b.in_value) == hi.name;
        if (a_is_h == b_is_h) return fail(err, "encoder layer " + std::to_string(l) + ": cannot tell input and recurrent gate matmuls apart");
        const Linear &ih = a_is_h ? b : a, &hh = a_is_h ? a : b;
        if (ih.K != d || hh.K != d || ih.N != 4 * Hh || hh.N != 4 * Hh) return fail(err, "encoder layer " + std::to_string(l) + ": gate matmul shapes");
        // gates = ih + hh (+ bias) -> four equal parts (Split, or four Slices) -> activations.  WHICH part is which gate is
        // read off the cell update c' = sigma(f) * c_prev + sigma(i) * tanh(g), h = sigma(o) * tanh(c'): the sigmoid whose
        // product meets the previous cell state is f, the one multiplied with the tanh part is i, the remaining sigmoid
        // (multiplied with tanh of the new cell) is o.  torch.nn.LSTM order is i,f,g,o, but nothing here depends on it.
        const std::string lay = "encoder layer " + std::to_string(l) + ": ";
        std::vector<float> extra_bias;
        int part_of_gate[4] = {-1, -1, -1, -1};            // gate i,f,g,o -> index of the quarter of the 4H columns
        {
            const ONode *sum = nullptr;
            const std::string ih_out = v.downstream(ih.out_value), hh_out = v.downstream(hh.out_value);
            for (const ONode *c : v.consumers(ih_out)) if (c->op == "Add" && c->in.size() == 2 && (v.same(c->in[0], hh_out) || v.same(c->in[1], hh_out))) sum = c;
            if (!sum) {
                std::string found;
                for (const ONode *c : v.consumers(ih_out)) { found = "; the input product feeds " + View::describe(*c); break; }
                return fail(err, lay + "gate pre-activations are not summed by one Add" + found);
            }
            std::string gates = sum->out[0];
            for (const ONode *c : v.consumers(gates)) {        // bias added after the sum
                if (c->op != "Add" || c->in.size() != 2) continue;
                const std::string &o = v.same(c->in[0], gates) ? c->in[1] : c->in[0];
                std::vector<float> b;
                if (v.const_floats(o, b) && (int)b.size() == 4 * Hh) { extra_bias = b; gates = c->out[0]; break; }
            }
            for (int guard = 0; guard < 8; ++guard) {          // layout-only operators between the sum and its four parts (2-D Gemm operands reshaped back)
                auto cs = v.consumers(gates);
                if (cs.size() != 1 || cs[0]->out.empty()) break;
                const std::string &op = cs[0]->op;
                if (op == "Reshape" || op == "Squeeze" || op == "Unsqueeze" || op == "Identity" || op == "Flatten" || op == "Cast") gates = cs[0]->out[0];
                else break;
            }
            std::string part[4];
            const ONode *split = nullptr;
            for (const ONode *c : v.consumers(gates)) if (c->op == "Split") split = c;
            if (split) {
                if (split->out.size() != 4) return fail(err, lay + "expected 4 gate parts: " + View::describe(*split));
                if (const OAttr *sp = split->attr("split")) for (auto x : sp->ints) if (x != Hh) return fail(err, lay + "unequal gate parts: " + View::describe(*split));
                for (int k = 0; k < 4; ++k) part[k] = split->out[k];
            } else {
                int seen = 0;
                for (const ONode *c : v.consumers(gates)) {
                    if (c->op != "Slice" || c->out.empty()) continue;
                    std::vector<float> st, en;
                    if (c->in.size() >= 3) {
                        if (!v.const_floats(c->in[1], st) || !v.const_floats(c->in[2], en)) {
                            // bounds computed from Shape(gates) (torch.onnx, Tensor.chunk): the width of the sum is 4 * hidden
                            const View::DimOf width = [&](const std::string &t, long, long &o) { if (!v.same(t, gates)) return false; o = 4L * Hh; return true; };
                            long s1 = 0, e1 = 0;
                            if (!v.const_index(c->in[1], width, s1) || !v.const_index(c->in[2], width, e1)) continue;
                            st.assign(1, (float)s1); en.assign(1, (float)e1);
                        }
                    }
                    else { const OAttr *a = c->attr("starts"), *b = c->attr("ends"); if (!a || !b) continue; for (auto x : a->ints) st.push_back((float)x); for (auto x : b->ints) en.push_back((float)x); }
                    if (st.size() != 1 || en.size() != 1) continue;
                    const long s0 = (long)st[0], e0 = (long)en[0];
                    if (s0 % Hh != 0 || s0 < 0 || s0 >= 4L * Hh || !(e0 == s0 + Hh || (s0 == 3L * Hh && e0 >= 4L * Hh))) return fail(err, lay + "gate Slice bounds: " + View::describe(*c));
                    part[s0 / Hh] = c->out[0]; ++seen;
                }
                if (seen != 4) {
                    std::string found;
                    for (const ONode *c : v.consumers(gates)) { found = "; the sum feeds " + View::describe(*c); break; }
                    return fail(err, lay + "expected Split (or four Slices) of the gate pre-activations" + found);
                }
            }
            std::string act_out[4], act_op[4];
            for (int k = 0; k < 4; ++k) {
                auto cs = v.consumers(part[k]);
                if (cs.size() != 1 || (cs[0]->op != "Sigmoid" && cs[0]->op != "Tanh") || cs[0]->out.empty())
                    return fail(err, lay + "gate part " + std::to_string(k) + " must feed exactly one Sigmoid or Tanh" + (cs.empty() ? std::string() : ", found " + View::describe(*cs[0])));
                act_op[k] = cs[0]->op; act_out[k] = cs[0]->out[0];
            }
            int tanh_part = -1, n_tanh = 0;
            for (int k = 0; k < 4; ++k) if (act_op[k] == "Tanh") { tanh_part = k; ++n_tanh; }
            if (n_tanh != 1) return fail(err, lay + "expected three sigmoid gates and one tanh gate");
            part_of_gate[2] = tanh_part;
            for (int k = 0; k < 4; ++k) {
                if (k == tanh_part) continue;
                int role = -1;
                for (const ONode *m : v.consumers(act_out[k])) {
                    if (m->op != "Mul" || m->in.size() != 2) continue;
                    const std::string &other = v.same(m->in[0], act_out[k]) ? m->in[1] : m->in[0];
                    if (v.lineage(other) == ci.name) role = 1;                                  // * c_prev  -> forget gate
                    else if (v.same(other, act_out[tanh_part])) role = 0;                       // * tanh(g) -> input gate
                    else { const ONode *pp = v.producer(v.source(other)); if (pp && pp->op == "Tanh") role = 3; }   // * tanh(c') -> output gate
                }
                if (role < 0 || part_of_gate[role] >= 0) return fail(err, lay + "cannot tell the role of gate part " + std::to_string(k) + " from the cell update");
                part_of_gate[role] = k;
            }
        }
        // canonical column order i,f,g,o
        lw.w_gates.assign((size_t)2 * d * 4 * Hh, 0.0f);
        lw.b_gates.assign((size_t)4 * Hh, 0.0f);
        for (int gate = 0; gate < 4; ++gate) {
            const int src = part_of_gate[gate];
            for (int k = 0; k < d; ++k) {
                memcpy(&lw.w_gates[((size_t)k) * 4 * Hh + (size_t)gate * Hh], &ih.W[(size_t)k * 4 * Hh + (size_t)src * Hh], (size_t)Hh * 4);
                memcpy(&lw.w_gates[((size_t)(d + k)) * 4 * Hh + (size_t)gate * Hh], &hh.W[(size_t)k * 4 * Hh + (size_t)src * Hh], (size_t)Hh * 4);
            }
            for (int u = 0; u < Hh; ++u) {
                const size_t sidx = (size_t)src * Hh + u;
                // (b_ih + b_hh) first, a bias added after the sum last: any spelling of the same three terms gives these bits
                float b = (ih.b.empty() ? 0.0f : ih.b[sidx]) + (hh.b.empty() ? 0.0f : hh.b[sidx]);
                if (!extra_bias.empty()) b += extra_bias[sidx];
                lw.b_gates[(size_t)gate * Hh + u] = b;
            }
        }
        if (hr.K != Hh || hr.N != d || !hr.b.empty()) return fail(err, "encoder layer " + std::to_string(l) + ": projection matmul shape");
        lw.w_hr = hr.W;
        if (f1.K != d || f2.N != d || f1.N != f2.K) return fail(err, "encoder layer " + std::to_string(l) + ": feed-forward shapes");
        if (l == 0) D.ffn = f1.N; else if (D.ffn != f1.N) return fail(err, "feed-forward width differs between layers");
        lw.w_ff1 = f1.W; lw.b_ff1 = f1.b.empty() ? std::vector<float>((size_t)f1.N, 0.0f) : f1.b;
        lw.w_ff2 = f2.W; lw.b_ff2 = f2.b.empty() ? std::vector<float>((size_t)f2.N, 0.0f) : f2.b;
        std::string act;
        if (!v.double_swish(f1.out_value, act)) return fail(err, "encoder layer " + std::to_string(l) + ": " + v.err);
        // residual add after ff2, then BasicNorm
        bool normed = false;
        for (const ONode *c : v.consumers(f2.out_value)) if (c->op == "Add" && !c->out.empty()) {
            std::string o;
            if (v.basic_norm(c->out[0], lw.norm_eps, o)) { normed = true; break; }
        }
        if (!normed) return fail(err, "encoder layer " + std::to_string(l) + ": " + v.err);
    }
    Linear ep;
    if (!v.linear_at(wn.back(), ep)) return fail(err, "encoder_proj: " + v.err);
    if (ep.K != d || ep.N != D.joiner) return fail(err, "encoder_proj shape");
    if (!v.same(ep.out_value, g.outputs[0].name)) return fail(err, "encoder_proj does not produce the first graph output");
    M.w_encproj = ep.W; M.b_encproj = ep.b.empty() ? std::vector<float>((size_t)ep.N, 0.0f) : ep.b;
    return true;
}

bool extract_decoder(const OGraph &g, HostModel &M, std::string &err)
{
    View v(g);
    NetDims &D = M.dims;
    if (g.inputs.size() != 1 || g.outputs.size() != 1) return fail(err, "decoder must have 1 input and 1 output");
    if (g.inputs[0].dims.size() != 2 || g.inputs[0].dims[0] != 1) return fail(err, "Currently, only batch size 1 models are supported (decoder context)");
    D.context = (int)g.inputs[0].dims[1];
    if (D.context != 2) return fail(err, "decoder context size must be 2 (reference src/april_session.c:322,297 assume it)");
    const ONode *gather = nullptr, *conv = nullptr; bool relu = false; int mm = -1;
    for (size_t i = 0; i < g.nodes.size(); ++i) {
        const ONode &n = g.nodes[i];
        OTensor tt;
        if (n.op == "Gather" && !n.in.empty() && v.const_tensor(n.in[0], tt) && tt.dims.size() == 2) gather = &n;
        else if (n.op == "Conv") conv = &n;
        else if (n.op == "Relu") relu = true;
        else if (v.weighted(n) && n.op != "Conv") mm = (int)i;
    }
    if (!gather || !conv || !relu || mm < 0) return fail(err, "decoder: expected Gather(embedding) -> Conv -> Relu -> Linear");
    OTensor emb;
    if (!v.const_tensor(gather->in[0], emb) || emb.f.size() != emb.numel()) return fail(err, "decoder embedding table must be float32");
    M.emb = emb.f;
    const int V = (int)emb.dims[0], dd = (int)emb.dims[1];
    OTensor cw;
    if (conv->in.size() < 2 || !v.const_tensor(conv->in[1], cw)) return fail(err, "decoder conv weight must be a constant: " + View::describe(*conv));
    if (cw.dims.size() != 3 || cw.dims[0] != dd || cw.dims[2] != D.context) return fail(err, "decoder conv weight shape");
    D.dec_groups = (int)conv->attr_i("group", 1);
    if (cw.dims[1] * D.dec_groups != dd) return fail(err, "decoder conv groups do not divide channels");
    M.dec_conv = cw.f;
    if (conv->in.size() > 2 && !conv->in[2].empty()) { if (!v.const_floats(conv->in[2], M.dec_conv_b) || (int)M.dec_conv_b.size() != dd) return fail(err, "decoder conv bias must be a constant of length d_model"); }
    Linear p;
    if (!v.linear_at(mm, p)) return fail(err, "decoder_proj: " + v.err);
    if (p.K != dd) return fail(err, "decoder_proj input width");
    if (!v.same(p.out_value, g.outputs[0].name)) return fail(err, "decoder_proj does not produce the graph output");
    M.w_decproj = p.W; M.b_decproj = p.b.empty() ? std::vector<float>((size_t)p.N, 0.0f) : p.b;
    if (dd != D.d_model) return fail(err, "decoder embedding width must equal encoder d_model in this engine");
    if (p.N != D.joiner) return fail(err, "decoder_out width differs from encoder_out width");
    D.vocab = V;
    return true;
}

bool extract_joiner(const OGraph &g, HostModel &M, std::string &err)
{
    View v(g);
    NetDims &D = M.dims;
    if (g.inputs.size() != 2 || g.outputs.size() != 1) return fail(err, "joiner must have 2 inputs and 1 output");
    bool tanh_seen = false; int mm = -1;
    for (size_t i = 0; i < g.nodes.size(); ++i) {
        if (g.nodes[i].op == "Tanh") tanh_seen = true;
        if (v.weighted(g.nodes[i])) mm = (int)i;
    }
    if (!tanh_seen || mm < 0) return fail(err, "joiner: expected tanh(enc + dec) -> Linear");
    Linear o;
    if (!v.linear_at(mm, o)) return fail(err, "joiner output linear: " + v.err);
    if (o.K != D.joiner) return fail(err, "joiner input width");
    if (o.N != D.vocab) return fail(err, "joiner vocabulary differs from the decoder embedding table");
    if (!v.same(o.out_value, g.outputs[0].name)) return fail(err, "joiner linear does not produce the graph output");
    if (g.outputs[0].dims.size() != 3) return fail(err, "logits must be rank 3");
    M.w_out = o.W; M.b_out = o.b.empty() ? std::vector<float>((size_t)o.N, 0.0f) : o.b;
    return true;
}

}  // namespace

namespace {
// rows x cols (row-major) -> new_rows x new_cols, element (r, c) -> (rmap(r), cmap(c)), zeros elsewhere
template <class RM, class CM>
std::vector<float> pad2(const std::vector<float> &src, int rows, int cols, int new_rows, int new_cols, RM rmap, CM cmap)
{
    std::vector<float> dst((size_t)new_rows * new_cols, 0.0f);
    for (int r = 0; r < rows; ++r) {
        const float *s = src.data() + (size_t)r * cols;
        float *d = dst.data() + (size_t)rmap(r) * new_cols;
        for (int c = 0; c < cols; ++c) d[cmap(c)] = s[c];
    }
    return dst;
}
std::vector<float> pad1(const std::vector<float> &src, int new_n)
{
    std::vector<float> dst((size_t)new_n, 0.0f);
    std::copy(src.begin(), src.end(), dst.begin());
    return dst;
}
int up64(int v) { return (v + 63) & ~63; }
}  // namespace

bool pad_host_model(HostModel &m, std::string &err)
{
    NetDims &D = m.dims;
    const int d = D.d_model, h = D.hidden, f = D.ffn, j = D.joiner, c2 = D.conv_ch[2], F = D.f_out;
    const int d2 = up64(d), h2 = up64(h), f2 = up64(f), j2 = up64(j);
    int c22 = (c2 + 15) & ~15;
    while ((c22 * F) % 64) c22 += 16;
    if (d2 == d && h2 == h && f2 == f && j2 == j && c22 == c2) return true;
    const int gs = d / D.dec_groups;                     // channels per group of the decoder's convolution
    if ((d2 - d) % gs) { err = "d_model cannot be padded to a multiple of 64 in whole groups of the decoder convolution"; return false; }
    auto id = [](int i) { return i; };
    // third conv: [c2][c1 * 9] output channels; embed linear rows are channel-major (c * F + f)
    const int k9 = D.conv_ch[1] * 9;
    m.conv_w[2] = pad2(m.conv_w[2], c2, k9, c22, k9, id, id);
    m.conv_b[2] = pad1(m.conv_b[2], c22);
    m.w_embed = pad2(m.w_embed, c2 * F, d, c22 * F, d2, id, id);      // (rows c * F + f keep their index: new channels follow)
    m.b_embed = pad1(m.b_embed, d2);
    auto xh_row = [&](int r) { return r < d ? r : d2 + (r - d); };          // gate GEMM rows: x part, then h part
    auto gate_col = [&](int c) { return (c / h) * h2 + c % h; };             // gate-major columns i, f, g, o
    for (LayerWeights &lw : m.layers) {
        lw.w_gates = pad2(lw.w_gates, 2 * d, 4 * h, 2 * d2, 4 * h2, xh_row, gate_col);
        {
            std::vector<float> b((size_t)4 * h2, 0.0f);
            for (int c = 0; c < 4 * h; ++c) b[(size_t)gate_col(c)] = lw.b_gates[(size_t)c];
            lw.b_gates.swap(b);
        }
        lw.w_hr = pad2(lw.w_hr, h, d, h2, d2, id, id);
        lw.w_ff1 = pad2(lw.w_ff1, d, f, d2, f2, id, id); lw.b_ff1 = pad1(lw.b_ff1, f2);
        lw.w_ff2 = pad2(lw.w_ff2, f, d, f2, d2, id, id); lw.b_ff2 = pad1(lw.b_ff2, d2);
    }
    m.w_encproj = pad2(m.w_encproj, d, j, d2, j2, id, id); m.b_encproj = pad1(m.b_encproj, j2);
    m.emb = pad2(m.emb, D.vocab, d, D.vocab, d2, id, id);
    m.dec_conv = pad2(m.dec_conv, d, gs * D.context, d2, gs * D.context, id, id);
    if (!m.dec_conv_b.empty()) m.dec_conv_b = pad1(m.dec_conv_b, d2);
    m.w_decproj = pad2(m.w_decproj, d, j, d2, j2, id, id); m.b_decproj = pad1(m.b_decproj, j2);
    m.w_out = pad2(m.w_out, j, D.vocab, j2, D.vocab, id, id);
    D.d_norm = d;
    D.d_model = d2; D.hidden = h2; D.ffn = f2; D.joiner = j2; D.conv_ch[2] = c22; D.embed_in = c22 * F; D.dec_groups = d2 / gs;
    return true;
}

bool load_april_file(const char *path, HostModel &out, std::string &err)
{
    if (!path) { err = "no model path given"; return false; }
    FILE *fd = fopen(path, "rb");
    if (!fd) { err = std::string("cannot open ") + path; return false; }
    fseek(fd, 0, SEEK_END);
    long sz = ftell(fd);
    fseek(fd, 0, SEEK_SET);
    std::vector<uint8_t> blob(sz > 0 ? (size_t)sz : 0);
    size_t got = blob.empty() ? 0 : fread(blob.data(), 1, blob.size(), fd);
    fclose(fd);
    if (got != blob.size()) { err = "short read"; return false; }

    ContainerInfo info;
    if (!parse_container(blob, info, err)) return false;
    // reference src/april_model.c:36-40
    if (info.model_type != 1 || info.net_off.size() != 3) { err = "Model has unknown model type, or the wrong number of networks"; return false; }
    out.language = info.language; out.name = info.name; out.description = info.description;
    out.params = info.params;

    OGraph enc, dec, joi;
    std::string e;
    if (!parse_onnx(blob.data() + info.net_off[0], (size_t)info.net_size[0], enc, e)) { err = "encoder graph: " + e; return false; }
    if (!parse_onnx(blob.data() + info.net_off[1], (size_t)info.net_size[1], dec, e)) { err = "decoder graph: " + e; return false; }
    if (!parse_onnx(blob.data() + info.net_off[2], (size_t)info.net_size[2], joi, e)) { err = "joiner graph: " + e; return false; }
    if (!extract_encoder(enc, out.params, out, err)) return false;
    if (!extract_decoder(dec, out, err)) return false;
    if (!extract_joiner(joi, out, err)) return false;
    // reference src/april_model.c:99-102
    if (out.dims.vocab != out.params.token_count) { err = "logits width differs from PARAMS token_count"; return false; }
    return true;
}

}  // namespace aprilx
