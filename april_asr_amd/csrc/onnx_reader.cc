// See onnx_reader.h.  Field numbers follow onnx.proto3 (ModelProto.graph = 7;
// GraphProto.node = 1, initializer = 5, input = 11, output = 12; NodeProto
// input/output/name/op_type/attribute = 1/2/3/4/5; AttributeProto name/f/i/t/ints =
// 1/2/3/5/8; TensorProto dims/data_type/float_data/int32_data/int64_data/name/
// raw_data = 1/2/4/5/7/8/9).
#include "onnx_reader.h"
#include <cstring>

namespace aprilx {
namespace {

struct Cursor {
    const uint8_t *p, *end;
    bool ok = true;
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; p < end && shift < 70; shift += 7) {
            uint8_t c = *p++;
            v |= (uint64_t)(c & 0x7f) << shift;
            if (!(c & 0x80)) return v;
        }
        ok = false;
        return v;
    }
};

struct Field { int num = 0, wire = 0; uint64_t val = 0; Cursor sub{nullptr, nullptr}; };

bool next_field(Cursor &c, Field &f)
{
    if (c.p >= c.end || !c.ok) return false;
    uint64_t key = c.varint();
    f.num = (int)(key >> 3);
    f.wire = (int)(key & 7);
    switch (f.wire) {
    case 0: f.val = c.varint(); break;
    case 1: if (c.end - c.p < 8) { c.ok = false; return false; } memcpy(&f.val, c.p, 8); c.p += 8; break;
    case 5: { if (c.end - c.p < 4) { c.ok = false; return false; } uint32_t t; memcpy(&t, c.p, 4); f.val = t; c.p += 4; break; }
    case 2: {
        uint64_t len = c.varint();
        if ((uint64_t)(c.end - c.p) < len) { c.ok = false; return false; }
        f.sub = Cursor{c.p, c.p + len};
        c.p += len;
        break;
    }
    default: c.ok = false; return false;
    }
    return c.ok;
}

std::string as_string(const Cursor &s) { return std::string((const char *)s.p, (size_t)(s.end - s.p)); }

bool read_tensor(Cursor c, OTensor &t, std::string &err)
{
    Field f;
    const uint8_t *raw = nullptr; size_t raw_n = 0;
    std::vector<float> fd; std::vector<int64_t> id;
    while (next_field(c, f)) {
        switch (f.num) {
        case 1:
            if (f.wire == 2) { Cursor q = f.sub; while (q.p < q.end) t.dims.push_back((int64_t)q.varint()); }
            else t.dims.push_back((int64_t)f.val);
            break;
        case 2: t.dtype = (int)f.val; break;
        case 4:
            if (f.wire == 2) { size_t k = (size_t)(f.sub.end - f.sub.p) / 4; size_t o = fd.size(); fd.resize(o + k); memcpy(fd.data() + o, f.sub.p, k * 4); }
            else { uint32_t u = (uint32_t)f.val; float x; memcpy(&x, &u, 4); fd.push_back(x); }
            break;
        case 5: case 7:
            if (f.wire == 2) { Cursor q = f.sub; while (q.p < q.end) id.push_back((int64_t)q.varint()); }
            else id.push_back((int64_t)f.val);
            break;
        case 8: t.name = as_string(f.sub); break;
        case 9: raw = f.sub.p; raw_n = (size_t)(f.sub.end - f.sub.p); break;
        default: break;
        }
    }
    if (!c.ok) { err = "malformed TensorProto"; return false; }
    const size_t n = t.numel();
    if (t.dtype == 1) {
        if (raw) { if (raw_n != n * 4) { err = "tensor '" + t.name + "': raw_data size mismatch"; return false; } t.f.resize(n); memcpy(t.f.data(), raw, raw_n); }
        else { if (fd.size() != n) { err = "tensor '" + t.name + "': float_data size mismatch"; return false; } t.f.swap(fd); }
    } else if (t.dtype == 7) {
        if (raw) { if (raw_n != n * 8) { err = "tensor '" + t.name + "': raw_data size mismatch"; return false; } t.i.resize(n); memcpy(t.i.data(), raw, raw_n); }
        else { if (id.size() != n) { err = "tensor '" + t.name + "': int64_data size mismatch"; return false; } t.i.swap(id); }
    } else if (t.dtype == 6) {
        t.i.resize(n);
        if (raw) { if (raw_n != n * 4) { err = "tensor '" + t.name + "': raw_data size mismatch"; return false; } for (size_t k = 0; k < n; ++k) { int32_t v; memcpy(&v, raw + 4 * k, 4); t.i[k] = v; } }
        else { if (id.size() != n) { err = "int32_data size mismatch"; return false; } for (size_t k = 0; k < n; ++k) t.i[k] = (int32_t)id[k]; }
    } else {
        err = "tensor '" + t.name + "': unsupported data_type " + std::to_string(t.dtype);
        return false;
    }
    return true;
}

bool read_attr(Cursor c, OAttr &a, std::string &err)
{
    Field f;
    while (next_field(c, f)) {
        switch (f.num) {
        case 1: a.name = as_string(f.sub); break;
        case 2: { uint32_t u = (uint32_t)f.val; memcpy(&a.f, &u, 4); break; }
        case 3: a.i = (int64_t)f.val; break;
        case 5: if (!read_tensor(f.sub, a.t, err)) return false; a.has_t = true; break;
        case 8:
            if (f.wire == 2) { Cursor q = f.sub; while (q.p < q.end) a.ints.push_back((int64_t)q.varint()); }
            else a.ints.push_back((int64_t)f.val);
            break;
        default: break;
        }
    }
    return c.ok;
}

bool read_node(Cursor c, ONode &n, std::string &err)
{
    Field f;
    while (next_field(c, f)) {
        switch (f.num) {
        case 1: n.in.push_back(as_string(f.sub)); break;
        case 2: n.out.push_back(as_string(f.sub)); break;
        case 3: n.name = as_string(f.sub); break;
        case 4: n.op = as_string(f.sub); break;
        case 5: { OAttr a; if (!read_attr(f.sub, a, err)) return false; n.attrs.push_back(std::move(a)); break; }
        default: break;
        }
    }
    return c.ok;
}

void read_value_info(Cursor c, OValueInfo &v)
{
    Field f, f2, f3, f4, f5;
    while (next_field(c, f)) {
        if (f.num == 1) v.name = as_string(f.sub);
        else if (f.num == 2) {
            Cursor t = f.sub;
            while (next_field(t, f2)) if (f2.num == 1) {           // tensor_type
                Cursor tt = f2.sub;
                while (next_field(tt, f3)) {
                    if (f3.num == 1) v.elem = (int)f3.val;
                    else if (f3.num == 2) {                            // shape
                        Cursor sh = f3.sub;
                        while (next_field(sh, f4)) if (f4.num == 1) {  // dim
                            Cursor d = f4.sub; int64_t dv = -1;
                            while (next_field(d, f5)) if (f5.num == 1) dv = (int64_t)f5.val;
                            v.dims.push_back(dv);
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

bool parse_onnx(const uint8_t *data, size_t size, OGraph &g, std::string &err)
{
    Cursor m{data, data + size};
    Field f;
    Cursor graph{nullptr, nullptr};
    while (next_field(m, f)) if (f.num == 7 && f.wire == 2) graph = f.sub;
    if (!m.ok || !graph.p) { err = "not an ONNX ModelProto (no graph)"; return false; }
    Cursor c = graph;
    std::vector<OValueInfo> raw_inputs;
    while (next_field(c, f)) {
        if (f.num == 1) { ONode n; if (!read_node(f.sub, n, err)) { if (err.empty()) err = "malformed NodeProto"; return false; } g.nodes.push_back(std::move(n)); }
        else if (f.num == 5) { OTensor t; if (!read_tensor(f.sub, t, err)) return false; std::string nm = t.name; g.inits.emplace(nm, std::move(t)); }
        else if (f.num == 11) { OValueInfo v; read_value_info(f.sub, v); raw_inputs.push_back(std::move(v)); }
        else if (f.num == 12) { OValueInfo v; read_value_info(f.sub, v); g.outputs.push_back(std::move(v)); }
    }
    if (!c.ok) { err = "malformed GraphProto"; return false; }
    for (auto &v : raw_inputs) if (!g.inits.count(v.name)) g.inputs.push_back(v);
    for (size_t i = 0; i < g.nodes.size(); ++i) {
        for (auto &o : g.nodes[i].out) g.producer[o] = (int)i;
        for (auto &in : g.nodes[i].in) if (!in.empty()) g.consumers.emplace(in, (int)i);
    }
    return true;
}

}  // namespace aprilx
