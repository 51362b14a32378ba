// Minimal ONNX (protobuf wire format) reader for the three graphs embedded in a
// .april file.  Product code: used at model-load time only, to locate weights
// and constants structurally (replaces ORT's CreateSessionFromArray,
// reference src/ort_util.h:127-134).  No graph is ever *executed* on the host.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace aprilx {

struct OTensor {
    std::string name;
    int dtype = 0;                  // 1 float32, 7 int64, 6 int32
    std::vector<int64_t> dims;
    std::vector<float> f;           // when dtype == 1
    std::vector<int64_t> i;         // when dtype == 6/7
    size_t numel() const { size_t n = 1; for (auto d : dims) n *= (size_t)d; return n; }
};

struct OAttr {
    std::string name;
    float f = 0;
    int64_t i = 0;
    std::vector<int64_t> ints;
    OTensor t;
    bool has_t = false;
};

struct ONode {
    std::string op, name;
    std::vector<std::string> in, out;
    std::vector<OAttr> attrs;
    const OAttr *attr(const char *n) const {
        for (auto &a : attrs) if (a.name == n) return &a;
        return nullptr;
    }
    int64_t attr_i(const char *n, int64_t def) const { auto a = attr(n); return a ? a->i : def; }
};

struct OValueInfo { std::string name; int elem = 0; std::vector<int64_t> dims; };

struct OGraph {
    std::vector<ONode> nodes;
    std::map<std::string, OTensor> inits;
    std::vector<OValueInfo> inputs, outputs;        // inputs exclude initializers
    std::map<std::string, int> producer;            // value name -> node index
    std::multimap<std::string, int> consumers;      // value name -> node indices
};

bool parse_onnx(const uint8_t *data, size_t size, OGraph &g, std::string &err);

}  // namespace aprilx
