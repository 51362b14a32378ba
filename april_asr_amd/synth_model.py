"""Writer for synthetic `.april` model files (seeded random weights).

There is no real `aprilv0_en-us.april` in this environment (and no network), so
tests, smoke() and bench.py generate a model of the same architecture and
dimensions instead.  The container layout follows the reference's
`extra/file-format.md` and `extra/export-april.py:374-443`; the three ONNX graphs
are written in protobuf wire format by hand (no `onnx` package here) and mimic
what `torch.onnx.export(opset 11)` emits for the icefall
`lstm_transducer_stateless2` recipe after `convert_scaled_to_non_scaled(is_onnx=True)`
(reference: `extra/export-april.py:183-331,564`):

  encoder : Conv2dSubsampling (3x Conv2d + DoubleSwish, Linear, BasicNorm) ->
            L x { LSTM-with-projection written out as Gemm/Split/Sigmoid/Tanh/Mul/MatMul,
                  residual, FFN (Linear, DoubleSwish, Linear), residual, BasicNorm } ->
            Linear (joiner.encoder_proj)
  decoder : Embedding gather -> grouped Conv1d(k=context) -> ReLU -> Linear (joiner.decoder_proj)
  joiner  : tanh(enc + dec) -> Linear

`variant` knobs change *how* equivalent things are spelled (Gemm vs MatMul+Add,
folded vs unfolded BasicNorm eps) so the loader's structural weight extraction can
be tested against more than one spelling.
"""
import io
import struct
import numpy as np

# ------------------------------------------------------------------ protobuf wire helpers
FLOAT, INT64 = 1, 7
A_FLOAT, A_INT, A_STRING, A_TENSOR, A_FLOATS, A_INTS = 1, 2, 3, 4, 6, 7


def _varint(n):
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wt):
    return _varint((field << 3) | wt)


def _f_varint(field, v):
    return _key(field, 0) + _varint(v)


def _f_bytes(field, b):
    if isinstance(b, str):
        b = b.encode()
    if not isinstance(b, bytes):
        b = bytes(b)
    return b"".join((_key(field, 2), _varint(len(b)), b))


def _f_float(field, v):
    return _key(field, 5) + struct.pack("<f", v)


def tensor_proto(name, arr):
    arr = np.asarray(arr)
    if arr.ndim:      # np.ascontiguousarray would promote 0-d to 1-d
        arr = np.ascontiguousarray(arr)
    if arr.dtype == np.float32:
        dt = FLOAT
    elif arr.dtype == np.int64:
        dt = INT64
    else:
        raise TypeError(arr.dtype)
    out = b"".join(_f_varint(1, int(d)) for d in arr.shape)
    out += _f_varint(2, dt)
    if name:
        out += _f_bytes(8, name)
    out += _f_bytes(9, arr.tobytes())
    return out


def _attr(name, value):
    out = _f_bytes(1, name)
    if isinstance(value, float):
        out += _f_float(2, value) + _f_varint(20, A_FLOAT)
    elif isinstance(value, (int, np.integer)):
        out += _f_varint(3, int(value)) + _f_varint(20, A_INT)
    elif isinstance(value, np.ndarray):
        out += _f_bytes(5, tensor_proto("", value)) + _f_varint(20, A_TENSOR)
    elif isinstance(value, (list, tuple)):
        out += b"".join(_f_varint(8, int(v)) for v in value) + _f_varint(20, A_INTS)
    else:
        raise TypeError(type(value))
    return out


def node_proto(op, inputs, outputs, name="", **attrs):
    out = b"".join(_f_bytes(1, i) for i in inputs)
    out += b"".join(_f_bytes(2, o) for o in outputs)
    if name:
        out += _f_bytes(3, name)
    out += _f_bytes(4, op)
    out += b"".join(_f_bytes(5, _attr(k, v)) for k, v in attrs.items())
    return out


def value_info(name, elem_type, dims):
    shape = b"".join(_f_bytes(1, _f_varint(1, int(d))) for d in dims)
    tensor_type = _f_varint(1, elem_type) + _f_bytes(2, shape)
    return _f_bytes(1, name) + _f_bytes(2, _f_bytes(1, tensor_type))


class GraphBuilder:
    def __init__(self, name):
        self.name = name
        self.nodes, self.inits, self.inputs, self.outputs = [], [], [], []
        self._n = 0

    def fresh(self, hint="t"):
        self._n += 1
        return "%s_%d" % (hint, self._n)

    def init(self, name, arr):
        self.inits.append(tensor_proto(name, arr))
        return name

    def node(self, op, inputs, n_out=1, hint=None, **attrs):
        outs = [self.fresh(hint or op.lower()) for _ in range(n_out)]
        self.nodes.append(node_proto(op, inputs, outs, name="%s_%d" % (op, len(self.nodes)), **attrs))
        return outs[0] if n_out == 1 else outs

    def node_named(self, op, inputs, outputs, **attrs):
        self.nodes.append(node_proto(op, inputs, outputs, name="%s_%d" % (op, len(self.nodes)), **attrs))

    def const(self, arr):
        return self.node("Constant", [], value=np.asarray(arr))

    def model_bytes(self, opset=11):
        g = b"".join(_f_bytes(1, n) for n in self.nodes)
        g += _f_bytes(2, self.name)
        g += b"".join(_f_bytes(5, t) for t in self.inits)
        g += b"".join(_f_bytes(11, v) for v in self.inputs)
        g += b"".join(_f_bytes(12, v) for v in self.outputs)
        m = _f_varint(1, 6) + _f_bytes(2, "pytorch") + _f_bytes(3, "1.13.1")
        m += _f_bytes(7, g)
        m += _f_bytes(8, _f_bytes(1, "") + _f_varint(2, opset))
        return m


# ------------------------------------------------------------------ architecture
APRILV0_DIMS = dict(n_layers=12, d_model=512, hidden=1024, ffn=2048, joiner=512, vocab=500, mel=80, seg=9,
                    context=2, dec_groups=128, conv_ch=(8, 32, 128))
TINY_DIMS = dict(n_layers=2, d_model=64, hidden=128, ffn=128, joiner=64, vocab=40, mel=80, seg=9,
                 context=2, dec_groups=16, conv_ch=(8, 16, 32))
# odd-sized but legal: 3 layers, widths that are multiples of 64 but not powers of two, a vocabulary that is not a
# multiple of 16 (exercises the padded joiner columns), a different decoder grouping
MEDIUM_DIMS = dict(n_layers=3, d_model=192, hidden=320, ffn=448, joiner=192, vocab=131, mel=80, seg=9,
                   context=2, dec_groups=48, conv_ch=(8, 24, 64))
# widths that are multiples of 16 but not of 64 (the loader pads them: csrc/model_loader.cc pad_host_model)
NARROW_DIMS = dict(n_layers=2, d_model=144, hidden=208, ffn=304, joiner=80, vocab=60, mel=80, seg=9,
                   context=2, dec_groups=36, conv_ch=(8, 16, 48))
# widths that are not even multiples of 16 (the zero padding does not care): d 100, cell 150, ffn 210, joiner 70, 20 conv channels
ODD_DIMS = dict(n_layers=2, d_model=100, hidden=150, ffn=210, joiner=70, vocab=45, mel=80, seg=9,
                context=2, dec_groups=25, conv_ch=(8, 12, 20))
LARGE_DIMS = dict(n_layers=16, d_model=768, hidden=1536, ffn=3072, joiner=768, vocab=500, mel=80, seg=9,
                  context=2, dec_groups=192, conv_ch=(8, 32, 128))


def _uniform(rng, shape, fan_in, gain=1.0):
    b = gain / np.sqrt(float(fan_in))
    return rng.uniform(-b, b, size=shape).astype(np.float32)


def make_weights(dims, seed=2023, joiner_gain=6.0, blank_bias=4.7, blank_id=0):
    """Seeded weights in PyTorch parameter layout (what the exporter would start from)."""
    rng = np.random.RandomState(seed)
    d, H, F, J, V = dims["d_model"], dims["hidden"], dims["ffn"], dims["joiner"], dims["vocab"]
    c1, c2, c3 = dims["conv_ch"]
    mel = dims["mel"]
    f_out = ((mel - 3) // 2 - 1) // 2
    w = {}
    w["conv0.w"] = _uniform(rng, (c1, 1, 3, 3), 9);          w["conv0.b"] = _uniform(rng, (c1,), 9)
    w["conv1.w"] = _uniform(rng, (c2, c1, 3, 3), c1 * 9);    w["conv1.b"] = _uniform(rng, (c2,), c1 * 9)
    w["conv2.w"] = _uniform(rng, (c3, c2, 3, 3), c2 * 9);    w["conv2.b"] = _uniform(rng, (c3,), c2 * 9)
    w["embed.w"] = _uniform(rng, (d, c3 * f_out), c3 * f_out); w["embed.b"] = _uniform(rng, (d,), c3 * f_out)
    w["embed.eps"] = np.float32(0.25)
    for l in range(dims["n_layers"]):
        p = "l%d." % l
        w[p + "w_ih"] = _uniform(rng, (4 * H, d), H); w[p + "b_ih"] = _uniform(rng, (4 * H,), H)
        w[p + "w_hh"] = _uniform(rng, (4 * H, d), H); w[p + "b_hh"] = _uniform(rng, (4 * H,), H)
        w[p + "w_hr"] = _uniform(rng, (d, H), H)
        w[p + "ff1.w"] = _uniform(rng, (F, d), d);    w[p + "ff1.b"] = _uniform(rng, (F,), d)
        w[p + "ff2.w"] = _uniform(rng, (d, F), F, 0.25); w[p + "ff2.b"] = _uniform(rng, (d,), F, 0.25)
        w[p + "eps"] = np.float32(rng.uniform(-0.5, 0.5))
    w["enc_proj.w"] = _uniform(rng, (J, d), d); w["enc_proj.b"] = _uniform(rng, (J,), d)
    w["emb"] = rng.normal(0, 1.0, size=(V, d)).astype(np.float32)
    g = dims["dec_groups"]
    w["dec_conv.w"] = _uniform(rng, (d, d // g, dims["context"]), (d // g) * dims["context"])
    w["dec_proj.w"] = _uniform(rng, (J, d), d); w["dec_proj.b"] = _uniform(rng, (J,), d)
    w["out.w"] = _uniform(rng, (V, J), J, joiner_gain); w["out.b"] = _uniform(rng, (V,), J)
    w["out.b"][blank_id] += np.float32(blank_bias)
    return w


# Spelling knobs (the `variant` dict of write_model / build_*; every combination describes the SAME network and must load
# to the SAME packed weights, tests/test_loader.py):
#   lstm_gemm   True: gate products as Gemm(transB=1) | False: MatMul (+ Add)
#   gemm_bias   "both" | "separate" (bias added by an Add node after each product) | "single" (b_ih + b_hh folded onto the
#               input product) | "after_sum" (b_ih + b_hh added after the two products were summed)
#   gate_split  "split" | "slice" (four Slice nodes with constant inputs, opset >= 10) | "slice_attr" (opset < 10 attributes)
#   gate_order  a permutation of "ifgo": order of the gate blocks in the weight rows (torch.nn.LSTM: "ifgo")
#   fold_eps    BasicNorm epsilon as a folded constant | Exp(initializer)
#   norm        "pow" | "mulself" (x * x) | "sqrt_recip" (Reciprocal(Sqrt)) | "div_sqrt" (x / Sqrt)
#   swish       "sub" (x - 1) | "add_neg" (x + (-1))
#   passthrough Identity / Cast / Dropout nodes on the activation path and in front of constants
#   const_nodes small tensors and the conv weights as Constant nodes instead of initializers
#   w_transpose Linear weights stored [out, in] behind a Transpose node (constant folding switched off)
#   state_index "slice" | "gather" (per-layer state through Gather + Unsqueeze)
_V = {}


def _variant(variant):
    _V.clear()
    _V.update(variant or {})


def _pt(g, x, kind="Identity"):
    """optional value-preserving node"""
    if not _V.get("passthrough"):
        return x
    if kind == "Cast":
        return g.node("Cast", [x], to=1)
    if kind == "Dropout":
        return g.node("Dropout", [x], ratio=0.1)
    return g.node("Identity", [x])


def _small(g, name, arr):
    """bias-sized constants: initializer, or Constant node"""
    if _V.get("const_nodes"):
        return g.const(np.asarray(arr))
    c = g.init(name, np.asarray(arr))
    return g.node("Identity", [c]) if _V.get("passthrough") else c


def _double_swish(g, x):
    if _V.get("swish", "sub") == "add_neg":
        shifted = g.node("Add", [x, g.const(np.array(-1.0, np.float32))])
    else:
        shifted = g.node("Sub", [x, g.const(np.array(1.0, np.float32))])
    return g.node("Mul", [x, g.node("Sigmoid", [shifted])])


def _basic_norm(g, x, eps_log, name, fold):
    how = _V.get("norm", "pow")
    sq = g.node("Mul", [x, x]) if how == "mulself" else g.node("Pow", [x, g.const(np.array(2.0, np.float32))])
    mean = g.node("ReduceMean", [sq], axes=[-1], keepdims=1)
    if fold:
        e = g.const(np.array(np.exp(np.float32(eps_log)), np.float32))
    else:
        e = g.node("Exp", [g.init(name + ".eps", np.array(eps_log, np.float32))])
    v = g.node("Add", [mean, e])
    if how == "sqrt_recip":
        return g.node("Mul", [x, g.node("Reciprocal", [g.node("Sqrt", [v])])])
    if how == "div_sqrt":
        return g.node("Div", [x, g.node("Sqrt", [v])])
    scale = g.node("Pow", [v, g.const(np.array(-0.5, np.float32))])
    return g.node("Mul", [x, scale])


def _wconst(g, name, w_out_in):
    """a Linear weight given [out, in] as the [in, out] operand of a MatMul"""
    if _V.get("w_transpose"):
        return g.node("Transpose", [g.init(name, np.ascontiguousarray(w_out_in))], perm=[1, 0])
    c = g.init("onnx::MatMul_" + name, np.ascontiguousarray(w_out_in.T))
    return g.node("Identity", [c]) if _V.get("passthrough") else c


def _linear3d(g, x, w, b, name):
    """nn.Linear on a 3-D input: MatMul with the transposed weight as initializer, then Add(bias)."""
    y = g.node("MatMul", [x, _wconst(g, name, w)])
    return g.node("Add", [_small(g, name + ".bias", b), y]) if b is not None else y


def build_encoder(dims, w, variant):
    _variant(variant)
    g = GraphBuilder("torch_jit")
    L, d, H = dims["n_layers"], dims["d_model"], dims["hidden"]
    T, mel = dims["seg"], dims["mel"]
    g.inputs = [value_info("x", FLOAT, (1, T, mel)), value_info("h", FLOAT, (L, 1, d)), value_info("c", FLOAT, (L, 1, H))]
    g.outputs = [value_info("encoder_out", FLOAT, (1, 1, dims["joiner"])), value_info("next_h", FLOAT, (L, 1, d)),
                 value_info("next_c", FLOAT, (L, 1, H))]
    fold = variant.get("fold_eps", True)
    t = g.node("Unsqueeze", ["x"], axes=[1])
    for i, stride in enumerate((1, 2, 2)):
        attrs = dict(dilations=[1, 1], group=1, kernel_shape=[3, 3], pads=[0, 0, 0, 0], strides=[stride, stride])
        cw = g.const(w["conv%d.w" % i]) if _V.get("const_nodes") else g.init("encoder.encoder_embed.conv.%d.weight" % (3 * i), w["conv%d.w" % i])
        t = g.node("Conv", [t, cw, _small(g, "encoder.encoder_embed.conv.%d.bias" % (3 * i), w["conv%d.b" % i])], **attrs)
        t = _double_swish(g, _pt(g, t))
    c3 = dims["conv_ch"][2]
    f_out = ((mel - 3) // 2 - 1) // 2
    t_out = ((T - 3) // 2 - 1) // 2
    t = g.node("Transpose", [t], perm=[0, 2, 1, 3])
    t = g.node("Reshape", [t, g.const(np.array([1, t_out, c3 * f_out], np.int64))])
    t = _linear3d(g, t, w["embed.w"], w["embed.b"], "encoder.encoder_embed.out")
    t = _basic_norm(g, t, w["embed.eps"], "encoder.encoder_embed.out_norm", True)
    src = g.node("Transpose", [t], perm=[1, 0, 2])
    new_h, new_c = [], []
    for l in range(L):
        p = "l%d." % l
        nm = "encoder.encoder.layers.%d" % l
        def sl(inp):
            if _V.get("state_index", "slice") == "gather":
                return g.node("Unsqueeze", [g.node("Gather", [inp, g.const(np.array(l, np.int64))], axis=0)], axes=[0])
            return g.node("Slice", [inp, g.const(np.array([l], np.int64)), g.const(np.array([l + 1], np.int64)),
                                    g.const(np.array([0], np.int64)), g.const(np.array([1], np.int64))])
        h_l, c_l = sl("h"), sl("c")
        x_t = g.node("Gather", [src, g.const(np.array(0, np.int64))], axis=0)
        h0 = g.node("Squeeze", [h_l], axes=[0])
        c0 = _pt(g, g.node("Squeeze", [c_l], axes=[0]))
        # gate blocks in the order the variant asks for (rows of the [4H, d] weights)
        order = _V.get("gate_order", "ifgo")
        blk = ["ifgo".index(ch) for ch in order]
        def perm_rows(a):
            return np.ascontiguousarray(np.concatenate([a[k * H:(k + 1) * H] for k in blk], 0))
        w_ih, w_hh, b_ih, b_hh = perm_rows(w[p + "w_ih"]), perm_rows(w[p + "w_hh"]), perm_rows(w[p + "b_ih"]), perm_rows(w[p + "b_hh"])
        bias_mode = _V.get("gemm_bias", "both")
        b_sum = (b_ih + b_hh).astype(np.float32)
        def product(x_in, wt, b, tag):
            bb = {"both": b, "separate": None, "single": (b_sum if tag == "ih" else None), "after_sum": None}[bias_mode]
            if _V.get("lstm_gemm", True):
                ins = [x_in, g.init(nm + ".lstm.w_" + tag, wt)] + ([_small(g, nm + ".lstm.b_" + tag, bb)] if bb is not None else [])
                y = g.node("Gemm", ins, alpha=1.0, beta=1.0, transB=1)
            else:
                y = g.node("MatMul", [x_in, g.init(nm + ".lstm.w_" + tag + "_t", np.ascontiguousarray(wt.T))])
                if bb is not None:
                    y = g.node("Add", [y, _small(g, nm + ".lstm.b_" + tag, bb)])
            if bias_mode == "separate":
                y = g.node("Add", [y, _small(g, nm + ".lstm.b_" + tag, b)])
            return y
        g1 = product(x_t, w_ih, b_ih, "ih")
        g2 = product(h0, w_hh, b_hh, "hh")
        gates = g.node("Add", [g1, g2])
        if bias_mode == "after_sum":
            gates = g.node("Add", [gates, _small(g, nm + ".lstm.b_sum", b_sum)])
        gates = _pt(g, gates)
        how = _V.get("gate_split", "split")
        if how == "split":
            parts = g.node("Split", [gates], n_out=4, axis=1, split=[H] * 4)
        elif how == "slice":
            parts = [g.node("Slice", [gates, g.const(np.array([k * H], np.int64)), g.const(np.array([(k + 1) * H], np.int64)),
                                      g.const(np.array([1], np.int64)), g.const(np.array([1], np.int64))]) for k in range(4)]
        else:
            parts = [g.node("Slice", [gates], axes=[1], starts=[k * H], ends=[(k + 1) * H]) for k in range(4)]
        byname = {ch: parts[k] for k, ch in enumerate(order)}
        si, sf, tg, so = g.node("Sigmoid", [byname["i"]]), g.node("Sigmoid", [byname["f"]]), g.node("Tanh", [byname["g"]]), g.node("Sigmoid", [byname["o"]])
        c1 = g.node("Add", [g.node("Mul", [sf, c0]), g.node("Mul", [si, tg])])
        hh = g.node("Mul", [so, g.node("Tanh", [c1])])
        h1 = g.node("MatMul", [hh, _wconst(g, nm + ".lstm.w_hr", w[p + "w_hr"])])
        ys = g.node("Concat", [g.node("Unsqueeze", [h1], axes=[0])], axis=0)
        new_h.append(g.node("Unsqueeze", [h1], axes=[0]))
        new_c.append(g.node("Unsqueeze", [c1], axes=[0]))
        src1 = _pt(g, g.node("Add", [ys, src]), "Cast")
        ff = _linear3d(g, src1, w[p + "ff1.w"], w[p + "ff1.b"], nm + ".feed_forward.0")
        ff = _pt(g, _double_swish(g, ff), "Dropout")
        ff = _linear3d(g, ff, w[p + "ff2.w"], w[p + "ff2.b"], nm + ".feed_forward.4")
        src2 = g.node("Add", [src1, _pt(g, ff, "Dropout")])
        src = _basic_norm(g, src2, w[p + "eps"], nm + ".norm_final", fold)
    g.node_named("Concat", new_h, ["next_h"], axis=0)
    g.node_named("Concat", new_c, ["next_c"], axis=0)
    t = g.node("Transpose", [src], perm=[1, 0, 2])
    y = g.node("MatMul", [t, _wconst(g, "encoder_proj", w["enc_proj.w"])])
    if _V.get("passthrough"):
        g.node_named("Identity", [g.node("Add", [_small(g, "encoder_proj.bias", w["enc_proj.b"]), y])], ["encoder_out"])
    else:
        g.node_named("Add", [_small(g, "encoder_proj.bias", w["enc_proj.b"]), y], ["encoder_out"])
    return g.model_bytes()


def build_decoder(dims, w, variant):
    _variant(variant)
    g = GraphBuilder("torch_jit")
    d, J, ctx = dims["d_model"], dims["joiner"], dims["context"]
    g.inputs = [value_info("context", INT64, (1, ctx))]
    g.outputs = [value_info("decoder_out", FLOAT, (1, 1, J))]
    e = g.node("Gather", [g.init("decoder.embedding.weight", w["emb"]), "context"], axis=0)
    e = g.node("Transpose", [e], perm=[0, 2, 1])
    e = g.node("Conv", [e, g.init("decoder.conv.weight", w["dec_conv.w"])], dilations=[1], group=dims["dec_groups"],
               kernel_shape=[ctx], pads=[0, 0], strides=[1])
    e = g.node("Transpose", [e], perm=[0, 2, 1])
    e = g.node("Relu", [e])
    y = g.node("MatMul", [_pt(g, e), _wconst(g, "decoder_proj", w["dec_proj.w"])])
    g.node_named("Add", [_small(g, "decoder_proj.bias", w["dec_proj.b"]), y], ["decoder_out"])
    return g.model_bytes()


def build_joiner(dims, w, variant):
    _variant(variant)
    g = GraphBuilder("torch_jit")
    J, V = dims["joiner"], dims["vocab"]
    g.inputs = [value_info("encoder_out", FLOAT, (1, 1, J)), value_info("decoder_out", FLOAT, (1, 1, J))]
    g.outputs = [value_info("logits", FLOAT, (1, 1, V))]
    t = g.node("Tanh", [g.node("Add", ["encoder_out", "decoder_out"])])
    y = g.node("MatMul", [_pt(g, t), _wconst(g, "output_linear", w["out.w"])])
    g.node_named("Add", [_small(g, "output_linear.bias", w["out.b"]), y], ["logits"])
    return g.model_bytes()


# ------------------------------------------------------------------ vocabulary + params + container
def make_tokens(vocab, seed=7, punctuation=True):
    """Deterministic BPE-like vocabulary. id 0 = <blk>; optional punctuation and digits so the
    reference's punctuation / digit-dot branches (april_session.c:340-358) are reachable."""
    rng = np.random.RandomState(seed)
    toks = ["<blk>"]
    special = [".", ",", "?", "!", " 1", "2", " 3", "4"] if punctuation else []
    for s in special:
        if len(toks) < vocab:
            toks.append(s)
    letters = "etaoinshrdlucmfwypvbgkqjxz"
    seen = set(toks)
    while len(toks) < vocab:
        n = int(rng.randint(1, 5))
        s = "".join(letters[int(i)] for i in rng.randint(0, len(letters), size=n))
        if rng.rand() < 0.45:
            s = " " + s
        if s in seen:
            continue
        seen.add(s)
        toks.append(s)
    return toks


def params_block(dims, tokens, blank_id=0, rate=16000, step=4, shift_ms=10, length_ms=25, round_pow2=1, mel_low=20,
                 mel_high=0):
    b = io.BytesIO()
    b.write(b"PARAMS\0\0")
    for v in (1, dims["seg"], step, dims["mel"], rate, shift_ms, length_ms, round_pow2, mel_low, mel_high, 0,
              len(tokens), blank_id):
        b.write(struct.pack("<i", int(v)))
    for t in tokens:
        raw = t.encode("utf-8")
        b.write(struct.pack("<i", len(raw)))
        b.write(raw)
    return b.getvalue()


def container_bytes(networks, params, name="synthetic", description="seeded synthetic weights", language="en-us",
                    model_type=1, version=1):
    """Layout of extra/file-format.md: magic, version, header_size, header, networks..., params."""
    lang = language.encode("utf-8").ljust(8, b"\0")[:8]
    nm, ds = name.encode("utf-8"), description.encode("utf-8")
    header_len = 8 + 8 + len(nm) + 8 + len(ds) + 4 + 16 + 8 + 16 * len(networks)
    base = 8 + 4 + 8 + header_len
    offs = []
    pos = base
    for n in networks:
        offs.append(pos)
        pos += len(n)
    params_off = pos
    h = io.BytesIO()
    h.write(lang)
    h.write(struct.pack("<Q", len(nm))); h.write(nm)
    h.write(struct.pack("<Q", len(ds))); h.write(ds)
    h.write(struct.pack("<I", model_type))
    h.write(struct.pack("<QQ", params_off, len(params)))
    h.write(struct.pack("<Q", len(networks)))
    for o, n in zip(offs, networks):
        h.write(struct.pack("<QQ", o, len(n)))
    hb = h.getvalue()
    assert len(hb) == header_len
    out = io.BytesIO()
    out.write(b"APRILMDL")
    out.write(struct.pack("<I", version))
    out.write(struct.pack("<Q", header_len))
    out.write(hb)
    for n in networks:
        out.write(n)
    out.write(params)
    return out.getvalue()


def write_model(path, dims=None, seed=2023, variant=None, punctuation=True, joiner_gain=6.0, blank_bias=4.7,
                name=None, params=None):
    """Write a synthetic .april file; returns (dims, weights, tokens).  `params`: keyword overrides of the PARAMS block
    (rate, shift_ms, length_ms, round_pow2, mel_low, mel_high -- e.g. round_pow2=0 for a 400-point FFT at 16 kHz / 25 ms)."""
    dims = dict(dims or APRILV0_DIMS)
    variant = dict(variant or {})
    w = make_weights(dims, seed=seed, joiner_gain=joiner_gain, blank_bias=blank_bias)
    toks = make_tokens(dims["vocab"], punctuation=punctuation)
    nets = [build_encoder(dims, w, variant), build_decoder(dims, w, variant), build_joiner(dims, w, variant)]
    blob = container_bytes(nets, params_block(dims, toks, **dict(params or {})), name=name or ("synthetic-L%d-d%d" % (dims["n_layers"], dims["d_model"])))
    with open(path, "wb") as f:
        f.write(blob)
    return dims, w, toks


def lcg_pcm16(n, seed=12345):
    """Seeded white-noise PCM16: s = s*1664525 + 1013904223 (mod 2^32), sample = (int16)(s >> 16).
    Vectorised: A[k] = a^k and S[k] = 1 + a + ... + a^(k-1) built by doubling, x_k = A[k] x_0 + c S[k]."""
    a, c = 1664525, 1013904223
    M = np.uint64(0xFFFFFFFF)
    A = np.array([a], np.uint64); S = np.array([1], np.uint64)
    while A.size < n:
        am, sm = A[-1], S[-1]
        A = np.concatenate([A, (am * A) & M])
        S = np.concatenate([S, (sm + ((am * S) & M)) & M])
    A, S = A[:n], S[:n]
    x = (((A * np.uint64(seed)) & M) + ((np.uint64(c) * S) & M)) & M
    return ((x >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.uint16).view(np.int16)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="write a synthetic .april model")
    ap.add_argument("out")
    ap.add_argument("--dims", default="aprilv0", choices=["aprilv0", "tiny", "large"])
    ap.add_argument("--seed", type=int, default=2023)
    a = ap.parse_args()
    write_model(a.out, {"aprilv0": APRILV0_DIMS, "tiny": TINY_DIMS, "large": LARGE_DIMS}[a.dims], seed=a.seed)
    print("wrote", a.out)
