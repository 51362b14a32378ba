"""Weight extraction + MFMA packing (host-only load, no GPU): the packed blob the engine uploads must
contain exactly the weights the synthetic exporter wrote, whatever ONNX spelling was used."""
import os
import struct

import numpy as np
import pytest

import april_asr_amd as A
from april_asr_amd import synth_model as SM


def split_blob(blob):
    magic, meta_bytes, wfloats, woff = struct.unpack("<8sQQQ", blob[:32].tobytes())
    assert magic == b"APXBLOB2"
    return np.frombuffer(blob[woff:woff + wfloats * 4].tobytes(), np.float32)


def unpack_mfma(p, K, N, Npad):
    """inverse of the kernel layout: Wp[((t*KB+kb)*64+lane)*4+j] = W[kb*16+(lane>>4)*4+j][t*16+(lane&15)]"""
    KB, NT = K // 16, Npad // 16
    a = p[:Npad * K].reshape(NT, KB, 64, 4)
    W = np.zeros((K, Npad), np.float32)
    lane = np.arange(64)
    for j in range(4):
        k = (np.arange(KB)[:, None] * 16 + (lane >> 4)[None, :] * 4 + j)            # [KB,64]
        n = (np.arange(NT)[:, None, None] * 16 + (lane & 15)[None, None, :])        # [NT,1,64]
        W[np.broadcast_to(k[None], (NT, KB, 64)), np.broadcast_to(n, (NT, KB, 64))] = a[..., j]
    return W[:, :N]


def layout_offsets(d):
    """mirror of plan_layout (engine.cc): 64-float aligned sections in a fixed order"""
    off = [0]

    def take(n):
        o = off[0]
        off[0] = (o + n + 63) & ~63
        return o
    L = {}
    c0, c1, c2 = d["conv_ch"]
    k3 = (c1 * 9 + 63) & ~63
    L["conv_w0"] = take(c0 * 9); L["conv_b0"] = take(c0)
    L["conv_w1"] = take(c1 * c0 * 9); L["conv_b1"] = take(c1)
    L["conv_w2"] = take(k3 * c2); L["conv_b2"] = take(c2)
    L["k3"] = k3
    f_out = ((d["mel"] - 3) // 2 - 1) // 2
    ein = d["conv_ch"][2] * f_out
    L["w_embed"] = take(ein * d["d_model"]); L["b_embed"] = take(d["d_model"])
    for l in range(d["n_layers"]):
        L["wg%d" % l] = take(2 * d["d_model"] * 4 * d["hidden"]); L["bg%d" % l] = take(4 * d["hidden"])
        L["whr%d" % l] = take(d["hidden"] * d["d_model"])
        L["wff1%d" % l] = take(d["d_model"] * d["ffn"]); L["bff1%d" % l] = take(d["ffn"])
        L["wff2%d" % l] = take(d["ffn"] * d["d_model"]); L["bff2%d" % l] = take(d["d_model"])
    L["w_encproj"] = take(d["d_model"] * d["joiner"]); L["b_encproj"] = take(d["joiner"])
    L["emb"] = take(d["vocab"] * d["d_model"])
    L["dec_conv"] = take(d["d_model"] * (d["d_model"] // d["dec_groups"]) * d["context"]); L["dec_conv_b"] = take(d["d_model"])
    L["w_decproj"] = take(d["d_model"] * d["joiner"]); L["b_decproj"] = take(d["joiner"])
    vp = (d["vocab"] + 31) & ~31
    L["w_out"] = take(d["joiner"] * vp); L["b_out"] = take(vp)
    return L, ein, vp


def check_blob(w, d, wts):
    L, ein, vp = layout_offsets(d)
    D, H, F, J, V = d["d_model"], d["hidden"], d["ffn"], d["joiner"], d["vocab"]
    for i in range(3):
        if i < 2:
            assert np.array_equal(w[L["conv_w%d" % i]:][:wts["conv%d.w" % i].size], wts["conv%d.w" % i].ravel())
        assert np.array_equal(w[L["conv_b%d" % i]:][:wts["conv%d.b" % i].size], wts["conv%d.b" % i])
    c0, c1, c2 = d["conv_ch"]
    w3 = unpack_mfma(w[L["conv_w2"]:], L["k3"], c2, c2)                # third conv as [k = ci*9+i*3+j][out channel]
    assert np.array_equal(w3[:c1 * 9], wts["conv2.w"].reshape(c2, c1 * 9).T) and not w3[c1 * 9:].any()
    f_out = ein // c2
    we = unpack_mfma(w[L["w_embed"]:], ein, D, D)                      # rows permuted to position-major (f*c2 + c)
    assert np.array_equal(we.reshape(f_out, c2, D), wts["embed.w"].T.reshape(c2, f_out, D).transpose(1, 0, 2))
    assert np.array_equal(w[L["b_embed"]:][:D], wts["embed.b"])
    for l in range(d["n_layers"]):
        p = "l%d." % l
        Wg = unpack_mfma(w[L["wg%d" % l]:], 2 * D, 4 * H, 4 * H)                 # columns unit-major: u*4+g
        src = np.concatenate([wts[p + "w_ih"].T, wts[p + "w_hh"].T], 0)           # [2D][4H], gate-major
        want = src.reshape(2 * D, 4, H).transpose(0, 2, 1).reshape(2 * D, 4 * H)
        assert np.array_equal(Wg, want)
        bsum = (wts[p + "b_ih"] + wts[p + "b_hh"]).reshape(4, H).T.ravel()
        assert np.array_equal(w[L["bg%d" % l]:][:4 * H], bsum)
        assert np.array_equal(unpack_mfma(w[L["whr%d" % l]:], H, D, D), wts[p + "w_hr"].T)
        assert np.array_equal(unpack_mfma(w[L["wff1%d" % l]:], D, F, F), wts[p + "ff1.w"].T)
        assert np.array_equal(unpack_mfma(w[L["wff2%d" % l]:], F, D, D), wts[p + "ff2.w"].T)
        assert np.array_equal(w[L["bff1%d" % l]:][:F], wts[p + "ff1.b"]) and np.array_equal(w[L["bff2%d" % l]:][:D], wts[p + "ff2.b"])
    assert np.array_equal(unpack_mfma(w[L["w_encproj"]:], D, J, J), wts["enc_proj.w"].T)
    assert np.array_equal(w[L["emb"]:][:V * D], wts["emb"].ravel())
    assert np.array_equal(w[L["dec_conv"]:][:wts["dec_conv.w"].size], wts["dec_conv.w"].ravel())
    assert np.array_equal(unpack_mfma(w[L["w_decproj"]:], D, J, J), wts["dec_proj.w"].T)
    out = unpack_mfma(w[L["w_out"]:], J, vp, vp)
    assert np.array_equal(out[:, :V], wts["out.w"].T) and not out[:, V:].any()
    assert np.array_equal(w[L["b_out"]:][:V], wts["out.b"])


def test_extraction_and_packing(built, tiny_model):
    m = A.Model.load_host_only(tiny_model["path"])
    d = tiny_model["dims"]
    assert (m.dims.n_layers, m.dims.d_model, m.dims.hidden, m.dims.ffn, m.dims.joiner, m.dims.vocab) == \
           (d["n_layers"], d["d_model"], d["hidden"], d["ffn"], d["joiner"], d["vocab"])
    assert (m.dims.seg, m.dims.seg_step, m.dims.mel, m.dims.context, m.dims.fft_size, m.dims.frame_shift) == (9, 4, 80, 2, 512, 160)
    check_blob(split_blob(m.export_blob()), d, tiny_model["weights"])
    m.close()


def test_widths_that_are_multiples_of_16_are_padded_to_64(built, narrow_model):
    """csrc/model_loader.cc pad_host_model: d 144, cell 208, ffn 304, joiner 80, 48 conv-3 channels -> 192 / 256 / 320 / 128 / 64.
    Every real weight sits where the padded network expects it, every padded row / column / bias is zero, the decoder convolution
    grows by whole groups, and the BasicNorm mean keeps the file's width (dims.d_model_file)."""
    m = A.Model.load_host_only(narrow_model["path"])
    d0, wts = narrow_model["dims"], narrow_model["weights"]
    assert (m.dims.d_model, m.dims.hidden, m.dims.ffn, m.dims.joiner, m.dims.vocab, m.dims.d_model_file) == (192, 256, 320, 128, 60, 144)
    D0, H0, F0, J0, V = d0["d_model"], d0["hidden"], d0["ffn"], d0["joiner"], d0["vocab"]
    c0, c1, c20 = d0["conv_ch"]
    d = dict(d0, d_model=192, hidden=256, ffn=320, joiner=128, conv_ch=(c0, c1, 64), dec_groups=192 // (D0 // d0["dec_groups"]))
    D, H, F, J = 192, 256, 320, 128
    w = split_blob(m.export_blob())
    L, ein, vp = layout_offsets(d)

    def padded(src, rows, cols):
        out = np.zeros((rows, cols), np.float32)
        out[:src.shape[0], :src.shape[1]] = src
        return out
    w3 = unpack_mfma(w[L["conv_w2"]:], L["k3"], 64, 64)
    assert np.array_equal(w3[:c1 * 9], padded(wts["conv2.w"].reshape(c20, c1 * 9).T, c1 * 9, 64)) and not w3[c1 * 9:].any()
    assert np.array_equal(w[L["conv_b2"]:][:64], np.concatenate([wts["conv2.b"], np.zeros(16, np.float32)]))
    f_out = ein // 64
    we = unpack_mfma(w[L["w_embed"]:], ein, D, D).reshape(f_out, 64, D)                 # position-major rows (f * c2 + c)
    src = wts["embed.w"].T.reshape(c20, f_out, D0).transpose(1, 0, 2)
    assert np.array_equal(we[:, :c20, :D0], src) and not we[:, c20:, :].any() and not we[:, :, D0:].any()
    assert np.array_equal(w[L["b_embed"]:][:D], np.concatenate([wts["embed.b"], np.zeros(D - D0, np.float32)]))
    for l in range(d["n_layers"]):
        p = "l%d." % l
        Wg = unpack_mfma(w[L["wg%d" % l]:], 2 * D, 4 * H, 4 * H).reshape(2, D, H, 4)      # [x | h][row][unit][gate]
        for half, key in enumerate(("w_ih", "w_hh")):
            want = wts[p + key].T.reshape(D0, 4, H0).transpose(0, 2, 1)                   # [row][unit][gate]
            assert np.array_equal(Wg[half, :D0, :H0], want) and not Wg[half, D0:].any() and not Wg[half, :, H0:].any()
        bg = w[L["bg%d" % l]:][:4 * H].reshape(H, 4)
        assert np.array_equal(bg[:H0], (wts[p + "b_ih"] + wts[p + "b_hh"]).reshape(4, H0).T) and not bg[H0:].any()
        assert np.array_equal(unpack_mfma(w[L["whr%d" % l]:], H, D, D), padded(wts[p + "w_hr"].T, H, D))
        assert np.array_equal(unpack_mfma(w[L["wff1%d" % l]:], D, F, F), padded(wts[p + "ff1.w"].T, D, F))
        assert np.array_equal(unpack_mfma(w[L["wff2%d" % l]:], F, D, D), padded(wts[p + "ff2.w"].T, F, D))
        assert np.array_equal(w[L["bff1%d" % l]:][:F], np.concatenate([wts[p + "ff1.b"], np.zeros(F - F0, np.float32)]))
        assert np.array_equal(w[L["bff2%d" % l]:][:D], np.concatenate([wts[p + "ff2.b"], np.zeros(D - D0, np.float32)]))
    assert np.array_equal(unpack_mfma(w[L["w_encproj"]:], D, J, J), padded(wts["enc_proj.w"].T, D, J))
    assert np.array_equal(w[L["emb"]:][:V * D].reshape(V, D), padded(wts["emb"], V, D))
    gs = D0 // d0["dec_groups"]
    dc = w[L["dec_conv"]:][:D * gs * d["context"]].reshape(D, gs * d["context"])
    assert np.array_equal(dc[:D0], wts["dec_conv.w"].reshape(D0, -1)) and not dc[D0:].any()
    assert np.array_equal(unpack_mfma(w[L["w_decproj"]:], D, J, J), padded(wts["dec_proj.w"].T, D, J))
    out = unpack_mfma(w[L["w_out"]:], J, vp, vp)
    assert np.array_equal(out[:J0, :V], wts["out.w"].T) and not out[J0:].any() and not out[:, V:].any()
    m.close()


def test_alternative_onnx_spelling_gives_identical_blob(built, tiny_model, tiny_model_variant):
    """MatMul+Add instead of Gemm for the LSTM gates, BasicNorm eps as Exp(initializer)."""
    a = A.Model.load_host_only(tiny_model["path"]); b = A.Model.load_host_only(tiny_model_variant["path"])
    ba, bb = a.export_blob(), b.export_blob()
    assert np.array_equal(split_blob(ba), split_blob(bb))
    a.close(); b.close()


EXPORTER_SPELLINGS = [
    dict(gemm_bias="separate"),
    dict(gemm_bias="single"),
    dict(gemm_bias="after_sum"),
    dict(lstm_gemm=False, gemm_bias="separate"),
    dict(lstm_gemm=False, gemm_bias="after_sum", gate_split="slice"),
    dict(gate_split="slice"),
    dict(gate_split="slice_attr"),
    dict(gate_order="fiog"),
    dict(gate_order="gofi", gate_split="slice", gemm_bias="single"),
    dict(passthrough=True),
    dict(passthrough=True, const_nodes=True, fold_eps=False),
    dict(const_nodes=True),
    dict(w_transpose=True),
    dict(norm="mulself"),
    dict(norm="sqrt_recip", swish="add_neg"),
    dict(norm="div_sqrt", state_index="gather"),
    dict(state_index="gather", passthrough=True, w_transpose=True, gate_split="slice", gate_order="oifg", gemm_bias="after_sum", lstm_gemm=False),
]


@pytest.mark.parametrize("variant", EXPORTER_SPELLINGS, ids=lambda v: ",".join("%s=%s" % kv for kv in sorted(v.items())))
def test_exporter_spellings_give_identical_blob(built, tiny_model, tmp_path, variant):
    """What an ONNX exporter / constant folder may spell freely (Gemm vs MatMul+Add, where the LSTM biases sit, Split vs
    Slice, the ORDER of the gate blocks -- the loader reads each gate's role off the cell update --, constants as
    initializers / Constant nodes / behind Identity, Cast or Transpose, Identity / Cast / Dropout on the activation path,
    the spelling of BasicNorm and DoubleSwish, per-layer state through Slice or Gather) loads to the SAME packed weights."""
    p = str(tmp_path / "variant.april")
    SM.write_model(p, SM.TINY_DIMS, variant=variant)
    a = A.Model.load_host_only(tiny_model["path"]); b = A.Model.load_host_only(p)
    assert np.array_equal(split_blob(a.export_blob()), split_blob(b.export_blob()))
    a.close(); b.close()


def test_rejection_names_the_offending_node(built, tmp_path, capfd):
    """An unsupported graph is refused with a message that says which node the loader stopped at (op, name, inputs)."""
    import april_asr_amd.synth_model as S
    dims = dict(S.TINY_DIMS)
    w = S.make_weights(dims); toks = S.make_tokens(dims["vocab"])
    enc = S.build_encoder(dims, w, {}).replace(b"\x22\x05Split", b"\x22\x05Spliz", 1)      # op_type of the first layer's Split
    blob = S.container_bytes([enc, S.build_decoder(dims, w, {}), S.build_joiner(dims, w, {})], S.params_block(dims, toks))
    p = tmp_path / "bad.april"; p.write_bytes(blob)
    with pytest.raises(Exception):
        A.Model.load_host_only(str(p))
    msg = capfd.readouterr().err
    assert "encoder layer 0" in msg and "Spliz" in msg and "inputs:" in msg


def test_blob_roundtrip_host(built, tiny_model):
    a = A.Model.load_host_only(tiny_model["path"])
    blob = a.export_blob()
    b = A.Model.from_blob(blob, init_gpu=False)
    assert np.array_equal(blob, b.export_blob())
    assert b.get_name() == a.get_name() and b.dims.param_count == a.dims.param_count
    assert [b.token(i) for i in range(b.dims.vocab)] == tiny_model["tokens"]
    a.close(); b.close()


def test_blob_file_cache(built, tiny_model, tmp_path):
    """save_blob / load_blob: the packed-weight cache file reproduces the model."""
    a = A.Model.load_host_only(tiny_model["path"])
    p = str(tmp_path / "tiny.apxblob")
    a.save_blob(p)
    b = A.Model.load_blob(p, init_gpu=False)
    assert np.array_equal(a.export_blob(), b.export_blob())
    assert b.get_name() == a.get_name() and [b.token(i) for i in range(b.dims.vocab)] == tiny_model["tokens"]
    with open(p, "r+b") as f:
        f.write(b"XXXXXXXX")                                   # bad magic
    with pytest.raises(Exception):
        A.Model.load_blob(p, init_gpu=False)
    with pytest.raises(Exception):
        A.Model.load_blob(str(tmp_path / "missing.apxblob"), init_gpu=False)
    a.close(); b.close()


def test_blob_file_cache_f16(built, tiny_model, tmp_path):
    """The fp16 cache file (BASELINE configs[4]): MFMA-packed matrices as binary16, the rest fp32; about
    half the size; it expands to the fp32 blob with exactly the binary16-rounded matrices (so the engine's fp16 copies are the
    same bits as those derived from the original model) and everything else untouched."""
    import os
    a = A.Model.load_host_only(tiny_model["path"])
    p32, p16 = str(tmp_path / "tiny.apxblob"), str(tmp_path / "tiny.apxblob16")
    a.save_blob(p32); a.save_blob(p16, f16=True)
    assert os.path.getsize(p16) < 0.62 * os.path.getsize(p32)
    b = A.Model.load_blob(p16, init_gpu=False)
    wa, wb = split_blob(a.export_blob()), split_blob(b.export_blob())
    assert wa.shape == wb.shape
    L, ein, vp = layout_offsets(tiny_model["dims"])
    d = tiny_model["dims"]
    D, H, F, J = d["d_model"], d["hidden"], d["ffn"], d["joiner"]
    gemm = np.zeros(wa.size, bool)
    secs = [(L["w_embed"], ein * D), (L["w_encproj"], D * J), (L["w_decproj"], D * J), (L["w_out"], J * vp)]
    for l in range(d["n_layers"]):
        secs += [(L["wg%d" % l], 2 * D * 4 * H), (L["whr%d" % l], H * D), (L["wff1%d" % l], D * F), (L["wff2%d" % l], F * D)]
    for off, n in secs:
        gemm[off:off + n] = True
    assert np.array_equal(wb[gemm], wa[gemm].astype(np.float16).astype(np.float32))       # round to nearest even, like the device conversion
    assert np.array_equal(wb[~gemm], wa[~gemm])
    assert [b.token(i) for i in range(b.dims.vocab)] == tiny_model["tokens"]
    a.close(); b.close()


def test_param_count_aprilv0_dims():
    """84.18 M parameters at aprilv0 dimensions (SURVEY.md Appendix C cross-check), from shapes alone."""
    d = SM.APRILV0_DIMS
    D, H, F, J, V = d["d_model"], d["hidden"], d["ffn"], d["joiner"], d["vocab"]
    embed = 8 * 9 + 8 + 32 * 8 * 9 + 32 + 128 * 32 * 9 + 128 + 2304 * D + D
    layer = 2 * 4 * H * D + 2 * 4 * H + H * D + D * F + F + F * D + D + 1
    total = embed + 12 * layer + D * J + J + V * D + D * (D // d["dec_groups"]) * 2 + D * J + J + J * V + V
    assert embed == 1219568 and layer == 6826497 and total == 84179440


@pytest.mark.parametrize("breakage", ["wrong_gate_act", "swish_const", "no_norm"])
def test_unsupported_graphs_are_rejected(built, tmp_path, breakage, capfd):
    """The loader verifies the operators around each weight instead of trusting positions."""
    import april_asr_amd.synth_model as S
    dims = dict(S.TINY_DIMS)
    w = S.make_weights(dims)
    toks = S.make_tokens(dims["vocab"])
    orig_ds, orig_bn = S._double_swish, S._basic_norm
    try:
        if breakage == "swish_const":
            S._double_swish = lambda g, x: g.node("Mul", [x, g.node("Sigmoid", [g.node("Sub", [x, g.const(np.array(2.0, np.float32))])])])
        if breakage == "no_norm":
            S._basic_norm = lambda g, x, e, n, f: g.node("Identity", [x])
        enc = S.build_encoder(dims, w, {})
        if breakage == "wrong_gate_act":
            enc = enc.replace(b"\x22\x04Tanh", b"\x22\x04Relu", 1)      # op_type field (4) of the first Tanh node
    finally:
        S._double_swish, S._basic_norm = orig_ds, orig_bn
    blob = S.container_bytes([enc, S.build_decoder(dims, w, {}), S.build_joiner(dims, w, {})], S.params_block(dims, toks))
    p = tmp_path / "bad.april"
    p.write_bytes(blob)
    with pytest.raises(Exception):
        A.Model.load_host_only(str(p))
    assert "failed to load" in capfd.readouterr().err


def test_corrupted_models_never_crash(built, tiny_model, tmp_path):
    """Model files are untrusted input: 400 corrupted copies (byte flips, header flips, truncations, 0xff runs) must be
    either rejected or loaded -- never crash the process (product loader and oracle loader, in a subprocess)."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_loader_worker.py")
    for seed in (1, 2):
        out = subprocess.run([sys.executable, worker, tiny_model["path"], str(tmp_path / "fuzz.april"), str(seed), "200"],
                             capture_output=True, text=True, errors="replace", timeout=600)
        assert out.returncode == 0, "loader crashed (seed %d): %s" % (seed, (out.stdout + out.stderr)[-2000:])
        assert "accepted product=" in out.stdout


# ---------------------------------------------------------------- a REAL exporter's spelling of the graphs
ODD_DIMS = dict(SM.TINY_DIMS, n_layers=3, d_model=128, hidden=192, ffn=256, joiner=64, vocab=131)


@pytest.fixture(scope="module")
def torch_exported(built, tmp_path_factory):
    """The network as PyTorch modules (icefall structure, LSTM with projection as explicit operations), exported by
    torch.onnx.export exactly like the reference's extra/export-april.py:226-331 (opset 11, static shapes) and wrapped into a
    .april container (tests/torch_export.py)."""
    pytest.importorskip("torch")
    import torch_export as TE
    out = {}
    for tag, dims, style in (("tiny", dict(SM.TINY_DIMS), {}), ("odd", ODD_DIMS, {}), ("icefall", dict(SM.TINY_DIMS), dict(chunk="icefall"))):
        w = SM.make_weights(dims, seed=2023)
        toks = SM.make_tokens(dims["vocab"])
        nets, mods = TE.export_networks(w, dims, **style)
        d = tmp_path_factory.mktemp("torch_export_" + tag)
        p_t, p_s = str(d / "torch.april"), str(d / "synth.april")
        with open(p_t, "wb") as f:
            f.write(SM.container_bytes(nets, SM.params_block(dims, toks), name="torch-export"))
        SM.write_model(p_s, dims, seed=2023)
        out[tag] = dict(torch=p_t, synth=p_s, dims=dims, weights=w, modules=mods)
    return out


@pytest.mark.parametrize("tag", ["tiny", "odd", "icefall"])
def test_torch_onnx_export_loads_to_the_same_weights(torch_exported, tag):
    """Graphs written by torch.onnx.export (constant-folded transposed weights named onnx::MatMul_###, MatMul + Add for every
    Linear, Tensor.chunk as Shape -> Gather -> Add/Div/Mul -> four Slices with computed bounds, per-layer state through
    Slice, new states through Concat, BasicNorm as Pow / ReduceMean / Exp / Add / Pow / Mul) load to exactly the packed
    weights of the same network written by the repo's own graph writer.  "icefall": the LSTM with projection in the shape of
    icefall's own ONNX-exportable module -- input product in 3-D, unbind over time, recurrent product on 2-D operands (Gemm),
    chunk along dim 1, outputs stacked, states squeezed / unsqueezed."""
    t = torch_exported[tag]
    a = A.Model.load_host_only(t["torch"]); b = A.Model.load_host_only(t["synth"])
    assert (a.dims.n_layers, a.dims.d_model, a.dims.hidden, a.dims.ffn, a.dims.joiner, a.dims.vocab, a.dims.context) == \
           (b.dims.n_layers, b.dims.d_model, b.dims.hidden, b.dims.ffn, b.dims.joiner, b.dims.vocab, b.dims.context)
    assert np.array_equal(split_blob(a.export_blob()), split_blob(b.export_blob()))
    a.close(); b.close()


@pytest.mark.parametrize("tag", ["tiny", "odd", "icefall"])
def test_oracle_runs_torch_exported_graphs(torch_exported, tag):
    """The CPU oracle's ONNX interpreter executes the torch-exported graphs (incl. the Shape / Gather / Div index arithmetic
    they carry) and agrees with the PyTorch modules they were exported from."""
    import torch
    from oracle import orc_py as O
    t = torch_exported[tag]
    dims = t["dims"]
    enc, dec, joi = t["modules"]
    om = O.Model(t["torch"])
    rng = np.random.RandomState(4)
    for _ in range(2):
        x = rng.uniform(-16, 8, size=(1, dims["seg"], dims["mel"])).astype(np.float32)
        h = rng.uniform(-0.5, 0.5, size=(dims["n_layers"], 1, dims["d_model"])).astype(np.float32)
        c = rng.uniform(-1, 1, size=(dims["n_layers"], 1, dims["hidden"])).astype(np.float32)
        e0, h0, c0 = om.encoder(x, h, c)
        with torch.no_grad():
            e1, h1, c1 = enc(torch.from_numpy(x), torch.from_numpy(h), torch.from_numpy(c))
        assert np.abs(e0.ravel() - e1.numpy().ravel()).max() < 2e-5
        assert np.abs(h0.ravel() - h1.numpy().ravel()).max() < 2e-5 and np.abs(c0.ravel() - c1.numpy().ravel()).max() < 2e-5
        ctx = rng.randint(0, dims["vocab"], size=dims["context"])
        d0 = om.decoder(ctx).ravel()
        with torch.no_grad():
            d1 = dec(torch.from_numpy(ctx.astype(np.int64))[None]).numpy().ravel()
            l1 = joi(e1, torch.from_numpy(d0.reshape(1, 1, -1))).numpy().ravel()
        assert np.abs(d0 - d1).max() < 2e-5
        l0 = om.joiner(e0.reshape(1, 1, -1), d0.reshape(1, 1, -1)).ravel()
        assert np.abs(l0 - l1).max() < 5e-5
    om.close()


TORCH_EXPORT_STYLES = [
    dict(opset=11, chunk="split", state="index", linear2d=False),
    dict(opset=11, chunk="narrow", state="slice", linear2d=True),
    dict(opset=11, chunk="chunk", state="index", linear2d=True),
    dict(opset=13, chunk="chunk", state="slice", linear2d=False),
    dict(opset=13, chunk="split", state="index", linear2d=True),
    dict(opset=13, chunk="narrow", state="index", linear2d=False),
    dict(opset=17, chunk="chunk", state="index", linear2d=True),
    dict(opset=17, chunk="split", state="slice", linear2d=False),
    dict(opset=13, chunk="icefall", state="index"),
    dict(opset=17, chunk="icefall", state="slice"),
]


@pytest.mark.parametrize("style", TORCH_EXPORT_STYLES, ids=lambda v: ",".join("%s=%s" % kv for kv in sorted(v.items())))
def test_torch_onnx_export_styles(built, tiny_model, tmp_path, style):
    """The same modules written with other PyTorch idioms (torch.split / narrow instead of chunk, h[i].unsqueeze + torch.stack
    instead of h[i:i+1] + torch.cat, 2-D operands so that the exporter emits Gemm and reshapes around the gate sum) and other
    opsets (13: Split / Squeeze / Unsqueeze take their sizes and axes as inputs; 17): identical packed weights."""
    pytest.importorskip("torch")
    import torch_export as TE
    dims = dict(SM.TINY_DIMS)
    nets, _ = TE.export_networks(tiny_model["weights"], dims, **style)
    p = tmp_path / "styled.april"
    p.write_bytes(SM.container_bytes(nets, SM.params_block(dims, tiny_model["tokens"])))
    a = A.Model.load_host_only(tiny_model["path"]); b = A.Model.load_host_only(str(p))
    assert np.array_equal(split_blob(a.export_blob()), split_blob(b.export_blob()))
    a.close(); b.close()


def test_convert_cli(built, tiny_model, tmp_path):
    """python -m april_asr_amd.convert: .april -> cache file on the host (no GPU); the cache loads to the same packed weights;
    the fp16 variant is half the size of the matrices and carries its own magic; a broken input is refused with exit code 1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    out32, out16 = str(tmp_path / "m.aprilx"), str(tmp_path / "m.aprilx16")
    subprocess.check_call([sys.executable, "-m", "april_asr_amd.convert", tiny_model["path"], out32], env=env, cwd=str(tmp_path))
    subprocess.check_call([sys.executable, "-m", "april_asr_amd.convert", tiny_model["path"], out16, "--f16"], env=env, cwd=str(tmp_path))
    a = A.Model.load_host_only(tiny_model["path"]); b = A.Model.load_blob(out32, init_gpu=False)
    assert np.array_equal(a.export_blob(), b.export_blob())
    a.close(); b.close()
    assert open(out16, "rb").read(8) == b"APXBL16B" and os.path.getsize(out16) < 0.62 * os.path.getsize(out32)
    bad = tmp_path / "bad.april"; bad.write_bytes(open(tiny_model["path"], "rb").read()[:5000])
    r = subprocess.run([sys.executable, "-m", "april_asr_amd.convert", str(bad), str(tmp_path / "x")], env=env, cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 1
