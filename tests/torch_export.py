"""The network as PyTorch modules in the icefall lstm_transducer_stateless2 structure, exported with torch.onnx.export the way
the reference's extra/export-april.py:226-331 does it (opset 11, static shapes N=1, T=9, three graphs), and wrapped into a
.april container.  This is the loader's check against a REAL exporter's spelling of the graphs (constant folding, Gemm vs
MatMul + Add, Slice/Split/Gather/Unsqueeze/Concat around the states, initializer names ...), not against the repo's own
graph writer (april_asr_amd/synth_model.py).

The LSTM with projection is written out as explicit tensor operations, which is what the exported models contain (ONNX has
no projected LSTM; icefall's scaling_converter(is_onnx=True) does the same).

torch's TorchScript exporter imports the `onnx` package only to append onnxscript functions to the finished protobuf (none
are used here); the package is not installed in this image, so that one post-processing step is bypassed.
Test infrastructure only; nothing in the product imports it.
"""
import io
import warnings

import numpy as np
import torch
import torch.nn as nn


def _p(a):
    return nn.Parameter(torch.from_numpy(np.ascontiguousarray(a, np.float32)), requires_grad=False)


class DoubleSwish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x - 1.0)


class BasicNorm(nn.Module):
    def __init__(self, eps_log):
        super().__init__()
        self.eps = _p(np.float32(eps_log))

    def forward(self, x):
        scales = (torch.mean(x ** 2, dim=-1, keepdim=True) + self.eps.exp()) ** -0.5
        return x * scales


class Conv2dSubsampling(nn.Module):
    def __init__(self, w, dims):
        super().__init__()
        c1, c2, c3 = dims["conv_ch"]
        self.conv = nn.Sequential(
            nn.Conv2d(1, c1, 3, padding=0), DoubleSwish(),
            nn.Conv2d(c1, c2, 3, stride=2), DoubleSwish(),
            nn.Conv2d(c2, c3, 3, stride=2), DoubleSwish())
        for i, k in enumerate((0, 2, 4)):
            self.conv[k].weight = _p(w["conv%d.w" % i]); self.conv[k].bias = _p(w["conv%d.b" % i])
        self.out = nn.Linear(w["embed.w"].shape[1], dims["d_model"])
        self.out.weight = _p(w["embed.w"]); self.out.bias = _p(w["embed.b"])
        self.out_norm = BasicNorm(w["embed.eps"])

    def forward(self, x):
        x = x.unsqueeze(1)                               # (N, T, idim) -> (N, 1, T, idim)
        x = self.conv(x)
        b, c, t, f = x.size()
        x = self.out(x.transpose(1, 2).contiguous().view(b, t, c * f))
        return self.out_norm(x)


STYLE = dict(chunk="chunk", state="slice", linear2d=False)      # how the modules below spell what an exporter leaves free


class LSTMP(nn.Module):
    """LSTM with projection as explicit operations (torch.nn.LSTM gate order i, f, g, o)."""

    def __init__(self, w, p):
        super().__init__()
        self.hidden = w[p + "w_hh"].shape[0] // 4
        self.weight_ih = _p(w[p + "w_ih"]); self.weight_hh = _p(w[p + "w_hh"])
        self.bias_ih = _p(w[p + "b_ih"]); self.bias_hh = _p(w[p + "b_hh"])
        self.weight_hr = _p(w[p + "w_hr"])

    def forward(self, x, state):
        h, c = state                                     # x (T=1, N, d), h (1, N, d), c (1, N, H)
        if STYLE["chunk"] == "icefall":
            # the shape of icefall's own ONNX-exportable LSTM with projection (icefall/lstmp.py, as far as it is remembered
            # here): input product over all time steps in 3-D, then per time step (unbind) the recurrent product on 2-D
            # operands, chunk along dim 1, new states unsqueezed back, outputs stacked
            h0, c0 = h.squeeze(0), c.squeeze(0)
            wx = torch.nn.functional.linear(x, self.weight_ih, self.bias_ih)
            ys = []
            for xt in wx.unbind(dim=0):
                gates = torch.nn.functional.linear(h0, self.weight_hh, self.bias_hh) + xt
                i, f, g, o = gates.chunk(4, dim=1)
                c0 = f.sigmoid() * c0 + i.sigmoid() * g.tanh()
                h0 = torch.nn.functional.linear(o.sigmoid() * c0.tanh(), self.weight_hr)
                ys.append(h0)
            return torch.stack(ys, dim=0), (h0.unsqueeze(0), c0.unsqueeze(0))
        if STYLE["linear2d"]:                            # 2-D operands: the exporter writes Gemm
            gates = (torch.nn.functional.linear(x.reshape(-1, x.shape[-1]), self.weight_ih, self.bias_ih) +
                     torch.nn.functional.linear(h.reshape(-1, h.shape[-1]), self.weight_hh, self.bias_hh)).reshape(1, 1, -1)
        else:
            gates = torch.nn.functional.linear(x, self.weight_ih, self.bias_ih) + torch.nn.functional.linear(h, self.weight_hh, self.bias_hh)
        if STYLE["chunk"] == "chunk":
            i, f, g, o = gates.chunk(4, dim=-1)
        elif STYLE["chunk"] == "split":
            i, f, g, o = torch.split(gates, self.hidden, dim=-1)
        else:
            H = self.hidden
            i, f, g, o = gates[..., 0:H], gates[..., H:2 * H], gates[..., 2 * H:3 * H], gates[..., 3 * H:4 * H]
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h2 = torch.nn.functional.linear(torch.sigmoid(o) * torch.tanh(c2), self.weight_hr)
        return h2, (h2, c2)


class RNNEncoderLayer(nn.Module):
    def __init__(self, w, dims, l):
        super().__init__()
        p = "l%d." % l
        self.lstm = LSTMP(w, p)
        self.feed_forward = nn.Sequential(nn.Linear(dims["d_model"], dims["ffn"]), DoubleSwish(), nn.Linear(dims["ffn"], dims["d_model"]))
        self.feed_forward[0].weight = _p(w[p + "ff1.w"]); self.feed_forward[0].bias = _p(w[p + "ff1.b"])
        self.feed_forward[2].weight = _p(w[p + "ff2.w"]); self.feed_forward[2].bias = _p(w[p + "ff2.b"])
        self.norm_final = BasicNorm(w[p + "eps"])

    def forward(self, src, states):
        src_lstm, new_states = self.lstm(src, states)
        src = src + src_lstm
        src = src + self.feed_forward(src)
        return self.norm_final(src), new_states


class MergedEncoder(nn.Module):
    """encoder(x, (h, c)) then joiner.encoder_proj (export-april.py:183-203)."""

    def __init__(self, w, dims):
        super().__init__()
        self.encoder_embed = Conv2dSubsampling(w, dims)
        self.layers = nn.ModuleList([RNNEncoderLayer(w, dims, l) for l in range(dims["n_layers"])])
        self.encoder_proj = nn.Linear(dims["d_model"], dims["joiner"])
        self.encoder_proj.weight = _p(w["enc_proj.w"]); self.encoder_proj.bias = _p(w["enc_proj.b"])

    def forward(self, x, h, c):
        x = self.encoder_embed(x)
        x = x.permute(1, 0, 2)                           # (N, T, C) -> (T, N, C)
        new_h, new_c = [], []
        for i, layer in enumerate(self.layers):
            if STYLE["state"] == "slice":
                x, (h2, c2) = layer(x, (h[i:i + 1, :, :], c[i:i + 1, :, :]))
                new_h.append(h2); new_c.append(c2)
            else:                                        # h[i] + unsqueeze in, squeeze + stack out
                x, (h2, c2) = layer(x, (h[i].unsqueeze(0), c[i].unsqueeze(0)))
                new_h.append(h2.squeeze(0)); new_c.append(c2.squeeze(0))
        x = x.permute(1, 0, 2)
        if STYLE["state"] == "slice":
            return self.encoder_proj(x), torch.cat(new_h, dim=0), torch.cat(new_c, dim=0)
        return self.encoder_proj(x), torch.stack(new_h, dim=0), torch.stack(new_c, dim=0)


class MergedDecoder(nn.Module):
    """stateless decoder (embedding, grouped conv over the context, ReLU) then joiner.decoder_proj (export-april.py:206-223)."""

    def __init__(self, w, dims):
        super().__init__()
        V, d = w["emb"].shape
        self.embedding = nn.Embedding(V, d)
        self.embedding.weight = _p(w["emb"])
        self.conv = nn.Conv1d(d, d, kernel_size=dims["context"], padding=0, groups=dims["dec_groups"], bias=False)
        self.conv.weight = _p(w["dec_conv.w"])
        self.decoder_proj = nn.Linear(d, dims["joiner"])
        self.decoder_proj.weight = _p(w["dec_proj.w"]); self.decoder_proj.bias = _p(w["dec_proj.b"])

    def forward(self, y):
        e = self.embedding(y)
        e = e.permute(0, 2, 1)
        e = self.conv(e)
        e = e.permute(0, 2, 1)
        return self.decoder_proj(torch.relu(e))


class Joiner(nn.Module):
    def __init__(self, w, dims):
        super().__init__()
        self.output_linear = nn.Linear(dims["joiner"], dims["vocab"])
        self.output_linear.weight = _p(w["out.w"]); self.output_linear.bias = _p(w["out.b"])

    def forward(self, encoder_out, decoder_out):
        return self.output_linear(torch.tanh(encoder_out + decoder_out))


def _export(module, args, input_names, output_names, opset):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils as U
    keep = U._add_onnxscript_fn
    U._add_onnxscript_fn = lambda proto, custom_opsets: proto          # needs the (absent) onnx package; nothing to add here
    buf = io.BytesIO()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(module.eval(), args, buf, verbose=False, opset_version=opset, input_names=input_names,
                              output_names=output_names, dynamo=False)
    finally:
        U._add_onnxscript_fn = keep
    return buf.getvalue()


def export_networks(w, dims, opset=11, **style):
    """-> [encoder, decoder, joiner] ONNX bytes, exported like export-april.py:270-331."""
    STYLE.update(dict(chunk="chunk", state="slice", linear2d=False))
    STYLE.update(style)
    enc, dec, joi = MergedEncoder(w, dims), MergedDecoder(w, dims), Joiner(w, dims)
    x = torch.zeros(1, dims["seg"], dims["mel"])
    h = torch.rand(dims["n_layers"], 1, dims["d_model"])
    c = torch.rand(dims["n_layers"], 1, dims["hidden"])
    context = torch.zeros(1, dims["context"], dtype=torch.int64)
    e_b = _export(enc, (x, h, c), ["x", "h", "c"], ["encoder_out", "next_h", "next_c"], opset)
    d_b = _export(dec, (context,), ["context"], ["decoder_out"], opset)
    with torch.no_grad():
        eo, _, _ = enc(x, h, c)
        do = dec(context)
    j_b = _export(joi, (eo, do), ["encoder_out", "decoder_out"], ["logits"], opset)
    return [e_b, d_b, j_b], (enc, dec, joi)
