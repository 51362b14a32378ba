"""Plain PyTorch fp32 statement of the network (icefall lstm_transducer_stateless2 as exported by the reference's
extra/export-april.py:183-371), built from the same weight dictionary the synthetic .april writer uses.

An independent second opinion on the floating-point path: torch.nn.LSTM(proj_size=...) supplies the LSTM-with-projection
semantics (gate order i, f, g, o), torch.nn.functional the convolutions and linear maps.  Used to check
 * the CPU oracle's ONNX interpreter (tests/test_oracle_vs_torch.py, no GPU), and
 * the HIP kernels (tests/test_gpu_parity.py).
Test infrastructure only; nothing in the product imports it.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32))


def double_swish(x):
    return x * torch.sigmoid(x - 1.0)


def basic_norm(x, eps_log):
    eps = float(np.exp(np.float32(eps_log)))
    return x * (x.pow(2).mean(dim=-1, keepdim=True) + eps).pow(-0.5)


@torch.no_grad()
def encoder(w, dims, x, h, c):
    """x [9, mel], h [L, d_model], c [L, hidden]  ->  encoder_out [joiner], next_h [L, d_model], next_c [L, hidden]"""
    L, d, H = dims["n_layers"], dims["d_model"], dims["hidden"]
    t = _t(x)[None, None]                                           # (1, 1, T, mel)
    for i, stride in enumerate((1, 2, 2)):
        t = double_swish(F.conv2d(t, _t(w["conv%d.w" % i]), _t(w["conv%d.b" % i]), stride=stride))
    b, ch, tt, ff = t.shape                                         # (1, c3, 1, f_out)
    t = t.permute(0, 2, 1, 3).reshape(1, tt, ch * ff)
    src = basic_norm(F.linear(t, _t(w["embed.w"]), _t(w["embed.b"])), w["embed.eps"]).permute(1, 0, 2)   # (T'=1, N=1, d)
    hs, cs = [], []
    for l in range(L):
        p = "l%d." % l
        lstm = torch.nn.LSTM(input_size=d, hidden_size=H, proj_size=d, num_layers=1, bias=True)
        lstm.weight_ih_l0.copy_(_t(w[p + "w_ih"])); lstm.weight_hh_l0.copy_(_t(w[p + "w_hh"]))
        lstm.bias_ih_l0.copy_(_t(w[p + "b_ih"])); lstm.bias_hh_l0.copy_(_t(w[p + "b_hh"]))
        lstm.weight_hr_l0.copy_(_t(w[p + "w_hr"]))
        y, (h1, c1) = lstm(src, (_t(h[l])[None, None], _t(c[l])[None, None]))
        hs.append(h1[0, 0]); cs.append(c1[0, 0])
        src1 = y + src
        ffn = F.linear(double_swish(F.linear(src1, _t(w[p + "ff1.w"]), _t(w[p + "ff1.b"]))), _t(w[p + "ff2.w"]), _t(w[p + "ff2.b"]))
        src = basic_norm(src1 + ffn, w[p + "eps"])
    eout = F.linear(src.permute(1, 0, 2), _t(w["enc_proj.w"]), _t(w["enc_proj.b"]))
    return eout.reshape(-1).numpy(), torch.stack(hs).numpy(), torch.stack(cs).numpy()


@torch.no_grad()
def decoder(w, dims, ctx):
    """ctx: `context` token ids -> decoder_out [joiner]"""
    e = _t(w["emb"])[torch.as_tensor(np.asarray(ctx, np.int64))][None]               # (1, ctx, d)
    e = F.conv1d(e.permute(0, 2, 1), _t(w["dec_conv.w"]), None, groups=dims["dec_groups"]).permute(0, 2, 1)
    return F.linear(torch.relu(e), _t(w["dec_proj.w"]), _t(w["dec_proj.b"])).reshape(-1).numpy()


@torch.no_grad()
def joiner(w, dims, e, d):
    return F.linear(torch.tanh(_t(e) + _t(d)), _t(w["out.w"]), _t(w["out.b"])).reshape(-1).numpy()
