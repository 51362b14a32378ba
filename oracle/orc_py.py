"""ORACLE (test infrastructure, NOT product code): ctypes bindings.

`Orc`  -> oracle/liborc.so      (our plain-C restatement, orc_*.c)
`Ref`  -> oracle/_ref/libaprilref.so (the reference's own fbank.c / pocketfft.c /
          params.c / model_file.c compiled by oracle/Makefile; may be absent)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# APRIL_ORC_SO: another build of the oracle to load instead (tests/mutate_state_machine.py runs the hand-derived state-machine
# fixtures against deliberately broken copies of orc_session.c); never set outside that test
ORC_SO = os.environ.get("APRIL_ORC_SO") or os.path.join(HERE, "liborc.so")
REF_SO = os.path.join(HERE, "_ref", "libaprilref.so")


def build(force=False):
    """Compile liborc.so (and _ref when /root/reference exists)."""
    if os.environ.get("APRIL_ORC_SO"):
        return                                      # (a mutant build: loaded as it is)
    if force or not os.path.exists(ORC_SO) or any(
        os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(ORC_SO)
        for f in os.listdir(HERE) if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-C", HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


class OrcToken(C.Structure):
    _fields_ = [("id", C.c_int32), ("logprob", C.c_float), ("flags", C.c_int32), ("time_ms", C.c_uint64)]


HANDLER = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(OrcToken))


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "batch_size", "segment_size", "segment_step", "mel_features", "sample_rate",
        "frame_shift_ms", "frame_length_ms", "round_pow2", "mel_low", "mel_high", "snip_edges",
        "token_count", "blank_id")] + [("token_stride", C.c_size_t), ("tokens", C.c_void_p)]


class OrcFile(C.Structure):
    _fields_ = [("language", C.c_char * 9), ("name", C.c_char_p), ("description", C.c_char_p),
                ("model_type", C.c_uint32), ("params_off", C.c_uint64), ("params_size", C.c_uint64),
                ("n_networks", C.c_uint64), ("net_off", C.c_uint64 * 8), ("net_size", C.c_uint64 * 8),
                ("blob", C.c_void_p), ("blob_size", C.c_size_t), ("params", OrcParams)]


class OrcModel(C.Structure):
    _fields_ = [("file", C.POINTER(OrcFile)), ("enc", C.c_void_p), ("dec", C.c_void_p), ("joi", C.c_void_p),
                ("x_dim", C.c_int64 * 3), ("h_dim", C.c_int64 * 3), ("c_dim", C.c_int64 * 3),
                ("eout_dim", C.c_int64 * 3), ("dout_dim", C.c_int64 * 3), ("ctx_dim", C.c_int64 * 2),
                ("logits_dim", C.c_int64 * 3)]


ENC_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float))
DEC_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_float))
JOI_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float))


class OrcNets(C.Structure):
    _fields_ = [("ud", C.c_void_p), ("encoder", ENC_FN), ("decoder", DEC_FN), ("joiner", JOI_FN)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORC_SO)
        L.orc_fbank_new.restype = C.c_void_p
        L.orc_fbank_new.argtypes = [C.c_int] * 9
        L.orc_fbank_free.argtypes = [C.c_void_p]
        L.orc_fbank_accept.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_fbank_flush.argtypes = [C.c_void_p]
        L.orc_fbank_pull.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_fbank_window_ptr.restype = C.POINTER(C.c_float)
        L.orc_fbank_window_ptr.argtypes = [C.c_void_p]
        L.orc_fbank_mel_ptr.restype = C.POINTER(C.c_float)
        L.orc_fbank_mel_ptr.argtypes = [C.c_void_p]
        L.orc_fbank_padded.argtypes = [C.c_void_p]
        L.orc_fbank_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_file_open.restype = C.POINTER(OrcFile)
        L.orc_file_open.argtypes = [C.c_char_p]
        L.orc_file_free.argtypes = [C.POINTER(OrcFile)]
        L.orc_token.restype = C.c_char_p
        L.orc_token.argtypes = [C.POINTER(OrcParams), C.c_size_t]
        L.orc_graph_parse.restype = C.c_void_p
        L.orc_graph_parse.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_graph_free.argtypes = [C.c_void_p]
        L.orc_graph_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                                    C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        L.orc_graph_last_error.restype = C.c_char_p
        L.orc_model_load.restype = C.POINTER(OrcModel)
        L.orc_model_load.argtypes = [C.c_char_p]
        L.orc_model_free.argtypes = [C.POINTER(OrcModel)]
        L.orc_session_new.restype = C.c_void_p
        L.orc_session_new.argtypes = [C.POINTER(OrcModel), HANDLER, C.c_void_p]
        L.orc_session_new_scripted.restype = C.c_void_p
        L.orc_session_new_scripted.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcNets), C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, HANDLER, C.c_void_p]
        L.orc_session_free.argtypes = [C.c_void_p]
        L.orc_session_feed_pcm16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_session_flush.argtypes = [C.c_void_p]
        L.orc_session_set_logit_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_session_set_chunk_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_session_chunks.restype = C.c_uint64
        L.orc_session_chunks.argtypes = [C.c_void_p]
        _lib = L
    return _lib


# ----------------------------------------------------------------------------
# fbank wrappers
# ----------------------------------------------------------------------------
APRILV0_FBANK = dict(rate=16000, shift_ms=10, len_ms=25, nbins=80, round_pow2=1, mel_lo=20, mel_hi=0,
                     seg_count=9, seg_step=4)


class OrcFbank:
    def __init__(self, **kw):
        o = dict(APRILV0_FBANK); o.update(kw)
        self.o = o
        self.L = lib()
        self.h = self.L.orc_fbank_new(o["rate"], o["shift_ms"], o["len_ms"], o["nbins"], o["round_pow2"],
                                      o["mel_lo"], o["mel_hi"], o["seg_count"], o["seg_step"])
        assert self.h
        self.chunk_shape = (o["seg_count"], o["nbins"])

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_fbank_free(self.h); self.h = None

    def accept(self, wave):
        if wave is None:
            raise ValueError
        w = np.ascontiguousarray(wave, dtype=np.float32)
        self.L.orc_fbank_accept(self.h, w.ctypes.data, w.size)

    def accept_zeros(self, n):
        self.L.orc_fbank_accept(self.h, None, n)

    def flush(self):
        return bool(self.L.orc_fbank_flush(self.h))

    def pull(self):
        out = np.empty(self.chunk_shape, np.float32)
        return out if self.L.orc_fbank_pull(self.h, out.ctypes.data) else None

    def pull_all(self):
        r = []
        while True:
            c = self.pull()
            if c is None:
                return r
            r.append(c)

    def window(self):
        n = self.L.orc_fbank_padded(self.h)
        return np.ctypeslib.as_array(self.L.orc_fbank_window_ptr(self.h), (n,)).copy()

    def mel(self):
        n = self.L.orc_fbank_padded(self.h) // 2
        return np.ctypeslib.as_array(self.L.orc_fbank_mel_ptr(self.h), (self.o["nbins"], n)).copy()

    def frame(self, samples):
        s = np.ascontiguousarray(samples, np.float32)
        out = np.empty(self.o["nbins"], np.float32)
        self.L.orc_fbank_frame(self.h, s.ctypes.data, out.ctypes.data)
        return out


class RefFBankOptions(C.Structure):
    # /root/reference/src/fbank.h:26-66
    _fields_ = [("sample_freq", C.c_int), ("frame_shift_ms", C.c_int), ("frame_length_ms", C.c_int),
                ("num_bins", C.c_int), ("round_pow2", C.c_bool), ("mel_low", C.c_int), ("mel_high", C.c_int),
                ("snip_edges", C.c_bool), ("pull_segment_count", C.c_int), ("pull_segment_step", C.c_int),
                ("use_sonic", C.c_bool), ("remove_dc_offset", C.c_bool), ("preemph_coeff", C.c_float)]


_ref = None


def ref_available():
    build()
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        build()
        R = C.CDLL(REF_SO)
        R.make_fbank.restype = C.c_void_p
        R.make_fbank.argtypes = [RefFBankOptions]
        R.fbank_accept_waveform.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        R.fbank_pull_segments.restype = C.c_bool
        R.fbank_pull_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        R.fbank_flush.restype = C.c_bool
        R.fbank_flush.argtypes = [C.c_void_p]
        R.free_fbank.argtypes = [C.c_void_p]
        R.generate_povey_window.argtypes = [C.c_void_p, C.c_int]
        R.generate_banks.argtypes = [C.c_void_p] + [C.c_int] * 6
        R.model_read.restype = C.c_void_p
        R.model_read.argtypes = [C.c_char_p]
        R.model_type.argtypes = [C.c_void_p]
        R.model_name.restype = C.c_char_p
        R.model_name.argtypes = [C.c_void_p]
        R.model_desc.restype = C.c_char_p
        R.model_desc.argtypes = [C.c_void_p]
        R.model_network_count.restype = C.c_size_t
        R.model_network_count.argtypes = [C.c_void_p]
        R.model_network_size.restype = C.c_size_t
        R.model_network_size.argtypes = [C.c_void_p, C.c_size_t]
        R.model_network_read.restype = C.c_size_t
        R.model_network_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        R.model_read_params.restype = C.c_bool
        R.model_read_params.argtypes = [C.c_void_p, C.c_void_p]
        R.free_model.argtypes = [C.c_void_p]
        R.get_token.restype = C.c_char_p
        R.get_token.argtypes = [C.c_void_p, C.c_size_t]
        _ref = R
    return _ref


class RefModelParameters(C.Structure):
    # /root/reference/src/params.h:26-46
    _fields_ = [("batch_size", C.c_int), ("segment_size", C.c_int), ("segment_step", C.c_int),
                ("mel_features", C.c_int), ("sample_rate", C.c_int), ("frame_shift_ms", C.c_int),
                ("frame_length_ms", C.c_int), ("round_pow2", C.c_bool), ("mel_low", C.c_int), ("mel_high", C.c_int),
                ("snip_edges", C.c_bool), ("blank_id", C.c_int), ("token_count", C.c_int),
                ("token_length", C.c_size_t), ("tokens", C.c_void_p)]


class RefFbank:
    """The reference's own OnlineFBank (compiled from /root/reference/src/fbank.c)."""

    def __init__(self, **kw):
        o = dict(APRILV0_FBANK); o.update(kw)
        self.R = ref()
        opts = RefFBankOptions(o["rate"], o["shift_ms"], o["len_ms"], o["nbins"], bool(o["round_pow2"]), o["mel_lo"],
                               o["mel_hi"], True, o["seg_count"], o["seg_step"], False, True, 0.97)
        self.h = self.R.make_fbank(opts)
        self.chunk_shape = (o["seg_count"], o["nbins"])

    def __del__(self):
        if getattr(self, "h", None):
            self.R.free_fbank(self.h); self.h = None

    def accept(self, wave):
        w = np.array(wave, dtype=np.float32, copy=True)
        self.R.fbank_accept_waveform(self.h, w.ctypes.data, w.size)

    def accept_zeros(self, n):
        self.R.fbank_accept_waveform(self.h, None, n)

    def flush(self):
        return bool(self.R.fbank_flush(self.h))

    def pull(self):
        out = np.empty(self.chunk_shape, np.float32)
        return out if self.R.fbank_pull_segments(self.h, out.ctypes.data, out.nbytes) else None

    def pull_all(self):
        r = []
        while True:
            c = self.pull()
            if c is None:
                return r
            r.append(c)


def set_f16_linear(on):
    """Process-wide checker mode for the product's APRIL_PRECISION=f16: MatMul/Gemm operands rounded to binary16
    (ties to even), fp32 accumulation; Conv and everything else fp32.  Models loaded before the switch keep any
    rounded weight copies they already built, so set it before the first network call."""
    lib().orc_set_f16_linear(1 if on else 0)


def round_f16(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().orc_round_f16(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    return y


def lcg_pcm16(n, seed=12345):
    """SURVEY.md Appendix E recipe: s = s*1664525 + 1013904223; v = (int16)(s >> 16)."""
    out = np.empty(n, np.int16)
    s = np.uint32(seed)
    a, c = np.uint32(1664525), np.uint32(1013904223)
    with np.errstate(over="ignore"):
        for i in range(n):
            s = s * a + c
            out[i] = np.int16(np.uint16(s >> np.uint32(16)))
    return out


def lcg_pcm16_fast(n, seed=12345):
    """Vectorised equivalent of lcg_pcm16: A[k] = a^k, S[k] = 1 + a + ... + a^(k-1) (mod 2^32)
    built by doubling, then x_k = A[k] x_0 + c S[k]."""
    a, c = 1664525, 1013904223
    M = np.uint64(0xFFFFFFFF)
    A = np.array([a], np.uint64); S = np.array([1], np.uint64)
    while A.size < n:
        am, sm = A[-1], S[-1]
        A2 = (am * A) & M
        S2 = (sm + ((am * S) & M)) & M
        A = np.concatenate([A, A2]); S = np.concatenate([S, S2])
    A, S = A[:n], S[:n]
    x = (((A * np.uint64(seed)) & M) + ((np.uint64(c) * S) & M)) & M
    return ((x >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.uint16).view(np.int16)


# ----------------------------------------------------------------------------
# model / session wrappers
# ----------------------------------------------------------------------------
class Model:
    def __init__(self, path):
        self.L = lib()
        self.p = self.L.orc_model_load(path.encode())
        if not self.p:
            raise RuntimeError("oracle: failed to load " + path)
        m = self.p.contents
        self.params = m.file.contents.params
        self.vocab = int(m.logits_dim[2])
        self.x_dim = tuple(m.x_dim); self.h_dim = tuple(m.h_dim); self.c_dim = tuple(m.c_dim)
        self.eout_dim = tuple(m.eout_dim); self.ctx_dim = tuple(m.ctx_dim)

    def token(self, i):
        return self.L.orc_token(C.byref(self.params), i).decode("utf-8", "replace")

    def _run(self, g, ins, outs):
        n_in, n_out = len(ins), len(outs)
        in_names = (C.c_char_p * n_in)(*[k.encode() for k in ins])
        in_bufs = (C.c_void_p * n_in)(*[v.ctypes.data for v in ins.values()])
        out_names = (C.c_char_p * n_out)(*[k.encode() for k in outs])
        out_bufs = (C.c_void_p * n_out)(*[v.ctypes.data for v in outs.values()])
        rc = self.L.orc_graph_run(g, n_in, in_names, in_bufs, n_out, out_names, out_bufs)
        if rc:
            raise RuntimeError(self.L.orc_graph_last_error().decode())

    def encoder(self, x, h, c):
        m = self.p.contents
        eout = np.empty(self.eout_dim, np.float32); h2 = np.empty(self.h_dim, np.float32); c2 = np.empty(self.c_dim, np.float32)
        self._run(m.enc, {"x": np.ascontiguousarray(x, np.float32), "h": np.ascontiguousarray(h, np.float32),
                          "c": np.ascontiguousarray(c, np.float32)},
                  {"encoder_out": eout, "next_h": h2, "next_c": c2})
        return eout, h2, c2

    def decoder(self, ctx):
        m = self.p.contents
        dout = np.empty(self.eout_dim, np.float32)
        self._run(m.dec, {"context": np.ascontiguousarray(ctx, np.int64).reshape(self.ctx_dim)}, {"decoder_out": dout})
        return dout

    def joiner(self, e, d):
        m = self.p.contents
        lg = np.empty((1, 1, self.vocab), np.float32)
        self._run(m.joi, {"encoder_out": np.ascontiguousarray(e, np.float32), "decoder_out": np.ascontiguousarray(d, np.float32)},
                  {"logits": lg})
        return lg

    def close(self):
        if self.p:
            self.L.orc_model_free(self.p); self.p = None


class Session:
    """Runs the oracle pipeline and records the callback transcript.

    events: list of (type, [(id, logprob, flags, time_ms), ...])
    """

    def __init__(self, model, trace_logits=0, trace_chunks=0):
        self.L = lib()
        self.model = model
        self.events = []

        def _h(ud, typ, count, toks):
            self.events.append((int(typ), [(int(toks[i].id), float(toks[i].logprob), int(toks[i].flags), int(toks[i].time_ms))
                                           for i in range(count)]))
        self._cb = HANDLER(_h)
        self.h = self.L.orc_session_new(model.p, self._cb, None)
        assert self.h
        self._lt = self._ct = None
        if trace_logits:
            self._lt = np.zeros(trace_logits * model.vocab, np.float32); self._lt_used = C.c_size_t(0)
            self.L.orc_session_set_logit_trace(self.h, self._lt.ctypes.data, self._lt.size, C.byref(self._lt_used))
        if trace_chunks:
            n = model.x_dim[1] * model.x_dim[2]
            self._ct = np.zeros(trace_chunks * n, np.float32); self._ct_used = C.c_size_t(0)
            self.L.orc_session_set_chunk_trace(self.h, self._ct.ctypes.data, self._ct.size, C.byref(self._ct_used))

    def feed(self, pcm):
        p = np.ascontiguousarray(pcm, np.int16)
        self.L.orc_session_feed_pcm16(self.h, p.ctypes.data, p.size)

    def flush(self):
        self.L.orc_session_flush(self.h)

    def chunks(self):
        return int(self.L.orc_session_chunks(self.h))

    def logits(self):
        return self._lt[: self._lt_used.value].reshape(-1, self.model.vocab)

    def chunk_trace(self):
        return self._ct[: self._ct_used.value].reshape(-1, self.model.x_dim[1], self.model.x_dim[2])

    def close(self):
        if self.h:
            self.L.orc_session_free(self.h); self.h = None
