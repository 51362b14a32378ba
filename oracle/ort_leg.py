"""ORACLE-side measurement / parity leg (test infrastructure, NOT product code): the reference's network backend.

The reference evaluates its three ONNX graphs with ONNXRuntime's CPU provider, one Run per graph per chunk, intra = inter = 1
thread (src/april_model.c:45-61, call sites src/april_session.c:145,160,176).  ONNXRuntime is not installed in the build
container and no real `.april` model is available offline, so this leg is CONDITIONAL: when the `onnxruntime` Python module
is importable, `OrtSession` runs the embedded graphs of a model file through ORT and drives them with the oracle's
restatement of april_session.c (orc_session.c: fbank, chunk loop, greedy search, result state machine) through its network
hooks (OrcNets).  bench.py times it next to the GPU engine; tests/test_real_model.py compares the
GPU engine with it token for token when APRIL_MODEL points at a real model.

The session logic is exercised here without ORT too (tests/test_ort_leg.py) by plugging the oracle's own graph interpreter
in behind the same `InferenceSession.run` interface.
"""
import ctypes as C

import numpy as np

from . import orc_py as O


def available():
    try:
        import onnxruntime  # noqa: F401
        return True
    except Exception:
        return False


def _ort_factory(graph_bytes):
    import onnxruntime as ort
    so = ort.SessionOptions()
    so.intra_op_num_threads = 1          # reference src/april_model.c:54-55
    so.inter_op_num_threads = 1
    return ort.InferenceSession(graph_bytes, so, providers=["CPUExecutionProvider"])


class OrtSession:
    """One recognition session whose encoder / decoder / joiner calls go to `make_session(graph_bytes).run(...)`."""

    def __init__(self, model_path, make_session=None, trace_logits=0):
        self.L = O.lib()
        self.model = O.Model(model_path)
        f = self.model.p.contents.file.contents
        blob = C.string_at(f.blob, f.blob_size)
        nets = [blob[f.net_off[i]: f.net_off[i] + f.net_size[i]] for i in range(3)]
        make_session = make_session or _ort_factory
        self.enc, self.dec, self.joi = (make_session(b) for b in nets)
        m = self.model
        L, d, H = int(m.h_dim[0]), int(m.h_dim[2]), int(m.c_dim[2])
        T, mel, J, V = int(m.x_dim[1]), int(m.x_dim[2]), int(m.eout_dim[2]), int(m.vocab)
        self.vocab = V
        self.events = []

        def view(ptr, n, dt=np.float32):
            return np.ctypeslib.as_array(ptr, shape=(n,)).view(dt) if dt != np.float32 else np.ctypeslib.as_array(ptr, shape=(n,))

        def enc_cb(ud, x, h, c, eout, h2, c2):
            r = self.enc.run(["encoder_out", "next_h", "next_c"],
                             {"x": view(x, T * mel).reshape(1, T, mel).copy(), "h": view(h, L * d).reshape(L, 1, d).copy(),
                              "c": view(c, L * H).reshape(L, 1, H).copy()})
            view(eout, J)[:] = np.asarray(r[0], np.float32).ravel()
            view(h2, L * d)[:] = np.asarray(r[1], np.float32).ravel()
            view(c2, L * H)[:] = np.asarray(r[2], np.float32).ravel()

        def dec_cb(ud, ctx, dout):
            c = np.ctypeslib.as_array(ctx, shape=(2,)).reshape(1, 2).astype(np.int64)
            view(dout, J)[:] = np.asarray(self.dec.run(["decoder_out"], {"context": c})[0], np.float32).ravel()

        def joi_cb(ud, eout, dout, logits):
            r = self.joi.run(["logits"], {"encoder_out": view(eout, J).reshape(1, 1, J).copy(), "decoder_out": view(dout, J).reshape(1, 1, J).copy()})
            view(logits, V)[:] = np.asarray(r[0], np.float32).ravel()

        self._cbs = (O.ENC_FN(enc_cb), O.DEC_FN(dec_cb), O.JOI_FN(joi_cb))
        self._nets = O.OrcNets(None, *self._cbs)

        def _h(ud, typ, count, toks):
            self.events.append((int(typ), [(int(toks[i].id), float(toks[i].logprob), int(toks[i].flags), int(toks[i].time_ms)) for i in range(count)]))
        self._handler = O.HANDLER(_h)
        self.h = self.L.orc_session_new_scripted(C.byref(f.params), C.byref(self._nets), L, L * d, L * H, J, V, self._handler, None)
        assert self.h
        self._lt = None
        if trace_logits:
            self._lt = np.zeros(trace_logits * V, np.float32); self._lt_used = C.c_size_t(0)
            self.L.orc_session_set_logit_trace(self.h, self._lt.ctypes.data, self._lt.size, C.byref(self._lt_used))

    def feed(self, pcm):
        p = np.ascontiguousarray(pcm, np.int16)
        self.L.orc_session_feed_pcm16(self.h, p.ctypes.data, p.size)

    def flush(self):
        self.L.orc_session_flush(self.h)

    def chunks(self):
        return int(self.L.orc_session_chunks(self.h))

    def logits(self):
        return self._lt[: self._lt_used.value].reshape(-1, self.vocab)

    def close(self):
        if self.h:
            self.L.orc_session_free(self.h); self.h = None
        self.model.close()
