"""april_asr_amd -- host-side mirror of the reference's Python binding on top of the
MI355X-native libaprilasr.so.

Same public surface as `april_asr` (reference bindings/python/april_asr/_april.py:
Result :11-30, Token :32-57, Model :59-97, Session :110-179): `Model(path)`,
`Session(model, callback, asynchronous=False, no_rt=False, speaker_name="")`,
`session.feed_pcm16(bytes)`, `session.flush()`, `session.get_rt_speedup()`.
Extras that only make sense on a GPU engine are on `Model`/`SessionGroup`:
batched feeding of many sessions in one call, network-level parity entry points,
weight-blob export/import for the RCCL broadcast at load time.
"""
import ctypes as C
import struct
import weakref
from enum import IntEnum
from typing import Callable, List, Sequence

import numpy as np

from . import _ffi

__all__ = ["Result", "Token", "Model", "Session", "SessionGroup"]


class Result(IntEnum):
    PARTIAL_RECOGNITION = 1   # text so far; the next call repeats it, updated
    FINAL_RECOGNITION = 2     # final; the next call starts from an empty list
    ERROR_CANT_KEEP_UP = 3    # asynchronous sessions: ingest ring overflowed, audio dropped
    SILENCE = 4               # some silence passed; empty token list


class Token:
    """One emitted token: text carries its own spacing; `logprob` is the raw joiner logit."""
    __slots__ = ("token", "logprob", "word_boundary", "sentence_end", "time")

    def __init__(self, raw):
        self.token = raw.token.decode("utf-8", "replace")
        self.logprob = float(raw.logprob)
        self.word_boundary = bool(raw.flags & 1)
        self.sentence_end = bool(raw.flags & 2)
        self.time = float(raw.time_ms) / 1000.0

    def __repr__(self):
        return "Token(%r, %.3f, wb=%d, eos=%d, t=%.2f)" % (self.token, self.logprob, self.word_boundary,
                                                             self.sentence_end, self.time)


class Model:
    def __init__(self, path: str = None, _handle=None):
        self._L = _ffi.init()
        if _handle is None:
            _handle = self._L.aam_create_model(path.encode("utf-8"))
        if not _handle:
            raise Exception("Failed to load model")
        self._handle = _handle
        d = _ffi.AprilxDims()
        self._L.aprilx_model_dims(self._handle, C.byref(d))
        self.dims = d

    # ---- reference surface
    def get_name(self) -> str:
        return self._L.aam_get_name(self._handle).decode("utf-8")

    def get_description(self) -> str:
        return self._L.aam_get_description(self._handle).decode("utf-8")

    def get_language(self) -> str:
        return self._L.aam_get_language(self._handle).decode("utf-8")

    def get_sample_rate(self) -> int:
        return int(self._L.aam_get_sample_rate(self._handle))

    def close(self):
        if getattr(self, "_handle", None):
            # sessions must not outlive their model (reference april_api.h:72-73): close the ones still open first, so a
            # session leaked by a failing caller cannot touch a freed model later (interpreter exit order is arbitrary)
            for s in list(getattr(self, "_sessions", ())):
                s.close()
            self._L.aam_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- engine-level extras
    def token(self, idx: int) -> str:
        t = self._L.aprilx_model_token(self._handle, idx)
        return t.decode("utf-8", "replace") if t is not None else ""

    def export_blob(self) -> np.ndarray:
        n = self._L.aprilx_model_blob_size(self._handle)
        buf = np.empty(n, np.uint8)
        if self._L.aprilx_model_export_blob(self._handle, buf.ctypes.data, n) != 0:
            raise RuntimeError("blob export failed")
        return buf

    @classmethod
    def load_host_only(cls, path: str):
        """Parse + extract + pack without any GPU object (loader / state-machine tests)."""
        L = _ffi.lib()
        h = L.aprilx_model_load_host(path.encode("utf-8"))
        if not h:
            raise Exception("Failed to load model")
        m = cls.__new__(cls)
        m._L = L; m._handle = h
        m.dims = _ffi.AprilxDims()
        L.aprilx_model_dims(h, C.byref(m.dims))
        return m

    def fbank_tables(self):
        n = self.dims.fft_size
        w = np.empty(n, np.float32); mel = np.empty((self.dims.mel, n // 2), np.float32)
        self._L.aprilx_model_fbank_tables(self._handle, w.ctypes.data, mel.ctypes.data)
        return w, mel

    @classmethod
    def from_blob(cls, blob, device_ptr: int = 0, size: int = 0, init_gpu: bool = True):
        """Build a model from an exported blob: a uint8 ndarray (host) or (device_ptr, size)."""
        L = _ffi.init() if (device_ptr or init_gpu) else _ffi.lib()
        if device_ptr:
            h = L.aprilx_model_from_blob(C.c_void_p(device_ptr), size, 1)
        else:
            blob = np.ascontiguousarray(blob, np.uint8)
            h = L.aprilx_model_from_blob(blob.ctypes.data, blob.size, 0)
        if not h:
            raise Exception("Failed to build model from blob")
        m = cls.__new__(cls)
        m._L = L; m._handle = h
        m.dims = _ffi.AprilxDims()
        L.aprilx_model_dims(h, C.byref(m.dims))
        return m

    # ---- one process per GPU: the model travels from rank 0 to the other ranks over RCCL inside the library
    @staticmethod
    def broadcast_id() -> bytes:
        """RCCL unique id (rank 0 creates it; hand the bytes to the other ranks by any means)."""
        L = _ffi.init()
        buf = C.create_string_buffer(128)
        n = L.aprilx_broadcast_get_id(buf, 128)
        if n <= 0:
            raise RuntimeError("aprilx_broadcast_get_id failed")
        return buf.raw[:n]

    @classmethod
    def broadcast(cls, root_model, rank: int, world: int, id_bytes: bytes):
        """Every rank calls this; rank 0 passes its model (and gets it back), the others pass None."""
        L = _ffi.init()
        idb = C.create_string_buffer(bytes(id_bytes), 128)
        h = L.aprilx_model_broadcast(root_model._handle if root_model is not None else None, rank, world, idb)
        if not h:
            raise Exception("model broadcast failed")
        if root_model is not None:
            return root_model
        m = cls.__new__(cls)
        m._L = L; m._handle = h
        m.dims = _ffi.AprilxDims()
        L.aprilx_model_dims(h, C.byref(m.dims))
        return m

    def load_info(self):
        info = _ffi.AprilxLoadInfo()
        self._L.aprilx_model_load_info(self._handle, C.byref(info))
        return info

    def save_blob(self, path: str, f16: bool = False):
        """Cache of the parsed + packed weights next to the model (aprilx_model_save_blob); f16: the half-size variant for
        fp16-operand mode (aprilx_model_save_blob_f16)."""
        fn = self._L.aprilx_model_save_blob_f16 if f16 else self._L.aprilx_model_save_blob
        if fn(self._handle, path.encode("utf-8")) != 0:
            raise RuntimeError("blob save failed")

    @classmethod
    def load_blob(cls, path: str, init_gpu: bool = True):
        L = _ffi.init() if init_gpu else _ffi.lib()
        h = L.aprilx_model_load_blob(path.encode("utf-8"))
        if not h:
            raise Exception("Failed to load model blob")
        m = cls.__new__(cls)
        m._L = L; m._handle = h
        m.dims = _ffi.AprilxDims()
        L.aprilx_model_dims(h, C.byref(m.dims))
        return m

    def run_encoder(self, x, h, c):
        d = self.dims
        x = np.ascontiguousarray(x, np.float32); n = x.shape[0]
        h = np.ascontiguousarray(h, np.float32).reshape(n, d.n_layers, d.d_model)
        c = np.ascontiguousarray(c, np.float32).reshape(n, d.n_layers, d.hidden)
        eout = np.empty((n, d.joiner), np.float32); h2 = np.empty_like(h); c2 = np.empty_like(c)
        rc = self._L.aprilx_run_encoder(self._handle, n, x.ctypes.data, h.ctypes.data, c.ctypes.data,
                                        eout.ctypes.data, h2.ctypes.data, c2.ctypes.data)
        assert rc == 0
        return eout, h2, c2

    def run_decoder(self, ctx):
        ctx = np.ascontiguousarray(ctx, np.int64).reshape(-1, self.dims.context)
        out = np.empty((ctx.shape[0], self.dims.joiner), np.float32)
        assert self._L.aprilx_run_decoder(self._handle, ctx.shape[0], ctx.ctypes.data, out.ctypes.data) == 0
        return out

    def run_joiner(self, eout, dout):
        e = np.ascontiguousarray(eout, np.float32).reshape(-1, self.dims.joiner)
        dd = np.ascontiguousarray(dout, np.float32).reshape(-1, self.dims.joiner)
        out = np.empty((e.shape[0], self.dims.vocab), np.float32)
        assert self._L.aprilx_run_joiner(self._handle, e.shape[0], e.ctypes.data, dd.ctypes.data, out.ctypes.data) == 0
        return out

    def run_fbank(self, pcm_frames):
        p = np.ascontiguousarray(pcm_frames, np.int16).reshape(-1, self.dims.fft_size)
        out = np.empty((p.shape[0], self.dims.mel), np.float32)
        assert self._L.aprilx_run_fbank(self._handle, p.shape[0], p.ctypes.data, out.ctypes.data) == 0
        return out

    def stats(self, device_index: int = 0):
        s = _ffi.AprilxStats()
        self._L.aprilx_model_stats(self._handle, device_index, C.byref(s))
        return s

    def feed_latencies(self, device_index: int = 0, reset: bool = False) -> np.ndarray:
        """hand-over -> delivery latency (ms) of the last completed ticks of one GPU's stepping thread, oldest first"""
        n = int(self._L.aprilx_model_feed_latency(self._handle, device_index, None, 0, 0))
        out = np.zeros(n, np.float64)
        if n:
            n = int(self._L.aprilx_model_feed_latency(self._handle, device_index, out.ctypes.data, n, 1 if reset else 0))
        elif reset:
            self._L.aprilx_model_feed_latency(self._handle, device_index, None, 0, 1)
        return out[:n]

    def profile(self, enable):
        """True / 1: launches one by one with hipEvents around them (per-class times in stats().kernel_ms); 2: the gates clock (feeds run as
        always, the gates kernels time themselves; stats().gates_clock_* after the next profile(0)); False / 0: off"""
        self._L.aprilx_model_profile(self._handle, int(enable))


def _dispatch(userdata, result_type, count, tokens):
    sess = C.cast(userdata, C.py_object).value
    sess._on_result(result_type, count, tokens)


_HANDLER = _ffi.HANDLER(_dispatch)


class Session:
    def __init__(self, model: Model, callback: Callable[[Result, List[Token]], None], asynchronous: bool = False,
                 no_rt: bool = False, speaker_name: str = "", raw_events: bool = False, counters=None):
        """`counters`: a uint64 ndarray of 6 entries; when given, results are only counted by a C handler inside the
        library (calls, partial, final, cant_keep_up, silence, tokens) and `callback` is never invoked."""
        self._L = model._L
        self.model = model
        self.callback = callback
        self._raw = raw_events
        cfg = _ffi.AprilConfig()
        cfg.flags = (2 if no_rt else 1) if asynchronous else 0
        if speaker_name:
            cfg.speaker = _ffi.AprilSpeakerID.from_buffer_copy(struct.pack("@q", hash(speaker_name)) * 2)
        if counters is not None:
            self._counters = counters
            cfg.handler = C.cast(self._L.aprilx_counting_handler, _ffi.HANDLER)
            cfg.userdata = counters.ctypes.data
        else:
            cfg.handler = _HANDLER
            cfg.userdata = id(self)
        self._handle = self._L.aas_create_session(model._handle, cfg)
        if not self._handle:
            raise Exception("Failed to create session")
        if not hasattr(model, "_sessions"):
            model._sessions = weakref.WeakSet()
        model._sessions.add(self)

    def _on_result(self, result_type, count, tokens):
        if self._raw:
            # (type, [(token text, logprob, flags, time_ms)]) -- exact values, for parity tests
            self.callback(int(result_type), [(tokens[i].token, float(tokens[i].logprob), int(tokens[i].flags),
                                              int(tokens[i].time_ms)) for i in range(count)])
        else:
            self.callback(Result(result_type), [Token(tokens[i]) for i in range(count)])

    def get_rt_speedup(self) -> float:
        return float(self._L.aas_realtime_get_speedup(self._handle))

    def feed_pcm16(self, data) -> None:
        """`data`: bytes of native-endian int16 mono samples (as the reference), or an int16 ndarray."""
        if isinstance(data, (bytes, bytearray, memoryview)):
            buf = (C.c_char * len(data)).from_buffer_copy(bytes(data))
            self._L.aas_feed_pcm16(self._handle, C.addressof(buf), len(data) // 2)
        else:
            a = np.ascontiguousarray(data, np.int16)
            self._L.aas_feed_pcm16(self._handle, a.ctypes.data, a.size)

    def flush(self) -> None:
        self._L.aas_flush(self._handle)

    def drain(self) -> None:
        """Asynchronous sessions: wait until everything queued so far was processed."""
        self._L.aprilx_session_drain(self._handle)

    def chunks(self) -> int:
        return int(self._L.aprilx_session_chunks(self._handle))

    def frames(self) -> np.ndarray:
        """every log-mel row the session's feature ring has received so far (real frames and flush padding), [rows][mel];
        parity tests of the online fbank"""
        total = int(self._L.aprilx_session_read_frames(self._handle, 0, 0, None))
        out = np.zeros((total, self.model.dims.mel), np.float32)
        if total:
            got = int(self._L.aprilx_session_read_frames(self._handle, 0, total, out.ctypes.data))
            if got == 2 ** 64 - 1:      # UINT64_MAX: the first rows have left the ring (more than ring_frames rows written)
                raise RuntimeError("frames(): %d rows written, the feature ring no longer holds the first ones" % total)
            if got != total:            # (a count: rows of the range were not written yet -- cannot happen for [0, total) on an idle session)
                raise RuntimeError("frames(): asked for %d rows, the session reports %d" % (total, got))
        return out

    def contexts(self):
        """(host context [2], device search state [ctx0, ctx1, last token, last emission ms]) -- derived independently, must agree"""
        h = np.zeros(2, np.int32); d = np.zeros(4, np.int32)
        self._L.aprilx_session_context(self._handle, h.ctypes.data, d.ctypes.data)
        return h, d

    def trace_logits(self, max_rows: int):
        self._trace = np.zeros((max_rows, self.model.dims.vocab), np.float32)
        self._trace_used = C.c_size_t(0)
        self._L.aprilx_session_trace_logits(self._handle, self._trace.ctypes.data, self._trace.size, C.byref(self._trace_used))

    def traced_logits(self):
        return self._trace[: self._trace_used.value // self.model.dims.vocab]

    def close(self):
        if getattr(self, "_handle", None):
            self._L.aas_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SessionGroup:
    """Feeds many sessions in ONE library call so they advance in the same GPU steps."""

    def __init__(self, sessions: Sequence[Session]):
        self.sessions = list(sessions)
        self._L = self.sessions[0]._L
        n = len(self.sessions)
        self._handles = (C.c_void_p * n)(*[s._handle for s in self.sessions])
        self._ptrs = (C.c_void_p * n)()
        self._counts = (C.c_size_t * n)()

    def feed(self, pcm_list: Sequence[np.ndarray]):
        keep = []
        for i, p in enumerate(pcm_list):
            a = np.ascontiguousarray(p, np.int16)
            keep.append(a)
            self._ptrs[i] = a.ctypes.data
            self._counts[i] = a.size
        self._L.aprilx_feed_many(len(keep), self._handles, self._ptrs, self._counts)

    def plan(self, pcms: Sequence[np.ndarray], step_samples: int):
        """Pre-build the pointer/count arrays for feeding `step_samples` per session per call, so the
        per-step host cost is one library call (what a C/C++ host would do)."""
        n = len(self.sessions)
        self._plan_keep = [np.ascontiguousarray(p, np.int16) for p in pcms]
        steps = min(p.size for p in self._plan_keep) // step_samples
        self._plan = []
        for s in range(steps):
            ptrs = (C.c_void_p * n)(*[p.ctypes.data + 2 * s * step_samples for p in self._plan_keep])
            cnts = (C.c_size_t * n)(*([step_samples] * n))
            self._plan.append((ptrs, cnts))
        return steps

    def feed_planned(self, step: int):
        ptrs, cnts = self._plan[step]
        self._L.aprilx_feed_many(len(self.sessions), self._handles, ptrs, cnts)

    def feed_planned_pipelined(self, step: int, depth: int = 2):
        """Queue feed `step` of the plan and return once at most depth - 1 feeds per session are still open (depth 2: the
        library launches this feed behind the previous one and overlaps its host work with the GPU).  Call drain() at the end."""
        ptrs, cnts = self._plan[step]
        self._L.aprilx_feed_many_pipelined(len(self.sessions), self._handles, ptrs, cnts, depth)

    def feed_pipelined(self, pcm_list: Sequence[np.ndarray], depth: int = 2):
        keep = [np.ascontiguousarray(p, np.int16) for p in pcm_list]
        for i, a in enumerate(keep):
            self._ptrs[i] = a.ctypes.data
            self._counts[i] = a.size
        self._L.aprilx_feed_many_pipelined(len(keep), self._handles, self._ptrs, self._counts, depth)      # (samples are copied inside)

    def drain(self):
        self._L.aprilx_drain_many(len(self.sessions), self._handles)

    def flush(self):
        self._L.aprilx_flush_many(len(self.sessions), self._handles)
