"""ctypes view of libaprilasr.so (the MI355X build).

Struct layouts and enum values are the reference ABI's (include/april_api.h;
reference april_api.h:82-174, bindings/python/april_asr/_april_c_ffi.py:9-37):
AprilToken 32 bytes {0,8,12,16,24}, AprilConfig 40 bytes {0,16,24,32}.
The library is loaded from this package directory (built in-tree by
csrc/Makefile); there is no CPU fallback -- if it is missing, importing fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APRIL_ASR_LIB") or os.path.join(_HERE, "libaprilasr.so")


class AprilSpeakerID(C.Structure):
    _fields_ = [("data", C.c_uint8 * 16)]


class AprilToken(C.Structure):
    _fields_ = [("token", C.c_char_p), ("logprob", C.c_float), ("flags", C.c_int),
                ("time_ms", C.c_size_t), ("reserved", C.c_void_p)]


HANDLER = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(AprilToken))


class AprilConfig(C.Structure):
    _fields_ = [("speaker", AprilSpeakerID), ("handler", HANDLER), ("userdata", C.c_void_p), ("flags", C.c_int)]


class AprilxDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layers", "d_model", "hidden", "ffn", "joiner", "vocab", "mel", "seg",
                                         "seg_step", "context", "fft_size", "frame_shift", "sample_rate", "blank_id",
                                         "n_devices", "precision", "d_model_file")] + [("param_count", C.c_int64)]


class AprilxStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("ticks", "steps", "chunks", "rounds", "frames", "max_batch_seen")] + \
               [("kernel_ms", C.c_double * 6), ("kernel_launches", C.c_uint64 * 6), ("host_ms", C.c_double * 8),
                ("flights", C.c_uint64), ("replay_mismatch", C.c_uint64), ("kernels_per_step", C.c_uint64),
                ("lm_steps", C.c_uint64), ("lm_chunks", C.c_uint64),
                ("wave_steps", C.c_uint64), ("wave_chunks", C.c_uint64),
                ("gates_clock_ms", C.c_double), ("gates_clock_launches", C.c_uint64), ("gates_clock_rows", C.c_uint64),
                ("gates_clock_ms_by_n", C.c_double * 4), ("gates_clock_launches_by_n", C.c_uint64 * 4)]


class AprilxLoadInfo(C.Structure):
    _fields_ = [("broadcast_ms", C.c_double), ("comm_init_ms", C.c_double), ("broadcast_bytes", C.c_uint64),
                ("ranks", C.c_int32), ("used_rccl", C.c_int32)]


EXPORTED_REFERENCE_SYMBOLS = [
    "aam_api_init", "aam_create_model", "aam_get_name", "aam_get_description", "aam_get_language",
    "aam_get_sample_rate", "aam_free", "aas_create_session", "aas_feed_pcm16", "aas_flush",
    "aas_realtime_get_speedup", "aas_free",
]
EXPORTED_ENGINE_SYMBOLS = [
    "aprilx_model_dims", "aprilx_model_token", "aprilx_model_blob_size", "aprilx_model_export_blob",
    "aprilx_model_from_blob", "aprilx_model_save_blob", "aprilx_model_save_blob_f16", "aprilx_model_load_blob",
    "aprilx_broadcast_get_id", "aprilx_model_broadcast", "aprilx_model_load_info", "aprilx_feed_many", "aprilx_flush_many", "aprilx_session_drain", "aprilx_feed_many_pipelined", "aprilx_drain_many",
    "aprilx_run_encoder", "aprilx_run_decoder", "aprilx_run_joiner", "aprilx_run_fbank", "aprilx_run_decide", "aprilx_plan_gemm", "aprilx_stream_form",
    "aprilx_session_trace_logits", "aprilx_session_chunks", "aprilx_session_read_frames", "aprilx_session_context", "aprilx_model_stats", "aprilx_model_profile", "aprilx_model_feed_latency",
    "aprilx_greedy_create", "aprilx_greedy_step", "aprilx_greedy_finish", "aprilx_greedy_free", "aprilx_probe_file", "aprilx_model_load_host", "aprilx_model_fbank_tables", "aprilx_counting_handler",
]

_lib = None
_inited = False


def lib():
    """Load the shared library and declare prototypes (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libaprilasr.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C april_asr_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.aam_api_init.argtypes = [C.c_int]; L.aam_api_init.restype = None
    L.aam_create_model.argtypes = [C.c_char_p]; L.aam_create_model.restype = vp
    for f in ("aam_get_name", "aam_get_description", "aam_get_language"):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = C.c_char_p
    L.aam_get_sample_rate.argtypes = [vp]; L.aam_get_sample_rate.restype = sz
    L.aam_free.argtypes = [vp]; L.aam_free.restype = None
    L.aas_create_session.argtypes = [vp, AprilConfig]; L.aas_create_session.restype = vp
    L.aas_feed_pcm16.argtypes = [vp, vp, sz]; L.aas_feed_pcm16.restype = None
    L.aas_flush.argtypes = [vp]; L.aas_flush.restype = None
    L.aas_realtime_get_speedup.argtypes = [vp]; L.aas_realtime_get_speedup.restype = C.c_float
    L.aas_free.argtypes = [vp]; L.aas_free.restype = None
    L.aprilx_model_dims.argtypes = [vp, C.POINTER(AprilxDims)]; L.aprilx_model_dims.restype = C.c_int
    L.aprilx_model_token.argtypes = [vp, C.c_int32]; L.aprilx_model_token.restype = C.c_char_p
    L.aprilx_model_blob_size.argtypes = [vp]; L.aprilx_model_blob_size.restype = sz
    L.aprilx_model_export_blob.argtypes = [vp, vp, sz]; L.aprilx_model_export_blob.restype = C.c_int
    L.aprilx_model_from_blob.argtypes = [vp, sz, C.c_int]; L.aprilx_model_from_blob.restype = vp
    L.aprilx_model_save_blob.argtypes = [vp, C.c_char_p]; L.aprilx_model_save_blob.restype = C.c_int
    L.aprilx_model_save_blob_f16.argtypes = [vp, C.c_char_p]; L.aprilx_model_save_blob_f16.restype = C.c_int
    L.aprilx_model_load_blob.argtypes = [C.c_char_p]; L.aprilx_model_load_blob.restype = vp
    L.aprilx_broadcast_get_id.argtypes = [vp, sz]; L.aprilx_broadcast_get_id.restype = C.c_int
    L.aprilx_model_broadcast.argtypes = [vp, C.c_int, C.c_int, vp]; L.aprilx_model_broadcast.restype = vp
    L.aprilx_model_load_info.argtypes = [vp, C.POINTER(AprilxLoadInfo)]; L.aprilx_model_load_info.restype = C.c_int
    L.aprilx_feed_many.argtypes = [sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz)]; L.aprilx_feed_many.restype = None
    L.aprilx_flush_many.argtypes = [sz, C.POINTER(vp)]; L.aprilx_flush_many.restype = None
    L.aprilx_session_drain.argtypes = [vp]; L.aprilx_session_drain.restype = None
    L.aprilx_feed_many_pipelined.argtypes = [sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), C.c_int]; L.aprilx_feed_many_pipelined.restype = None
    L.aprilx_drain_many.argtypes = [sz, C.POINTER(vp)]; L.aprilx_drain_many.restype = None
    L.aprilx_run_encoder.argtypes = [vp, C.c_int] + [vp] * 6; L.aprilx_run_encoder.restype = C.c_int
    L.aprilx_run_decoder.argtypes = [vp, C.c_int, vp, vp]; L.aprilx_run_decoder.restype = C.c_int
    L.aprilx_run_joiner.argtypes = [vp, C.c_int, vp, vp, vp]; L.aprilx_run_joiner.restype = C.c_int
    L.aprilx_run_fbank.argtypes = [vp, C.c_int, vp, vp]; L.aprilx_run_fbank.restype = C.c_int
    L.aprilx_plan_gemm.argtypes = [C.c_int] * 6 + [vp]; L.aprilx_plan_gemm.restype = C.c_int
    L.aprilx_stream_form.argtypes = [C.c_int] * 6; L.aprilx_stream_form.restype = C.c_int
    L.aprilx_run_decide.argtypes = [vp, C.c_int, C.c_int, vp, C.c_float, vp, C.c_int, vp, vp]; L.aprilx_run_decide.restype = C.c_int
    L.aprilx_session_trace_logits.argtypes = [vp, vp, sz, C.POINTER(sz)]; L.aprilx_session_trace_logits.restype = None
    L.aprilx_session_chunks.argtypes = [vp]; L.aprilx_session_chunks.restype = C.c_uint64
    L.aprilx_session_read_frames.argtypes = [vp, C.c_uint64, C.c_int, C.c_void_p]; L.aprilx_session_read_frames.restype = C.c_uint64
    L.aprilx_session_context.argtypes = [vp, vp, vp]; L.aprilx_session_context.restype = None
    L.aprilx_model_stats.argtypes = [vp, C.c_int, C.POINTER(AprilxStats)]; L.aprilx_model_stats.restype = None
    L.aprilx_model_feed_latency.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]; L.aprilx_model_feed_latency.restype = C.c_int
    L.aprilx_model_profile.argtypes = [vp, C.c_int]; L.aprilx_model_profile.restype = None
    L.aprilx_greedy_create.argtypes = [vp, HANDLER, vp]; L.aprilx_greedy_create.restype = vp
    L.aprilx_greedy_step.argtypes = [vp, C.c_int32, C.c_float, C.c_float, C.c_float, sz, C.POINTER(C.c_int32)]
    L.aprilx_greedy_step.restype = C.c_int
    L.aprilx_greedy_finish.argtypes = [vp]; L.aprilx_greedy_finish.restype = None
    L.aprilx_greedy_free.argtypes = [vp]; L.aprilx_greedy_free.restype = None
    L.aprilx_model_load_host.argtypes = [C.c_char_p]; L.aprilx_model_load_host.restype = vp
    L.aprilx_model_fbank_tables.argtypes = [vp, vp, vp]; L.aprilx_model_fbank_tables.restype = C.c_int
    L.aprilx_probe_file.argtypes = [C.c_char_p, C.c_char_p, sz]; L.aprilx_probe_file.restype = C.c_int
    _lib = L
    return L


def init():
    """aam_api_init(APRIL_VERSION) once per process (reference binding does this at import)."""
    global _inited
    L = lib()
    if not _inited:
        L.aam_api_init(1)
        _inited = True
    return L
