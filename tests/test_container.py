"""`.april` container + PARAMS parsing: the oracle parser (oracle/orc_file.c), the product parser
(april_asr_amd/csrc/model_loader.cc via aprilx_probe_file) and -- where /root/reference exists --
the reference's own model_file.c / params.c compiled into oracle/_ref, on the same files,
including the rejection cases of model_file.c:68-127 and params.c:71-82 (G4)."""
import ctypes as C
import struct

import numpy as np
import pytest

from april_asr_amd import _ffi, synth_model as SM
from oracle import orc_py as O


def product_accepts(path):
    err = C.create_string_buffer(256)
    rc = _ffi.lib().aprilx_probe_file(path.encode(), err, 256)
    return rc == 0, err.value.decode()


def oracle_accepts(path):
    L = O.lib()
    f = L.orc_file_open(path.encode())
    ok = bool(f)
    if ok:
        L.orc_file_free(f)
    return ok


def reference_accepts(path):
    """model_read + model_read_params exactly as april_model.c:30-61 sequences them."""
    R = O.ref()
    m = R.model_read(path.encode())
    if not m:
        return False
    n = R.model_network_count(m)
    buf = C.create_string_buffer(1)
    for i in range(n):           # the reference reads the networks first, leaving the FILE position at params
        sz = R.model_network_size(m, i)
        b = C.create_string_buffer(max(sz, 1))
        R.model_network_read(m, i, b, sz)
    p = O.RefModelParameters()
    ok = bool(R.model_read_params(m, C.byref(p)))
    R.free_model(m)
    return ok


def base_parts():
    toks = SM.make_tokens(12)
    return [b"ENC", b"DECODER", b"J"], SM.params_block(dict(seg=9, mel=80), toks), toks


def write(tmp_path, name, blob):
    p = tmp_path / name
    p.write_bytes(blob)
    return str(p)


def mutate_params(**kw):
    nets, params, toks = base_parts()
    names = ["batch", "seg", "step", "mel", "rate", "shift", "length", "pow2", "mel_low", "mel_high", "snip", "count", "blank"]
    vals = list(struct.unpack("<13i", params[8:8 + 52]))
    for k, v in kw.items():
        vals[names.index(k)] = v
    return nets, params[:8] + struct.pack("<13i", *vals) + params[60:]


CASES = {
    "ok": lambda: SM.container_bytes(*base_parts()[:2]),
    "bad_magic": lambda: b"APRILMDX" + SM.container_bytes(*base_parts()[:2])[8:],
    "version_2": lambda: SM.container_bytes(*base_parts()[:2], version=2),
    "type_0": lambda: SM.container_bytes(*base_parts()[:2], model_type=0),
    "type_2": lambda: SM.container_bytes(*base_parts()[:2], model_type=2),
    "nine_networks": lambda: SM.container_bytes([b"x"] * 9, base_parts()[1]),
    "truncated": lambda: SM.container_bytes(*base_parts()[:2])[:-30],
    "params_bad_magic": lambda: SM.container_bytes(base_parts()[0], b"PARAMZ\0\0" + base_parts()[1][8:]),
    "batch_2": lambda: SM.container_bytes(*mutate_params(batch=2)),
    "seg_100": lambda: SM.container_bytes(*mutate_params(seg=100)),
    "step_gt_seg": lambda: SM.container_bytes(*mutate_params(step=10)),
    "mel_256": lambda: SM.container_bytes(*mutate_params(mel=256)),
    "rate_0": lambda: SM.container_bytes(*mutate_params(rate=0)),
    "blank_oob": lambda: SM.container_bytes(*mutate_params(blank=12)),
    "shift_gt_length": lambda: SM.container_bytes(*mutate_params(shift=30)),
    "mel_low_0": lambda: SM.container_bytes(*mutate_params(mel_low=0)),
    "mel_high_le_low": lambda: SM.container_bytes(*mutate_params(mel_high=10)),
}
EXPECT_OK = {"ok"}


@pytest.mark.parametrize("case", sorted(CASES))
def test_accept_reject(built, tmp_path, case):
    path = write(tmp_path, case + ".april", CASES[case]())
    want = case in EXPECT_OK
    ok, msg = product_accepts(path)
    assert ok == want, "product: %s (%s)" % (ok, msg)
    assert oracle_accepts(path) == want
    if O.ref_available() and case not in ("truncated",):      # the reference reads past EOF silently there
        assert reference_accepts(path) == want


def test_missing_file(built):
    assert not product_accepts("/nonexistent/model.april")[0]
    assert not oracle_accepts("/nonexistent/model.april")
    assert not O.lib().orc_model_load(b"/nonexistent/model.april")


def test_fields_match(built, tiny_model):
    """Header strings, PARAMS ints and the token table: oracle == product == (reference)."""
    import april_asr_amd as A
    L = O.lib()
    f = L.orc_file_open(tiny_model["path"].encode())
    fp = f.contents
    m = A.Model.load_host_only(tiny_model["path"])
    assert m.get_language() == fp.language.decode() == "en-us"
    assert m.get_name() == fp.name.decode()
    assert m.get_description() == fp.description.decode()
    assert m.get_sample_rate() == fp.params.sample_rate == 16000
    assert m.dims.vocab == fp.params.token_count == len(tiny_model["tokens"])
    assert m.dims.blank_id == fp.params.blank_id == 0
    for i, t in enumerate(tiny_model["tokens"]):
        assert m.token(i) == t == L.orc_token(C.byref(fp.params), i).decode()
    if O.ref_available():
        R = O.ref()
        rm = R.model_read(tiny_model["path"].encode())
        assert R.model_name(rm).decode() == m.get_name() and R.model_desc(rm).decode() == m.get_description()
        assert R.model_network_count(rm) == 3 and R.model_type(rm) == 1
        for i in range(3):
            assert R.model_network_size(rm, i) == fp.net_size[i]
            b = C.create_string_buffer(R.model_network_size(rm, i))
            R.model_network_read(rm, i, b, len(b))
        p = O.RefModelParameters()
        assert R.model_read_params(rm, C.byref(p))
        assert (p.segment_size, p.segment_step, p.mel_features, p.sample_rate, p.token_count, p.blank_id) == \
               (fp.params.segment_size, fp.params.segment_step, fp.params.mel_features, fp.params.sample_rate,
                fp.params.token_count, fp.params.blank_id)
        assert p.token_length == fp.params.token_stride
        for i, t in enumerate(tiny_model["tokens"]):
            assert R.get_token(C.byref(p), i).decode() == t
        R.free_model(rm)
    L.orc_file_free(f)
    m.close()


def test_oracle_f16_rounding_matches_ieee_binary16(built):
    """The checker mode for the fp16 path rounds exactly like an IEEE float32 -> binary16 conversion (ties to even),
    including subnormals, overflow to infinity and every half-way point."""
    from oracle import orc_py as O
    rng = np.random.RandomState(0)
    x = rng.standard_normal(100000).astype(np.float32) * rng.choice([1e-8, 1e-6, 1e-4, 1e-2, 1, 100, 1e4, 7e4], 100000).astype(np.float32)
    h = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)
    mid = ((h[:-1].astype(np.float64) + h[1:].astype(np.float64)) / 2).astype(np.float32)
    edge = np.array([0, -0.0, 65504, 65519.99, 65520, 65536, 1e9, -1e9, np.inf, -np.inf, 6.1e-5, 5.96e-8, 2.98e-8, 2.99e-8, 1e-9], np.float32)
    x = np.concatenate([x, h, -h, mid, -mid, np.nextafter(mid, np.float32(0)), np.nextafter(mid, np.float32(1e9)), edge]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(O.round_f16(x), want)
