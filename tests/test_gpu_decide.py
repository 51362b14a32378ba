"""The device's copy of the search decision (csrc/kernels_misc.hip decide_kernel: the part of aas_process_logits,
reference src/april_session.c:306-429, that the NEXT network call depends on) against the hand-derived fixtures of
tests/golden/state_machine_cases.py and against the host state machine, round by round, with scripted joiner planes injected
through aprilx_run_decide ("a second hand-kept copy ... with no scripted test of its own on the device";
`last_tok >= 0` standing in for `active_token_head > 0` in the digit-dot rule is exercised here)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import state_machine_cases as G  # noqa: E402

from april_asr_amd import _ffi  # noqa: E402
from test_state_machine_golden import product_rounds, symbols  # noqa: E402
from test_state_machine import random_triples  # noqa: E402

pytestmark = pytest.mark.gpu
VALID, BLANK, CTX = 1, 2, 4


@pytest.fixture(scope="module")
def gpu_tiny(tiny_model):
    import april_asr_amd as A
    m = A.Model(tiny_model["path"])
    yield m
    m.close()


class DeviceSearch:
    """One slot's search state on the device, advanced one joiner round at a time through aprilx_run_decide."""

    def __init__(self, model):
        self.L = _ffi.lib()
        self.m = model
        self.V = model.dims.vocab
        self.blank = model.dims.blank_id
        self.state = np.array([self.blank, self.blank, -1, 0], np.int32)

    def round(self, idx, mx, bl, early, now, rnd, tie=None):
        lg = np.full(self.V, -1000.0, np.float32)
        lg[idx] = mx
        if tie is not None:           # an arg-max tie with a higher token id: the device must still report idx (:311-320)
            assert tie > idx
            lg[tie] = mx
        lg[self.blank] = bl
        nowa = np.array([now], np.int32)
        rec = np.zeros(4, np.uint32)
        before = self.state.copy()
        rc = self.L.aprilx_run_decide(self.m._handle, 1, 0, lg.ctypes.data, C.c_float(early), nowa.ctypes.data, rnd, self.state.ctypes.data, rec.ctypes.data)
        assert rc == 0
        r_idx, r_max, r_bl, flags = int(rec.view(np.int32)[0]), float(rec.view(np.float32)[1]), float(rec.view(np.float32)[2]), int(rec[3])
        assert flags & VALID
        assert r_idx == idx and r_max == np.float32(mx) and r_bl == np.float32(bl)          # the record carries the arg-max and both logits
        changed = (before[0], before[1]) != (self.state[0], self.state[1])
        assert bool(flags & CTX) == changed
        return bool(flags & BLANK)

    def flush(self):
        rc = self.L.aprilx_run_decide(self.m._handle, 1, 1, None, C.c_float(0.0), None, 0, self.state.ctypes.data, None)
        assert rc == 0

    @property
    def ctx(self):
        return (int(self.state[0]), int(self.state[1]))

    @property
    def last_tok(self):
        return int(self.state[2])


@pytest.mark.parametrize("case", G.CASES, ids=[c["name"] for c in G.CASES])
def test_device_decision_matches_hand_derived(gpu_tiny, tiny_model, case):
    sym = symbols(tiny_model["tokens"])
    dev = DeviceSearch(gpu_tiny)
    exp = iter(case["rounds"])
    rnd = 0
    for it in product_rounds(case, sym):
        if it[0] == "flush":
            assert next(exp) == "FLUSH"
            dev.flush()
            want_ctx, want_last = case["flush_state"]
            assert dev.ctx == (sym[want_ctx[0]], sym[want_ctx[1]]) and dev.last_tok == (-1 if want_last is None else sym[want_last])
            continue
        _, idx, mx, bl, early, now, scripted, tie = it
        rnd = 0 if early == 1.0 else rnd + 1
        blank = dev.round(idx, mx, bl, early, now, rnd, tie)
        if not scripted:
            assert blank
            continue
        e = next(exp)
        assert blank == e[0], (case["name"], e)
        assert dev.ctx == (sym[e[1][0]], sym[e[1][1]]), (case["name"], e, dev.ctx)
        assert dev.last_tok == (-1 if e[2] is None else sym[e[2]]), (case["name"], e, dev.last_tok)
    assert next(exp, None) is None


@pytest.mark.parametrize("seed", [0, 1])
def test_device_decision_matches_host_state_machine_on_random_rounds(gpu_tiny, tiny_model, seed):
    """1500 random joiner results per seed (silences, token bursts past 72, punctuation, digits, repeats): the device's
    blank / context decisions equal the host state machine's at every round."""
    L = _ffi.lib()
    rng = np.random.RandomState(100 + seed)
    T = tiny_model["tokens"]
    special = [T.index(t) for t in (".", ",", "?", "!", " 1", "2", " 3", "4")]
    triples = random_triples(rng, len(T), 1500, special)
    h = _ffi.HANDLER(lambda ud, typ, count, toks: None)
    g = L.aprilx_greedy_create(gpu_tiny._handle, h, None)
    dev = DeviceSearch(gpu_tiny)
    ctx = (C.c_int32 * 2)()
    pos, now = 0, 0
    n_blank = n_tok = 0
    while pos < len(triples):
        now += 40
        for r in range(3):
            if pos >= len(triples):
                break
            idx, mx, bl = triples[pos]
            pos += 1
            early = 1.0 if r == 0 else 0.0
            hb = bool(L.aprilx_greedy_step(g, idx, mx, bl, early, now, ctx))
            db = dev.round(idx, mx, bl, early, now, r)
            assert hb == db, (pos, idx, mx, bl, r)
            assert dev.ctx == (int(ctx[0]), int(ctx[1])), (pos, dev.ctx, (ctx[0], ctx[1]))
            n_blank += hb
            n_tok += not hb
            if hb:
                break
    L.aprilx_greedy_free(g)
    assert n_blank > 100 and n_tok > 100


def test_every_mutant_of_the_device_decision_is_killed(built, tiny_model):
    """24 single-edit mutants of decide_kernel (csrc/kernels_misc.hip: tie order, the comparisons and constants of the blank decision, the punctuation
    override and the digit-dot rule, the context push, the silence rule), each compiled and linked into its own library on this box and run through
    aprilx_run_decide against the hand-derived cases (tests/mutate_device_decide.py): none may pass them all -- the device's copy is the third
    transcription of src/april_session.c:306-429 beside the oracle's and the host's"""
    import mutate_device_decide as DM
    killed, survivors, eq_killed, failures = DM.run_all(model_path=tiny_model["path"])
    assert not failures, failures
    assert not survivors, "mutants of decide_kernel that pass every hand-derived case: %r" % [n for n, _ in survivors]
    assert not eq_killed, "mutants listed as equivalent that a case does catch (the argument is wrong): %r" % eq_killed
    assert len(killed) == len(DM.MUTANTS) >= 20
