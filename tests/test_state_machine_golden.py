"""The hand-derived state-machine fixtures (tests/golden/state_machine_cases.py, worked out from the reference's
src/april_session.c:199-429,547-564 line by line) run through
  (1) the oracle's restatement (oracle/orc_session.c) with scripted networks,
  (2) the product's host state machine (csrc/session.cc `Greedy`, through aprilx_greedy_*),
and -- in tests/test_gpu_decide.py -- (3) the device's copy of the decision (decide_kernel).
This is the pin of the state machine the image allows: april_session.c needs onnxruntime_c_api.h to compile."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import state_machine_cases as G  # noqa: E402

import april_asr_amd as A  # noqa: E402
from april_asr_amd import _ffi  # noqa: E402
from oracle import orc_py as O  # noqa: E402
from test_state_machine import ScriptedOracle  # noqa: E402

KIND = {"PARTIAL": 1, "FINAL": 2, "SILENCE": 4}
FLUSH_CHUNKS_PRODUCT = 2          # padded chunks the product-side drivers run inside a "flush" phase


def symbols(tokens):
    """The fixture's vocabulary symbols -> token ids of this model."""
    def first(pred, skip=()):
        return next(i for i, t in enumerate(tokens) if i not in skip and pred(t))
    w1 = first(lambda t: t.startswith(" ") and len(t) > 2 and not t[1].isdigit())
    w2 = first(lambda t: t.startswith(" ") and len(t) > 2 and not t[1].isdigit(), (w1,))
    c1 = first(lambda t: not t.startswith(" ") and t.isalpha() and len(t) > 1)
    c2 = first(lambda t: not t.startswith(" ") and t.isalpha() and len(t) > 1, (c1,))
    return {"W1": w1, "W2": w2, "C1": c1, "C2": c2, "DOT": tokens.index("."), "COMMA": tokens.index(","), "D2": tokens.index("2"), "<blk>": 0}


def resolve_events(case, sym, post_base):
    """Expected events with ids and absolute times; post_base = number of chunks before the first post-flush chunk."""
    out = []
    for kind, toks in case["events"]:
        lst = []
        for (s, lp, fl, ch) in toks:
            t = 40 * (post_base + ch[1]) if isinstance(ch, tuple) else 40 * ch
            lst.append((sym[s], float(lp), int(fl), int(t)))
        out.append((KIND[kind], lst))
    return out


class PhasedOracle(ScriptedOracle):
    """ScriptedOracle whose joiner falls back to a configurable filler round once the script is exhausted."""

    def __init__(self, model_path):
        super().__init__(model_path, [])
        self.filler = None
        parent = self

        def joi(ud, e, d, logits):
            parent.ctx_before.append((parent.cur_ctx, parent.pos < len(parent.triples)))      # (context in front of this round, scripted?)
            if parent.pos < len(parent.triples):
                idx, mx, bl = parent.triples[parent.pos][:3]
                tie = parent.triples[parent.pos][3] if len(parent.triples[parent.pos]) > 3 else None
                parent.pos += 1
            else:
                assert parent.filler is not None, "the oracle ran a joiner round the script does not have"
                idx, mx, bl = parent.filler
                tie = None
                parent.fill_used += 1
            for i in range(parent.V):
                logits[i] = -1000.0
            logits[idx] = mx
            if tie is not None:
                logits[tie] = mx
            logits[parent.P.blank_id] = bl

        def dec(ud, ctx, dout):
            parent.cur_ctx = (int(ctx[0]), int(ctx[1]))

        self.ctx_before = []
        self.cur_ctx = None
        self.fill_used = 0
        self._fns = (self._fns[0], O.DEC_FN(dec), O.JOI_FN(joi))
        self.nets = O.OrcNets(None, *self._fns)
        self.L.orc_session_free(self.s)
        self.s = self.L.orc_session_new_scripted(C.byref(self.P), C.byref(self.nets), 1, 8, 8, 8, self.V, self._h, None)
        self.fed = 0

    def chunks(self):
        return int(self.L.orc_session_chunks(self.s))

    def run_more_chunks(self, n):
        """feed zeros until n more chunks have run (a chunk needs 640 new samples; the first one 1792)"""
        target = self.chunks() + n
        guard = 0
        while self.chunks() < target:
            step = 1792 if self.fed == 0 else 640
            pcm = np.zeros(step, np.int16)
            self.L.orc_session_feed_pcm16(self.s, pcm.ctypes.data, step)
            self.fed += step
            guard += 1
            assert guard < 10000
        assert self.chunks() == target


def run_oracle_case(case, model_path, sym):
    o = PhasedOracle(model_path)
    post_base = None
    for ph in case["phases"]:
        if ph[0] == "chunks":
            rounds = [(sym[r[0]], r[1], r[2]) + ((sym[r[3]],) if len(r) > 3 else ()) for ch in ph[1] for r in ch]
            o.triples = o.triples + rounds
            o.filler = None
            o.run_more_chunks(len(ph[1]))
            assert o.pos == len(o.triples), "the chunk loop consumed %d of %d scripted rounds" % (o.pos, len(o.triples))
        elif ph[0] == "flush":
            o.filler = (sym[ph[1][0]], ph[1][1], ph[1][2])
            o.L.orc_session_flush(o.s)
            post_base = o.chunks()
        elif ph[0] == "after":
            o.filler = (sym[ph[1][0]], ph[1][1], ph[1][2])
            o.run_more_chunks(3)
    ev = list(o.events)
    # the context after every scripted round = the context in front of the next joiner round (or the last one the decoder saw)
    before = o.ctx_before + [(o.cur_ctx, False)]
    ctx_after = [before[k + 1][0] for k in range(len(o.ctx_before)) if o.ctx_before[k][1]]
    o.close()
    return ev, post_base, ctx_after


def check_oracle_case(case, model_path, sym):
    """the oracle's callbacks AND its token context after every scripted round against the hand-derived expectations
    (also what tests/mutant_worker.py runs against every mutant of oracle/orc_session.c)"""
    ev, post_base, ctx_after = run_oracle_case(case, model_path, sym)
    want = resolve_events(case, sym, post_base or 0)
    assert [e[0] for e in ev] == [e[0] for e in want], (case["name"], [e[0] for e in ev], [e[0] for e in want])
    for i, (a, b) in enumerate(zip(ev, want)):
        assert a == b, (case["name"], i, a[0], a[1][-3:], b[1][-3:])
    exp = [e for e in case["rounds"] if e != "FLUSH"]
    assert len(ctx_after) == len(exp), (case["name"], len(ctx_after), len(exp))
    for i, (got, e) in enumerate(zip(ctx_after, exp)):
        assert got == (sym[e[1][0]], sym[e[1][1]]), (case["name"], "context after scripted round", i, got, e)


def product_rounds(case, sym):
    """The rounds a product-side driver feeds, in order: ('r', idx, max, blank, early, now_ms, scripted) or ('flush',).
    A flush phase runs FLUSH_CHUNKS_PRODUCT filler chunks first (one blank round each)."""
    out, chunk = [], 0
    for ph in case["phases"]:
        if ph[0] == "chunks":
            for ch in ph[1]:
                chunk += 1
                for r, rd in enumerate(ch):
                    s, mx, bl = rd[:3]
                    tie = sym[rd[3]] if len(rd) > 3 else None
                    # (an arg-max tie goes to the lower id, :311-320: the host state machine is handed the winner, the device finds it)
                    out.append(("r", sym[s] if tie is None else min(sym[s], tie), mx, bl, 1.0 if r == 0 else 0.0, 40 * chunk, True, tie if tie is None else max(sym[s], tie)))
        elif ph[0] == "flush":
            for _ in range(FLUSH_CHUNKS_PRODUCT):
                chunk += 1
                out.append(("r", sym[ph[1][0]], ph[1][1], ph[1][2], 1.0, 40 * chunk, False, None))
            out.append(("flush", chunk))
        elif ph[0] == "after":
            for _ in range(3):
                chunk += 1
                out.append(("r", sym[ph[1][0]], ph[1][1], ph[1][2], 1.0, 40 * chunk, False, None))
    return out


def run_product_case(case, model, sym):
    L = _ffi.lib()
    ev, decisions = [], []
    tok_index = {model.token(i).encode(): i for i in range(model.dims.vocab)}

    def handler(ud, typ, count, toks):
        ev.append((int(typ), [(tok_index[toks[i].token], float(toks[i].logprob), int(toks[i].flags), int(toks[i].time_ms)) for i in range(count)]))
    h = _ffi.HANDLER(handler)
    g = L.aprilx_greedy_create(model._handle, h, None)
    ctx = (C.c_int32 * 2)()
    post_base = None
    for it in product_rounds(case, sym):
        if it[0] == "flush":
            L.aprilx_greedy_finish(g)
            post_base = it[1]
            decisions.append("FLUSH")
            continue
        _, idx, mx, bl, early, now, scripted, _tie = it
        blank = bool(L.aprilx_greedy_step(g, idx, mx, bl, early, now, ctx))
        if scripted:
            decisions.append((blank, (int(ctx[0]), int(ctx[1]))))
        else:
            assert blank, "a filler round must resolve to blank"
    L.aprilx_greedy_free(g)
    return ev, decisions, post_base


@pytest.fixture(scope="module")
def host_model(tiny_model):
    m = A.Model.load_host_only(tiny_model["path"])
    yield m
    m.close()


@pytest.mark.parametrize("case", G.CASES, ids=[c["name"] for c in G.CASES])
def test_oracle_matches_hand_derived(built, tiny_model, case):
    sym = symbols(tiny_model["tokens"])
    check_oracle_case(case, tiny_model["path"], sym)


@pytest.mark.parametrize("case", G.CASES, ids=[c["name"] for c in G.CASES])
def test_product_host_state_machine_matches_hand_derived(built, tiny_model, host_model, case):
    sym = symbols(tiny_model["tokens"])
    ev, decisions, post_base = run_product_case(case, host_model, sym)
    want = resolve_events(case, sym, post_base or 0)
    assert [e[0] for e in ev] == [e[0] for e in want]
    for i, (a, b) in enumerate(zip(ev, want)):
        assert a == b, (case["name"], i, a[0], a[1][-3:], b[1][-3:])
    # per-round decisions: blank or not, context after the round
    exp = case["rounds"]
    assert len(decisions) == len(exp)
    for i, (got, e) in enumerate(zip(decisions, exp)):
        if e == "FLUSH":
            assert got == "FLUSH"
            continue
        assert got == (e[0], (sym[e[1][0]], sym[e[1][1]])), (case["name"], i, got, e)
