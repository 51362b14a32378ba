"""Runs every hand-derived state-machine case (tests/golden/state_machine_cases.py) through the PRODUCT's host state machine
(csrc/session.cc `Greedy`, aprilx_greedy_*) of the library named by APRIL_ASR_LIB -- a mutant of session.cc built by
tests/mutate_product_state_machine.py.  Host-only (no GPU).  Exit status 0 = every case passed (the mutant SURVIVES), 1 = a case caught it."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import state_machine_cases as G  # noqa: E402


def main():
    import april_asr_amd as A
    from test_state_machine_golden import resolve_events, run_product_case, symbols
    model_path = sys.argv[1]
    m = A.Model.load_host_only(model_path)
    tokens = [m.token(i) for i in range(m.dims.vocab)]
    sym = symbols(tokens)
    for case in G.CASES:
        try:
            ev, decisions, post_base = run_product_case(case, m, sym)
            want = resolve_events(case, sym, post_base or 0)
            assert [e[0] for e in ev] == [e[0] for e in want], "event types"
            for i, (a, b) in enumerate(zip(ev, want)):
                assert a == b, ("event", i)
            exp = case["rounds"]
            assert len(decisions) == len(exp), "round count"
            for i, (got, e) in enumerate(zip(decisions, exp)):
                if e == "FLUSH":
                    assert got == "FLUSH"
                    continue
                assert got == (e[0], (sym[e[1][0]], sym[e[1][1]])), ("round", i)
        except AssertionError as e:
            print("KILLED by %s: %s" % (case["name"], str(e)[:200]))
            return 1
    print("SURVIVED")
    return 0


if __name__ == "__main__":
    sys.exit(main())
