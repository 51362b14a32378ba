"""Runs every hand-derived state-machine case (tests/golden/state_machine_cases.py) through the DEVICE's copy of the search decision
(csrc/kernels_misc.hip decide_kernel, via aprilx_run_decide) of the library named by APRIL_ASR_LIB -- a mutant built by
tests/mutate_device_decide.py.  Needs a GPU.  Exit status 0 = every case passed (the mutant SURVIVES), 1 = a case caught it."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import state_machine_cases as G  # noqa: E402


def main():
    import april_asr_amd as A
    from test_gpu_decide import DeviceSearch
    from test_state_machine_golden import product_rounds, symbols
    m = A.Model(sys.argv[1])
    sym = symbols([m.token(i) for i in range(m.dims.vocab)])
    for case in G.CASES:
        try:
            dev = DeviceSearch(m)
            exp = iter(case["rounds"])
            rnd = 0
            for it in product_rounds(case, sym):
                if it[0] == "flush":
                    assert next(exp) == "FLUSH"
                    dev.flush()
                    want_ctx, want_last = case["flush_state"]
                    assert dev.ctx == (sym[want_ctx[0]], sym[want_ctx[1]]) and dev.last_tok == (-1 if want_last is None else sym[want_last]), "state after flush"
                    continue
                _, idx, mx, bl, early, now, scripted, tie = it
                rnd = 0 if early == 1.0 else rnd + 1
                blank = dev.round(idx, mx, bl, early, now, rnd, tie)
                if not scripted:
                    assert blank, "filler round"
                    continue
                e = next(exp)
                assert blank == e[0], "blank flag"
                assert dev.ctx == (sym[e[1][0]], sym[e[1][1]]), "context"
                assert dev.last_tok == (-1 if e[2] is None else sym[e[2]]), "last active token"
        except AssertionError as e:
            print("KILLED by %s: %s" % (case["name"], str(e)[:200]))
            return 1
    print("SURVIVED")
    return 0


if __name__ == "__main__":
    sys.exit(main())
