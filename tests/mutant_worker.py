"""Runs every hand-derived state-machine case (tests/golden/state_machine_cases.py) against the oracle library named by
APRIL_ORC_SO -- a mutant of oracle/orc_session.c built by tests/mutate_state_machine.py.  Exit status 0 = every case passed (the
mutant SURVIVES), 1 = a case caught it, anything else (crash, time-out handled by the caller) also counts as caught."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import state_machine_cases as G  # noqa: E402
from test_state_machine_golden import check_oracle_case, symbols  # noqa: E402


def main():
    model_path = sys.argv[1]
    from oracle import orc_py as O
    L = O.lib()
    f = L.orc_file_open(model_path.encode())
    P = f.contents.params
    tokens = [O.lib().orc_token(P, i).decode() for i in range(P.token_count)]
    sym = symbols(tokens)
    for case in G.CASES:
        try:
            check_oracle_case(case, model_path, sym)
        except AssertionError as e:
            print("KILLED by %s: %s" % (case["name"], str(e)[:300]))
            return 1
    print("SURVIVED")
    return 0


if __name__ == "__main__":
    sys.exit(main())
