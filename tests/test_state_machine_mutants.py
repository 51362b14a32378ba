"""Mutation test of the hand-derived state-machine fixtures (tests/mutate_state_machine.py): every single-edit mutant of the
oracle's restatement of src/april_session.c:199-476,547-564 -- comparisons flipped between strict and non-strict, the constants
3.5 / 4.0 / 8.0 / 2200 / 3000 / 71 / `i > 2` perturbed, context[0] <-> context[1], the early-emit schedule, every bookkeeping
statement dropped -- must be caught by at least one case of tests/golden/state_machine_cases.py; the mutants argued to be
equivalent must survive.  CPU only (gcc + one short Python process per mutant)."""
import mutate_state_machine as M


def test_every_mutant_of_the_state_machine_is_killed(built, tiny_model):
    killed, survivors, eq_killed, failures = M.run_all(model_path=tiny_model["path"])
    assert not failures, failures
    assert not survivors, "mutants of oracle/orc_session.c that pass every hand-derived case: %r" % [n for n, _ in survivors]
    assert not eq_killed, "mutants listed as equivalent that a case does catch (the argument is wrong): %r" % eq_killed
    assert len(killed) == len(M.MUTANTS) >= 25


def test_every_mutant_of_the_product_state_machine_is_killed(built, tiny_model):
    """the same kind of single edits in the PRODUCT's transcription (csrc/session.cc `Greedy`), each built into its own library and run
    through aprilx_greedy_* on the host (tests/mutate_product_state_machine.py): the oracle and the product are sibling transcriptions, so
    the fixtures have to catch a slip in either"""
    import mutate_product_state_machine as PM
    killed, survivors, eq_killed, failures = PM.run_all(model_path=tiny_model["path"])
    assert not failures, failures
    assert not survivors, "mutants of csrc/session.cc that pass every hand-derived case: %r" % [n for n, _ in survivors]
    assert not eq_killed, "mutants listed as equivalent that a case does catch (the argument is wrong): %r" % eq_killed
    assert len(killed) == len(PM.MUTANTS) >= 40
