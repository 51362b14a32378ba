"""Mutation test of the state-machine fixtures.

The oracle's result state machine (oracle/orc_session.c: decide, finalize_before_word, emit_partial, clear_context ...) and the
product's (csrc/session.cc `Greedy`) are sibling transcriptions of the reference's src/april_session.c:199-476,547-564, and
april_session.c itself cannot be compiled in this image.  What pins both to the reference is the hand-derived case list
tests/golden/state_machine_cases.py.  This script shows that the list would CATCH a wrong transcription: it builds single-edit
mutants of orc_session.c -- every comparison flipped between strict and non-strict, every constant perturbed (3.5, 4.0, 8.0, 2200,
3000, 71, `i > 2`), context[0] <-> context[1], the early-emit schedule, every bookkeeping statement dropped -- compiles each into
its own liborc and runs the cases against it (tests/mutant_worker.py, a fresh process per mutant).  A mutant that passes every
case SURVIVES; the run fails unless there are none.

Mutants that cannot change any observable behaviour (equivalent mutants) are listed separately with the argument why, and are
run as well: they must SURVIVE (a kill there would mean the argument is wrong).

usage: python tests/mutate_state_machine.py [-v]        (tests/test_state_machine_mutants.py runs it inside the CPU suite)
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
CFLAGS = ["-O1", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-w"]

# (name, text in oracle/orc_session.c -- must occur exactly once --, replacement).  Reference lines in the names' comments.
MUTANTS = [
    # ---- decide(): arg-max, equality with the previous token, early emit                                    (april_session.c:311-330)
    ("argmax_takes_last_of_equal", "if (lg[i] > best_v) { best = (int)i; best_v = lg[i]; }", "if (lg[i] >= best_v) { best = (int)i; best_v = lg[i]; }"),
    ("cleared_tests_context0", "const int cleared = s->ctx[1] == (int64_t)P->blank_id;", "const int cleared = s->ctx[0] == (int64_t)P->blank_id;"),
    ("same_tests_context0", "const int same = s->ctx[1] == (int64_t)best;", "const int same = s->ctx[0] == (int64_t)best;"),
    ("same_keeps_early_emit", "if (same) early_emit = 0.0f;", "if (0) early_emit = 0.0f;"),
    ("blank_test_not_strict", "int is_blank = (blank_v - early_emit) > best_v;", "int is_blank = (blank_v - early_emit) >= best_v;"),
    ("early_emit_added", "int is_blank = (blank_v - early_emit) > best_v;", "int is_blank = (blank_v + early_emit) > best_v;"),
    # ---- token classes and the punctuation override                                                          (:336-358)
    ("no_word_boundary_flag", "if (txt[0] == ' ') tok.flags |= 1;", "if (0) tok.flags |= 1;"),
    ("dot_is_no_sentence_end", "int eos = single && (txt[0] == '.' || txt[0] == '!' || txt[0] == '?');", "int eos = single && (txt[0] == ';' || txt[0] == '!' || txt[0] == '?');"),
    ("comma_is_no_punctuation", "int punct = eos || (single && txt[0] == ',');", "int punct = eos;"),
    ("digit_rule_needs_two_tokens", "if (punct && s->head > 0) {", "if (punct && s->head > 1) {"),
    ("digit_rule_for_every_punctuation", "if (last[0] >= '0' && last[0] <= '9' && txt[0] == '.') { eos = 0; punct = 0; }", "if (last[0] >= '0' && last[0] <= '9') { eos = 0; punct = 0; }"),
    ("digit_rule_keeps_punct", "if (last[0] >= '0' && last[0] <= '9' && txt[0] == '.') { eos = 0; punct = 0; }", "if (last[0] >= '0' && last[0] <= '9' && txt[0] == '.') { eos = 0; }"),
    ("digit_rule_keeps_eos", "if (last[0] >= '0' && last[0] <= '9' && txt[0] == '.') { eos = 0; punct = 0; }", "if (last[0] >= '0' && last[0] <= '9' && txt[0] == '.') { punct = 0; }"),
    ("digit_rule_excludes_nine", "last[0] <= '9'", "last[0] < '2'"),
    ("no_sentence_end_flag", "if (eos) tok.flags |= 2;", "if (0) tok.flags |= 2;"),
    ("override_margin_2_5", "best_v > (blank_v - 3.5f)) is_blank = 0;", "best_v > (blank_v - 2.5f)) is_blank = 0;"),
    ("override_margin_4_5", "best_v > (blank_v - 3.5f)) is_blank = 0;", "best_v > (blank_v - 4.5f)) is_blank = 0;"),
    ("override_not_strict", "best_v > (blank_v - 3.5f)) is_blank = 0;", "best_v >= (blank_v - 3.5f)) is_blank = 0;"),
    ("override_on_cleared_context", "if (!cleared && punct && !same && best_v", "if (punct && !same && best_v"),
    ("override_on_repeated_token", "if (!cleared && punct && !same && best_v", "if (!cleared && punct && best_v"),
    ("override_for_every_token", "if (!cleared && punct && !same && best_v", "if (!cleared && !same && best_v"),
    # ---- non-blank branch                                                                                     (:361-400)
    ("emission_time_not_recorded", "s->last_emit_ms = s->now_ms;", ";"),
    ("context_not_pushed", "push_context(s, (int64_t)best);\n        int fin", ";\n        int fin"),
    ("overflow_at_72", "int fin = s->head >= (ORC_MAX_ACTIVE - 1);", "int fin = s->head >= ORC_MAX_ACTIVE;"),
    ("overflow_at_70", "int fin = s->head >= (ORC_MAX_ACTIVE - 1);", "int fin = s->head >= (ORC_MAX_ACTIVE - 2);"),
    ("sentence_check_for_every_token", "if (s->head > 0 && (tok.flags & 1)) {", "if (s->head > 0) {"),
    ("no_retroactive_sentence_end", "if (prev_eos && !(prev->flags & 2)) prev->flags |= 2;", ";"),
    ("sentence_end_does_not_finalize", "if (prev_eos) fin = 1;", ";"),
    ("finalize_everything_instead_of_words", "if (fin) finalize_before_word(s, &tok);", "if (fin) finalize_all(s);"),
    ("token_emitted_unforced", "emit_partial(s, &tok, 1);", "emit_partial(s, &tok, 0);"),
    ("silence_flag_not_reset", "s->emitted_silence = 0;", ";"),
    # ---- blank branch                                                                                         (:401-426)
    ("decay_over_2000", "float decayed = best_v - (float)gap / 3000.0f;", "float decayed = best_v - (float)gap / 2000.0f;"),
    ("decay_over_4000", "float decayed = best_v - (float)gap / 3000.0f;", "float decayed = best_v - (float)gap / 4000.0f;"),
    ("decay_added", "float decayed = best_v - (float)gap / 3000.0f;", "float decayed = best_v + (float)gap / 3000.0f;"),
    ("confident_margin_3", "int confident = !same && decayed > (blank_v - 4.0f);", "int confident = !same && decayed > (blank_v - 3.0f);"),
    ("confident_margin_5", "int confident = !same && decayed > (blank_v - 4.0f);", "int confident = !same && decayed > (blank_v - 5.0f);"),
    ("confident_not_strict", "int confident = !same && decayed > (blank_v - 4.0f);", "int confident = !same && decayed >= (blank_v - 4.0f);"),
    ("confident_on_repeated_token", "int confident = !same && decayed > (blank_v - 4.0f);", "int confident = decayed > (blank_v - 4.0f);"),
    ("silence_after_2160", "if (gap >= 2200) {", "if (gap >= 2160) {"),
    ("silence_strictly_after_2200", "if (gap >= 2200) {", "if (gap > 2200) {"),
    ("silence_does_not_finalize", "if (gap >= 2200) {\n            finalize_all(s);", "if (gap >= 2200) {\n            ;"),
    ("silence_keeps_context", "finalize_all(s);\n            clear_context(s);\n            emit_silence(s);\n        } else if", "finalize_all(s);\n            ;\n            emit_silence(s);\n        } else if"),
    ("silence_not_reported", "clear_context(s);\n            emit_silence(s);\n        } else if", "clear_context(s);\n            ;\n        } else if"),
    ("provisional_logprob_minus_7", "tok.logprob -= 8.0f;", "tok.logprob -= 7.0f;"),
    ("provisional_token_stays", "if (emit_partial(s, &tok, 0)) s->head--;", "emit_partial(s, &tok, 0);"),
    ("provisional_always_withdrawn", "if (emit_partial(s, &tok, 0)) s->head--;", "emit_partial(s, &tok, 0); s->head--;"),
    ("no_partial_refresh", "} else {\n            emit_partial(s, NULL, 0);\n        }", "} else {\n            ;\n        }"),
    # ---- chunk loop                                                                                           (:449-454)
    ("early_emit_in_round_1_too", "if (decide(s, round == 0 ? 1.0f : 0.0f)) break;", "if (decide(s, round <= 1 ? 1.0f : 0.0f)) break;"),
    ("early_emit_never", "if (decide(s, round == 0 ? 1.0f : 0.0f)) break;", "if (decide(s, 0.0f)) break;"),
    ("two_rounds_per_chunk", "for (int round = 0; round < 3; ++round) {", "for (int round = 0; round < 2; ++round) {"),
    ("four_rounds_per_chunk", "for (int round = 0; round < 3; ++round) {", "for (int round = 0; round < 4; ++round) {"),
    ("rounds_go_on_after_blank", "if (decide(s, round == 0 ? 1.0f : 0.0f)) break;", "decide(s, round == 0 ? 1.0f : 0.0f);"),
    # ---- finalize_all / finalize_before_word                                                                  (:199-255)
    ("final_reported_as_partial", "s->handler(s->ud, 2, s->head, s->active);\n    s->last_call_head", "s->handler(s->ud, 1, s->head, s->active);\n    s->last_call_head"),
    ("final_keeps_tokens", "s->last_call_head = s->head;\n    s->head = 0;", "s->last_call_head = s->head;"),
    ("word_boundary_shortcut_removed", "if (incoming->flags & 1) { finalize_all(s); return; }", ";"),
    ("word_search_from_head_minus_2", "for (size_t i = s->head - 1; i > 2; --i)", "for (size_t i = s->head - 2; i > 2; --i)"),
    ("word_search_down_to_2", "for (size_t i = s->head - 1; i > 2; --i)", "for (size_t i = s->head - 1; i > 1; --i)"),
    ("word_search_down_to_4", "for (size_t i = s->head - 1; i > 2; --i)", "for (size_t i = s->head - 1; i > 3; --i)"),
    ("word_search_takes_lowest", "if (s->active[i].flags & 1) { start = i; break; }", "if (s->active[i].flags & 1) { start = i; }"),
    ("word_search_failure_keeps_everything", "if (start == ORC_MAX_ACTIVE) { finalize_all(s); return; }", "if (start == ORC_MAX_ACTIVE) { return; }"),
    ("word_final_reported_as_partial", "s->handler(s->ud, 2, start, s->active);", "s->handler(s->ud, 1, start, s->active);"),
    ("word_final_one_token_more", "s->handler(s->ud, 2, start, s->active);", "s->handler(s->ud, 2, start + 1, s->active);"),
    ("current_word_not_moved", "memmove(s->active, &s->active[start], sizeof(OrcToken) * (s->head - start));", ";"),
    ("head_not_reduced", "s->head -= start;", ";"),
    # ---- emit_silence / emit_partial / clear_context / push_context                                           (:181-196, 257-301)
    ("silence_repeated", "if (s->emitted_silence) return;", ";"),
    ("silence_flag_not_set", "s->emitted_silence = 1;\n    s->handler(s->ud, 4, 0, NULL);", "s->handler(s->ud, 4, 0, NULL);"),
    ("dedupe_ignores_token", "if (!force && s->last_call_head == s->head + 1 && s->active[s->head].id == tok->id) return 0;", "if (!force && s->last_call_head == s->head + 1) return 0;"),
    ("dedupe_ignores_head", "if (!force && s->last_call_head == s->head + 1 && s->active[s->head].id == tok->id) return 0;", "if (!force && s->active[s->head].id == tok->id) return 0;"),
    ("dedupe_compares_head", "if (!force && s->last_call_head == s->head + 1 && s->active[s->head].id == tok->id) return 0;", "if (!force && s->last_call_head == s->head && s->active[s->head].id == tok->id) return 0;"),
    ("dedupe_also_when_forced", "if (!force && s->last_call_head == s->head + 1 && s->active[s->head].id == tok->id) return 0;", "if (s->last_call_head == s->head + 1 && s->active[s->head].id == tok->id) return 0;"),
    ("refresh_always", "if (!force && s->last_call_head == s->head) return 0;", ";"),
    ("refresh_never", "if (!force && s->last_call_head == s->head) return 0;", "if (!force) return 0;"),
    ("partial_head_not_recorded", "s->handler(s->ud, 1, s->head, s->active);\n    s->last_call_head = s->head;\n    return 1;", "s->handler(s->ud, 1, s->head, s->active);\n    return 1;"),
    ("clear_context_tests_context1", "if (s->ctx[0] == s->P->blank_id) return;", "if (s->ctx[1] == s->P->blank_id) return;"),
    ("clear_context_always", "if (s->ctx[0] == s->P->blank_id) return;", ";"),
    ("clear_context_one_push", "for (int i = 0; i < s->ctx_n; ++i) push_context(s, s->P->blank_id);\n}", "for (int i = 0; i < 1; ++i) push_context(s, s->P->blank_id);\n}"),
    ("context_not_shifted", "for (int i = 0; i + 1 < s->ctx_n; ++i) s->ctx[i] = s->ctx[i + 1];", ";"),
    # ---- flush                                                                                                (:547-564)
    ("flush_does_not_finalize", "while (orc_fbank_flush(s->fb)) drain_chunks(s);\n    finalize_all(s);", "while (orc_fbank_flush(s->fb)) drain_chunks(s);\n    ;"),
    ("flush_keeps_context", "finalize_all(s);\n    clear_context(s);\n    emit_silence(s);\n}", "finalize_all(s);\n    ;\n    emit_silence(s);\n}"),
    ("flush_without_silence", "clear_context(s);\n    emit_silence(s);\n}", "clear_context(s);\n    ;\n}"),
]

# Mutants that no input can expose (with context_size == 2 and the reachable states of the machine); they must survive.
EQUIVALENT = [
    # active_token_head >= 71 after aas_finalize_previous_words is unreachable: the list is emptied (head = 0) or shortened by
    # start_of_word >= 3 (head <= 68) -- state_machine_cases.py, "No room left" note (:390-394)
    ("no_room_left_branch_removed", "if (s->head >= (ORC_MAX_ACTIVE - 1)) s->head = 0;", ";"),
    # last_handler_call_head after a FINAL (:208): the value it replaces is N (after a plain PARTIAL of the N tokens) or N + 1 (after a
    # provisional token was shown).  With head = 0 afterwards the two later comparisons are `== head` (N and N + 1 are both != 0: same
    # answer) and `== head + 1` (differs only for N = 1, and then the second condition compares the new provisional token with
    # active_tokens[0] = the finalised token, which is context[1] -- so the round is "equal to previous" and never provisional)
    ("final_head_not_recorded", "s->last_call_head = s->head;\n    s->head = 0;", "s->head = 0;"),
    # the scan starts below every real logit (the scripted planes use -1000): -9999999999 vs -3e38 never matters
    ("argmax_start_value", "float best_v = -9999999999.0f;", "float best_v = -3.0e38f;"),
]


def build_objects(tmp):
    """the oracle's other translation units, compiled once"""
    objs = []
    for f in ("orc_fbank.c", "orc_file.c", "orc_onnx.c"):
        o = os.path.join(tmp, f[:-2] + ".o")
        subprocess.check_call(["gcc"] + CFLAGS + ["-I", ORACLE, "-c", os.path.join(ORACLE, f), "-o", o])
        objs.append(o)
    return objs


def run_mutant(tmp, objs, src, name, old, new, model_path):
    assert src.count(old) == 1, "mutant %s: the text to replace occurs %d times in orc_session.c" % (name, src.count(old))
    d = os.path.join(tmp, name)
    os.makedirs(d)
    with open(os.path.join(d, "orc_session.c"), "w") as f:
        f.write(src.replace(old, new))
    so = os.path.join(d, "liborc_mut.so")
    r = subprocess.run(["gcc"] + CFLAGS + ["-I", ORACLE, "-shared", "-o", so, os.path.join(d, "orc_session.c")] + objs + ["-lm"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        return "BUILD FAILED", r.stdout.decode()[-400:]
    env = dict(os.environ, APRIL_ORC_SO=so)
    try:
        w = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mutant_worker.py"), model_path], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    except subprocess.TimeoutExpired:
        return "KILLED", "time-out (the mutant does not terminate)"
    out = w.stdout.decode().strip().splitlines()
    last = out[-1] if out else ""
    if w.returncode == 0 and last == "SURVIVED":
        return "SURVIVED", ""
    return "KILLED", last[:200] if last.startswith("KILLED") else "worker exit %d: %s" % (w.returncode, last[:160])


def run_all(verbose=False, model_path=None):
    """returns (killed, survivors, equivalent_killed, build_failures)"""
    src = open(os.path.join(ORACLE, "orc_session.c")).read()
    tmp = tempfile.mkdtemp(prefix="april_mutants_")
    try:
        if model_path is None:
            sys.path.insert(0, ROOT)
            from april_asr_amd import synth_model as SM
            model_path = os.path.join(tmp, "tiny.april")
            SM.write_model(model_path, SM.TINY_DIMS)
        objs = build_objects(tmp)
        status, why = run_mutant(tmp, objs, src, "identity", "static int decide(", "static int decide(", model_path)
        assert status == "SURVIVED", "the unmutated oracle fails its own fixtures: %s" % why
        killed, survivors, failures, eq_killed = [], [], [], []
        for name, old, new in MUTANTS:
            status, why = run_mutant(tmp, objs, src, name, old, new, model_path)
            if verbose:
                print("%-40s %s %s" % (name, status, why))
            (killed if status == "KILLED" else survivors if status == "SURVIVED" else failures).append((name, why))
        for name, old, new in EQUIVALENT:
            status, why = run_mutant(tmp, objs, src, "eq_" + name, old, new, model_path)
            if verbose:
                print("%-40s %s %s   (listed as equivalent)" % (name, status, why))
            if status != "SURVIVED":
                eq_killed.append((name, why))
        return killed, survivors, eq_killed, failures
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    k, s, e, f = run_all(verbose="-v" in sys.argv)
    print("%d mutants: %d killed, %d survived, %d failed to build; %d equivalent mutants, %d of them unexpectedly killed"
          % (len(MUTANTS), len(k), len(s), len(f), len(EQUIVALENT), len(e)))
    for name, _ in s:
        print("SURVIVOR:", name)
    for name, why in f:
        print("BUILD FAILURE:", name, why)
    for name, why in e:
        print("NOT EQUIVALENT AFTER ALL:", name, why)
    sys.exit(1 if (s or e or f) else 0)
