"""Hand-derived pins of the reference's result state machine (src/april_session.c:199-429, 547-564).

DATA, not code under test: every expectation below was worked out by hand from the reference source, line by line (the
derivation is in the comments, reference line numbers in parentheses); nothing here was produced by running the oracle or
the product.  april_session.c itself cannot be compiled in this image (it includes onnxruntime_c_api.h), so these cases are
what pins the oracle's restatement (oracle/orc_session.c), the product's host state machine (csrc/session.cc `Greedy`) and
the device's copy of the decision (csrc/kernels_misc.hip `decide_kernel`) to the reference's behaviour.

Vocabulary symbols (resolved by the tests against the model's token table):
    W1, W2   word-start tokens (text[0] == ' ', longer than 2 characters, second character not a digit)  -> WORD_BOUNDARY flag (:336)
    C1, C2   continuation tokens (no leading space, alphabetic, longer than 1 character)
    DOT "."  COMMA ","  D2 "2" (digit-start: text[0] in '0'..'9', :347)
Flags: 1 = APRIL_TOKEN_FLAG_WORD_BOUNDARY_BIT, 2 = APRIL_TOKEN_FLAG_SENTENCE_END_BIT.

A case is a list of PHASES.  ("chunks", [...]) = joiner rounds grouped the way the reference's chunk loop consumes them
(:449-454: up to 3 rounds per chunk, early_emit 1, 0, 0, the loop stops at the first blank round); chunk j (1-based, flush
chunks included) runs at current_time_ms = 40 j (:442-443).  A round is (token, max logit, blank logit): the scripted
joiner puts `max` on that token, `blank` on the blank id and -1000 everywhere else.  A round with a fourth element
(token, max, blank, other) puts `max` on `other` as well: an arg-max tie, which the reference's strict `>` scan (:311-320) gives to
the LOWER token id -- the expectations name the winner (the tests resolve W1 < W2 and C1 < C2 by construction).  ("flush", filler) = aas_flush (:547-564):
the padded chunks the flush itself runs (their number depends on the fbank state) all see the `filler` round.
("after", filler) = everything beyond the script sees `filler`.

Expected events: (kind, [token...]) with token = (symbol, logprob, flags, chunk) -- `chunk` is the chunk whose time stamp the
token carries (time_ms = 40 * chunk, :333); chunks of a later phase are written ("post", k) = the k-th chunk after the
flush (the tests know how many chunks the flush ran).
Expected rounds (for the device's decision and the host's replay): per scripted round
    (is_blank, (context[0], context[1]) after the round, last active token after the round or None)
where "last active token" is active_tokens[active_token_head - 1] when active_token_head > 0 (:345-351) -- the device keeps
only that (GreedyState.last_tok)."""

P, F, S = "PARTIAL", "FINAL", "SILENCE"
BLK = "<blk>"


def _t(sym, lp, fl, ch):
    return (sym, lp, fl, ch)


# ------------------------------------------------------------------------------------------------------------------
# 72-token overflow (MAX_ACTIVE_TOKENS = 72, src/april_session.h:30; :365, :213-255)
#
# 71 tokens are emitted without any FINAL in between: every round is (tok, 5.0, 0.0):
#   round 0: early_emit 1 -> is_blank = (0 - 1) > 5 = false;  rounds 1, 2: early_emit 0 -> (0 > 5) = false   (:329-330)
#   (a token equal to the previous one only zeroes early_emit, :326-327: still non-blank)
# so every chunk emits 3 tokens: token k (0-based) is emitted in chunk k // 3 + 1 with time 40 (k // 3 + 1), logprob 5.0,
# flags 1 iff its text starts with ' '.  Each emission: aas_update_context (:364), is_final = head >= 71 (:365) false
# while head <= 70, no sentence check fires (no '.', '!', '?' tokens), aas_emit_token(force) appends and calls
# PARTIAL with head tokens (:278-293) -> events PARTIAL(tokens[0..k]) for k = 0..70.
# Token 71 (chunk 24, round 2) arrives with active_token_head == 71 -> is_final = true (:365) -> aas_finalize_previous_words.
def _seq(word_starts):
    toks = []
    for k in range(71):
        sym = word_starts.get(k) or ("C1" if k % 2 else "C2")
        toks.append(_t(sym, 5.0, 1 if sym.startswith("W") else 0, k // 3 + 1))
    return toks


def _overflow_case(name, word_starts, last_sym, final_count, doc):
    toks = _seq(word_starts)
    last = _t(last_sym, 5.0, 1 if last_sym.startswith("W") else 0, 24)
    rounds = [(t[0], 5.0, 0.0) for t in toks] + [(last_sym, 5.0, 0.0)]
    chunks = [rounds[i:i + 3] for i in range(0, 72, 3)]
    chunks.append([("C1", -20.0, 10.0)])                      # chunk 25: blank, max' = -20 - 40/3000 < 10 - 4: not confident (:409);
    #                                                           aas_emit_token(NULL): last_handler_call_head == head -> no call (:287-291)
    events = [(P, toks[:k + 1]) for k in range(71)]
    if final_count == 71:
        # aas_finalize_tokens (:199-211): FINAL with all 71, head = 0; then the new token alone
        events += [(F, toks[:71]), (P, [last])]
    else:
        # FINAL "excluding the current word" (:236-241) = the first start_of_word tokens; memmove of the rest to the front
        # (:244-251); head -= start_of_word; then the new token is appended (:396)
        events += [(F, toks[:final_count]), (P, toks[final_count:] + [last])]
    syms = [t[0] for t in toks] + [last_sym]
    exp_rounds = [(False, (BLK if k == 0 else syms[k - 1], syms[k]), syms[k]) for k in range(72)]
    exp_rounds.append((True, (syms[70], syms[71]), syms[71]))
    return dict(name=name, doc=doc, phases=[("chunks", chunks)], events=events, rounds=exp_rounds)


CASES = [
    # new token starts a word (:216-218): aas_finalize_tokens -> FINAL(71), head = 0, last_handler_call_head = 71;
    # second is_final test (:390) 0 >= 71 false; aas_emit_token(force): active[0] = token, PARTIAL(1)
    _overflow_case("overflow_new_token_is_word_boundary", {0: "W1"}, "W2", 71,
                   "71 active tokens, token 72 starts a word: FINAL with all 71, then PARTIAL with the new token alone"),
    # new token continues a word; the search (:224-230) walks i = 70 .. 3 and finds the boundary flag at i = 68:
    # FINAL(68 tokens 0..67); memmove tokens 68..70 to the front; head = 71 - 68 = 3; is_final (:390) false; the new token is
    # appended -> PARTIAL(4 tokens: 68, 69, 70, new).  last_handler_call_head is NOT touched by this path (stays 71) --
    # irrelevant here because the emit is forced.
    _overflow_case("overflow_continuation_word_start_found", {0: "W1", 68: "W2"}, "C1", 68,
                   "71 active tokens, token 72 continues the word that started at index 68: FINAL excludes that word"),
    # the smallest index the search can see is 3 (`i > 2`, :226): FINAL(3 tokens), PARTIAL(68 + 1 tokens)
    _overflow_case("overflow_continuation_word_start_at_3", {0: "W1", 3: "W2"}, "C1", 3,
                   "word start at index 3, the lowest index the `i > 2` search reaches"),
    # word starts only at indices 0 and 2: the search stops above 2, start_of_word stays MAX_ACTIVE_TOKENS (:232-235)
    # -> aas_finalize_tokens: FINAL(71), then PARTIAL(new token alone)
    _overflow_case("overflow_continuation_no_word_start_above_2", {0: "W1", 2: "W2"}, "C1", 71,
                   "word starts only at indices 0 and 2 (below the search's reach): the whole list is finalised"),
    # "No room left even after finalizing previous words" (:390-394) is unreachable: active_token_head only grows in
    # aas_emit_token and is <= 71 at rest (an emission at head >= 71 first goes through aas_finalize_previous_words, which
    # leaves head = 0 or head - start_of_word <= 71 - 3); a provisional emission restores head (:421-424).  No case.

    dict(
        name="digit_dot_then_word_retro_sentence_end",
        doc='"W1 2 ." + word: the "." after a digit is no punctuation (:345-351), gets SENTENCE_END retroactively and forces FINAL when a word follows (:369-388)',
        phases=[("chunks", [
            # chunk 1 (t = 40)
            [("W1", 5.0, 0.0),       # r0 non-blank -> PARTIAL [W1]; context [blk, W1]
             ("D2", 5.0, 0.0),       # r1 non-blank -> PARTIAL [W1, 2]; context [W1, 2]
             ("DOT", 7.0, 10.0)],    # r2 early 0: blank by logits (10 > 7).  "." is single-char punctuation (:340-342) but head = 2 > 0 and
                                     #    active[1] = "2" starts with a digit -> is_end_of_sentence = is_punctuation = false (:345-351): the
                                     #    override (:356-358) does not apply -> blank.  Blank branch: time_since = 0, max' = 7 > 10 - 4 and not
                                     #    equal to previous -> reasonably confident (:409): provisional token, logprob 7 - 8 = -1 (:419),
                                     #    flags 0 (no SENTENCE_END after a digit).  aas_emit_token(!force): last_call_head 2 != head + 1 -> append,
                                     #    PARTIAL [W1, 2, .'], last_call_head = 3, then head-- -> 2 (:420-423)
            # chunk 2 (t = 80)
            [("DOT", 5.0, 0.0),      # r0 early 1: (0 - 1) > 5 false -> non-blank.  Still after a digit: flags 0.  context [2, .].  No sentence
                                     #    check (token is no word boundary, :368).  PARTIAL [W1, 2, .(5.0, 0, t 80)]
             ("W2", 5.0, 0.0),       # r1 non-blank, WORD_BOUNDARY.  head = 3 > 0: last token "." is a single '.', -> last_token_end_of_sentence
                                     #    (:372-373); its flags lack SENTENCE_END -> set now (:379-381); is_final = true (:384-386).
                                     #    aas_finalize_previous_words: boundary -> FINAL [W1, 2, .(flags 2)], head 0.  PARTIAL [W2]
             ("C1", -20.0, 10.0)],   # r2 blank, not confident; aas_emit_token(NULL): last_call_head 1 == head 1 -> nothing
        ])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("D2", 5.0, 0, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("D2", 5.0, 0, 1), _t("DOT", -1.0, 0, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("D2", 5.0, 0, 1), _t("DOT", 5.0, 0, 2)]),
            (F, [_t("W1", 5.0, 1, 1), _t("D2", 5.0, 0, 1), _t("DOT", 5.0, 2, 2)]),
            (P, [_t("W2", 5.0, 1, 2)]),
        ],
        rounds=[
            (False, (BLK, "W1"), "W1"), (False, ("W1", "D2"), "D2"), (True, ("W1", "D2"), "D2"),
            (False, ("D2", "DOT"), "DOT"), (False, ("DOT", "W2"), "W2"), (True, ("DOT", "W2"), "W2"),
        ],
    ),

    dict(
        name="punctuation_override_and_digit_suppression",
        doc="the punctuation override (:356-358) fires for '.' after a word and for ',' after a digit, not for '.' after a digit, not for a repeated token",
        phases=[("chunks", [
            # chunk 1 (t = 40)
            [("W1", 5.0, 0.0),       # r0 -> PARTIAL [W1]; context [blk, W1]
             ("DOT", 7.0, 10.0),     # r1 blank by logits (10 > 7); "." is end of sentence, last token W1 is no digit; context[1] = W1 != blk so
                                     #    was_context_cleared is false (:322); not equal to previous; 7 > 10 - 3.5 -> is_blank = false (:356-358).
                                     #    Emitted with SENTENCE_END (:353): PARTIAL [W1, .(7.0, 2)]; context [W1, .]
             ("W2", 5.0, 0.0)],      # r2 word boundary after "." (already flagged, no retro-flag) -> is_final -> FINAL [W1, .], PARTIAL [W2]
            # chunk 2 (t = 80)
            [("D2", 5.0, 0.0),       # r0 -> PARTIAL [W2, 2]; context [W2, 2]
             ("DOT", 7.0, 10.0)],    # r1 "." after a digit: no punctuation -> no override -> blank; provisional PARTIAL [W2, 2, .'(-1, 0)]
            # chunk 3 (t = 120)
            [("COMMA", 7.0, 10.0),   # r0 early 1: 9 > 7 blank by logits; "," is punctuation (:342); the digit rule only exempts '.', (:348);
                                     #    not cleared, not equal, 7 > 6.5 -> non-blank, flags 0: PARTIAL [W2, 2, ,]; context [2, ,]
             ("COMMA", 9.0, 10.0)],  # r1 equal to previous -> no override, blank (10 > 9); not "reasonably confident" either (:409 needs !equal);
                                     #    aas_emit_token(NULL): last_call_head 3 == head 3 -> nothing
        ])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("DOT", 7.0, 2, 1)]),
            (F, [_t("W1", 5.0, 1, 1), _t("DOT", 7.0, 2, 1)]),
            (P, [_t("W2", 5.0, 1, 1)]),
            (P, [_t("W2", 5.0, 1, 1), _t("D2", 5.0, 0, 2)]),
            (P, [_t("W2", 5.0, 1, 1), _t("D2", 5.0, 0, 2), _t("DOT", -1.0, 0, 2)]),
            (P, [_t("W2", 5.0, 1, 1), _t("D2", 5.0, 0, 2), _t("COMMA", 7.0, 0, 3)]),
        ],
        rounds=[
            (False, (BLK, "W1"), "W1"), (False, ("W1", "DOT"), "DOT"), (False, ("DOT", "W2"), "W2"),
            (False, ("W2", "D2"), "D2"), (True, ("W2", "D2"), "D2"),
            (False, ("D2", "COMMA"), "COMMA"), (True, ("D2", "COMMA"), "COMMA"),
        ],
    ),

    dict(
        name="silence_keeps_context_that_starts_with_blank",
        doc="aas_clear_context returns early when context[0] == blank (:296-297): after ONE token and 2.2 s of silence the context stays [blk, tok]",
        phases=[("chunks",
                 # chunk 1 (t = 40): W1 emitted (last_emission 40, context [blk, W1]); r1 blank, nothing to report
                 [[("W1", 5.0, 0.0), ("C1", -20.0, 10.0)]]
                 # chunks 2..55 (t = 80..2200): blank, time_since = t - 40 <= 2160 < 2200, not confident, last_call_head == head: nothing
                 + [[("C1", -20.0, 10.0)]] * 54
                 # chunk 56 (t = 2240): time_since = 2200 >= 2200 (:411) -> FINAL [W1]; aas_clear_context: context[0] == blk -> return,
                 # the context is still [blk, W1] (:296-297); SILENCE (:257-268)
                 + [[("C1", -20.0, 10.0)]]
                 # chunk 57 (t = 2280): W1 again with max 7.0 vs blank 7.5: context[1] == W1 -> is_equal_to_previous -> early_emit = 0
                 # (:326-327) -> 7.5 > 7.0 -> blank.  (Had the context been cleared, early_emit 1 would make it non-blank: 6.5 > 7 false.)
                 # Blank branch: time_since = 2240 >= 2200: nothing active, silence already emitted -> no call
                 + [[("W1", 7.0, 7.5)]]
                 # chunk 58 (t = 2320): C1, not equal -> early 1 -> (7.5 - 1) > 7 false -> non-blank: PARTIAL [C1]; context [W1, C1];
                 # r1 blank (time_since 0), nothing to report
                 + [[("C1", 7.0, 7.5), ("C2", -20.0, 10.0)]])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (F, [_t("W1", 5.0, 1, 1)]),
            (S, []),
            (P, [_t("C1", 7.0, 0, 58)]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (True, (BLK, "W1"), "W1")] + [(True, (BLK, "W1"), "W1")] * 54
        + [(True, (BLK, "W1"), None), (True, (BLK, "W1"), None), (False, ("W1", "C1"), "C1"), (True, ("W1", "C1"), "C1")],
    ),

    dict(
        name="silence_clears_context_after_two_tokens",
        doc="with two tokens emitted context[0] != blank: 2.2 s of silence resets the context to [blk, blk] (:299-300)",
        phases=[("chunks",
                 # chunk 1: W1, C1 emitted (context [W1, C1]), r2 blank
                 [[("W1", 5.0, 0.0), ("C1", 5.0, 0.0), ("C2", -20.0, 10.0)]]
                 + [[("C2", -20.0, 10.0)]] * 54
                 # chunk 56 (t = 2240): FINAL [W1, C1]; context[0] = W1 != blk -> two pushes of blk -> [blk, blk]; SILENCE
                 + [[("C2", -20.0, 10.0)]]
                 # chunk 57 (t = 2280): C1 7.0 vs 7.5: context[1] == blk, not equal -> early 1 -> non-blank: PARTIAL [C1].  (With the stale
                 # context [W1, C1] it would have been "equal to previous" -> blank.)  r1 blank, nothing
                 + [[("C1", 7.0, 7.5), ("C2", -20.0, 10.0)]])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 1)]),
            (F, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 1)]),
            (S, []),
            (P, [_t("C1", 7.0, 0, 57)]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (False, ("W1", "C1"), "C1"), (True, ("W1", "C1"), "C1")] + [(True, ("W1", "C1"), "C1")] * 54
        + [(True, (BLK, BLK), None), (False, (BLK, "C1"), "C1"), (True, (BLK, "C1"), "C1")],
    ),

    dict(
        name="provisional_then_flush_final_then_same_provisional",
        doc="the de-duplication of a repeated provisional token (:272-276) compares against active_tokens[head], which a FINAL leaves behind",
        phases=[
            ("chunks", [
                # chunk 1 (t = 40): W1 emitted; r1: C1 7.0 vs blank 10.0, early 0: blank; time_since 0, 7 > 6, not equal -> provisional:
                # last_call_head 1 != head + 1 = 2 -> active[1] = C1', PARTIAL [W1, C1'(-1.0)], last_call_head = 2, head back to 1
                [("W1", 5.0, 0.0), ("C1", 7.0, 10.0)],
            ]),
            # aas_flush: every padded chunk sees the same provisional C1 (early 1: 9 > 7 blank; time_since a few x 40 ms, 7 - dt/3000 > 6):
            # last_call_head 2 == head + 1 and active[1].token == C1 -> suppressed (:272-276).  End of flush (:561-563): FINAL [W1],
            # last_call_head = 1, head = 0; context [blk, W1] starts with blank -> kept; SILENCE (emitted_silence was false since W1)
            ("flush", ("C1", 7.0, 10.0)),
            ("chunks", [
                # first chunk after the flush: the same provisional C1: last_call_head 1 == head + 1 = 1 but active[0] is still W1
                # -> not a repeat: active[0] = C1', PARTIAL [C1'(-1.0)], last_call_head = 1, head back to 0
                [("C1", 7.0, 10.0)],
                # next chunk: last_call_head 1 == head + 1 and active[0].token == C1 -> suppressed
                [("C1", 7.0, 10.0)],
            ]),
            ("after", ("C1", 7.0, 10.0)),      # anything further: suppressed the same way (or, past 2.2 s, the silence branch with nothing to report)
        ],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", -1.0, 0, 1)]),
            (F, [_t("W1", 5.0, 1, 1)]),
            (S, []),
            (P, [_t("C1", -1.0, 0, ("post", 1))]),
        ],
        # rounds of the scripted chunks only (flush chunks: all blank, context unchanged); after the flush the device has
        # forgotten the last token (None) and kept the context
        rounds=[(False, (BLK, "W1"), "W1"), (True, (BLK, "W1"), "W1"), "FLUSH", (True, (BLK, "W1"), None), (True, (BLK, "W1"), None)],
        flush_state=((BLK, "W1"), None),
    ),

    dict(
        name="flush_with_active_tokens_clears_context",
        doc="flush with active tokens (:561-563): FINAL with them, context reset to [blk, blk] because it does not start with blank, SILENCE",
        phases=[
            ("chunks", [[("W1", 5.0, 0.0), ("C1", 5.0, 0.0), ("C2", -20.0, 10.0)]]),      # PARTIAL [W1], PARTIAL [W1, C1]; context [W1, C1]
            ("flush", ("C2", -20.0, 10.0)),        # padded chunks: blank, not confident, last_call_head 2 == head 2 -> nothing.  Then FINAL [W1, C1],
                                                   # context[0] = W1 != blk -> [blk, blk], SILENCE
            ("chunks", [
                # C1 7.0 vs 7.5 on a cleared context: not equal to previous (context[1] = blk) -> early 1 -> non-blank: PARTIAL [C1];
                # r1 blank: nothing
                [("C1", 7.0, 7.5), ("C2", -20.0, 10.0)],
            ]),
            ("after", ("C2", -20.0, 10.0)),
        ],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 1)]),
            (F, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 1)]),
            (S, []),
            (P, [_t("C1", 7.0, 0, ("post", 1))]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (False, ("W1", "C1"), "C1"), (True, ("W1", "C1"), "C1"), "FLUSH",
                (False, (BLK, "C1"), "C1"), (True, (BLK, "C1"), "C1")],
        flush_state=((BLK, BLK), None),
    ),

    # ---- round 6: cases added for the mutants of tests/mutate_state_machine.py that the cases above let live --------------------
    # (each was derived by hand from the reference lines quoted, like the ones above; the comment names the mutants it kills)
    _overflow_case("overflow_word_boundary_beats_word_search", {0: "W1", 68: "W2"}, "W1", 71,
                   "the new token starts a word: everything is finalised (:216-218) even though the search would find a word start at 68"),
    # word starts at 30 and 70: the search runs DOWN from head - 1 = 70 (:226) and stops at the first hit (`break`, :229): FINAL(70
    # tokens 0..69), token 70 moves to the front, head = 1, the new token is appended -> PARTIAL [tok 70, new]
    _overflow_case("overflow_search_starts_at_last_token_and_stops_at_first_hit", {0: "W1", 30: "W1", 70: "W2"}, "C1", 70,
                   "word starts at 30 and at 70 (the last active token): the search starts at head - 1 and takes the highest one"),

    dict(
        name="argmax_tie_goes_to_lower_id",
        doc="two tokens share the maximum: the strict `>` scan (:311-320) keeps the first = lower id",
        phases=[("chunks", [
            # chunk 1: W1 and W2 both at 5.0 -> W1 (lower id); r1: C2 and C1 both at 5.0 -> C1; r2 blank, nothing to report
            [("W2", 5.0, 0.0, "W1"), ("C2", 5.0, 0.0, "C1"), ("C2", -20.0, 10.0)],
        ])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 1)]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (False, ("W1", "C1"), "C1"), (True, ("W1", "C1"), "C1")],
    ),

    dict(
        name="override_and_provisional_thresholds",
        doc="the exact margins: override needs max > blank - 3.5 (:357), a provisional token max' > blank - 4.0 (:409), both strict",
        phases=[("chunks", [
            # chunk 1 (t = 40): W1 emitted (last_emission 40).  r1 "." 6.0 vs 10.0, early 0: blank by logits; punctuation after a word,
            # context not cleared, not equal: override iff 6.0 > 10 - 3.5 = 6.5 -> no.  Blank branch: time_since 0, max' = 6.0 > 10 - 4 = 6.0
            # is false (strict) -> not confident; aas_emit_token(NULL): last_call_head 1 == head 1 -> nothing
            [("W1", 5.0, 0.0), ("DOT", 6.0, 10.0)],
            # chunk 2 (t = 80): "." 6.5 vs 10.0, early 1: 9 > 6.5 blank by logits; override iff 6.5 > 6.5 -> no (strict).  Blank branch:
            # time_since 40, max' = 6.5 - 40/3000 = 6.48667 > 6.0 -> provisional, logprob 6.5 - 8 = -1.5, SENTENCE_END (:353): PARTIAL
            # [W1, .'], head back to 1
            [("DOT", 6.5, 10.0)],
            # chunk 3 (t = 120): "." 6.75 vs 10.0: 9 > 6.75 blank by logits; override: 6.75 > 6.5 -> non-blank (:356-358): PARTIAL [W1, .(6.75, 2)]
            # (forced: the provisional de-duplication does not apply); r1 blank -20: nothing (last_call_head 2 == head 2)
            [("DOT", 6.75, 10.0), ("C1", -20.0, 10.0)],
        ])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("DOT", -1.5, 2, 2)]),
            (P, [_t("W1", 5.0, 1, 1), _t("DOT", 6.75, 2, 3)]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (True, (BLK, "W1"), "W1"), (True, (BLK, "W1"), "W1"),
                (False, ("W1", "DOT"), "DOT"), (True, ("W1", "DOT"), "DOT")],
    ),

    dict(
        name="override_needs_uncleared_context",
        doc="punctuation on a cleared context ([blk, blk], :322) is not forced through (:356): it stays blank and only shows as a provisional token",
        phases=[("chunks", [
            # chunk 1 (t = 40): "." 7.0 vs 10.0, early 1: 9 > 7 blank by logits; punctuation (head = 0: no digit rule), 7 > 6.5, not equal --
            # but context[1] == blank -> was_context_cleared -> no override.  Blank branch: time_since = 40 - 0, max' = 7 - 40/3000 > 6 ->
            # provisional: PARTIAL [.'(-1.0, SENTENCE_END)], head back to 0
            [("DOT", 7.0, 10.0)],
            # chunk 2: the same again: last_call_head 1 == head + 1 and active[0] is that "." -> suppressed (:272-276)
            [("DOT", 7.0, 10.0)],
        ])],
        events=[(P, [_t("DOT", -1.0, 2, 1)])],
        rounds=[(True, (BLK, BLK), None), (True, (BLK, BLK), None)],
    ),

    dict(
        name="silence_clock_restarts_at_every_emission",
        doc="time_since_last_emission counts from the LAST emitted token (:362) and 2160 ms of it are not yet silence (:411)",
        phases=[("chunks",
                 # chunk 1 (t = 40): W1 (last_emission 40); chunks 2..55 (t = 80..2200): blank, time_since <= 2160 < 2200: nothing
                 [[("W1", 5.0, 0.0), ("C2", -20.0, 10.0)]] + [[("C2", -20.0, 10.0)]] * 54
                 # chunk 56 (t = 2240): C1 non-blank (early 1: -1 > 5 false): the non-blank branch never looks at the clock -> PARTIAL
                 # [W1, C1]; last_emission = 2240.  r1 blank: time_since 0, nothing to report
                 + [[("C1", 5.0, 0.0), ("C2", -20.0, 10.0)]]
                 # chunks 57..110 (t = 2280..4400): blank, time_since = t - 2240 <= 2160: still no silence
                 + [[("C2", -20.0, 10.0)]] * 54
                 # chunk 111 (t = 4440): time_since 2200 -> FINAL [W1, C1], context cleared, SILENCE
                 + [[("C2", -20.0, 10.0)]])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 56)]),
            (F, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 56)]),
            (S, []),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (True, (BLK, "W1"), "W1")] + [(True, (BLK, "W1"), "W1")] * 54
        + [(False, ("W1", "C1"), "C1"), (True, ("W1", "C1"), "C1")] + [(True, ("W1", "C1"), "C1")] * 54 + [(True, (BLK, BLK), None)],
    ),

    dict(
        name="provisional_confidence_decays_with_time",
        doc="max' = max - time_since / 3000 (:408): the same runner-up is shown 1200 ms after the last token and no longer 1520 ms after it; a blank round after a withdrawn provisional token refreshes the PARTIAL (:287-293)",
        phases=[("chunks",
                 # chunk 1 (t = 40): W1; chunks 2..30: blank, nothing
                 [[("W1", 5.0, 0.0), ("C2", -20.0, 10.0)]] + [[("C2", -20.0, 10.0)]] * 29
                 # chunk 31 (t = 1240): C1 6.5 vs 10.0 -> blank; time_since 1200: max' = 6.5 - 0.4 = 6.1 > 6.0 -> provisional PARTIAL
                 # [W1, C1'(-1.5)], last_call_head = 2, head back to 1
                 + [[("C1", 6.5, 10.0)]]
                 # chunk 32: blank, not confident: aas_emit_token(NULL): last_call_head 2 != head 1 -> PARTIAL [W1] (the provisional
                 # token is withdrawn), last_call_head = 1; chunks 33..38: nothing
                 + [[("C2", -20.0, 10.0)]] * 7
                 # chunk 39 (t = 1560): C1 6.5 vs 10.0 again; time_since 1520: max' = 6.5 - 0.50667 = 5.99333 > 6.0 is false -> not
                 # confident; last_call_head 1 == head 1 -> nothing
                 + [[("C1", 6.5, 10.0)]])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", -1.5, 0, 31)]),
            (P, [_t("W1", 5.0, 1, 1)]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (True, (BLK, "W1"), "W1")] + [(True, (BLK, "W1"), "W1")] * 38,
    ),

    dict(
        name="early_emit_bonus_only_in_the_first_round",
        doc="early_emit is 1.0 in round 0 and 0.0 in rounds 1, 2 (:449-454): the same margin is a token in round 0 and blank in round 1",
        phases=[("chunks", [
            # chunk 1: W1; r1 C1 7.0 vs 7.5, early 0: 7.5 > 7.0 -> blank; time_since 0, 7.0 > 3.5, not equal -> provisional PARTIAL [W1, C1'(-1.0)]
            [("W1", 5.0, 0.0), ("C1", 7.0, 7.5)],
            # chunk 2 (t = 80): r0 C1 7.0 vs 7.5, early 1: 6.5 > 7.0 false -> non-blank: PARTIAL [W1, C1(7.0, t 80)] (forced);
            # r1 C2 7.0 vs 7.5, early 0 -> blank; provisional PARTIAL [W1, C1, C2'(-1.0)]
            [("C1", 7.0, 7.5), ("C2", 7.0, 7.5)],
        ])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", -1.0, 0, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", 7.0, 0, 2)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", 7.0, 0, 2), _t("C2", -1.0, 0, 2)]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (True, (BLK, "W1"), "W1"), (False, ("W1", "C1"), "C1"), (True, ("W1", "C1"), "C1")],
    ),

    dict(
        name="flush_then_provisional_equal_to_stale_slot",
        doc="the provisional de-duplication (:272-276) needs BOTH conditions: a stale active_tokens[0] equal to the new provisional token does not suppress it when last_handler_call_head != head + 1",
        phases=[
            ("chunks", [[("W1", 5.0, 0.0), ("C1", 5.0, 0.0), ("C2", -20.0, 10.0)]]),      # PARTIAL [W1], PARTIAL [W1, C1]; context [W1, C1]
            # padded chunks: blank, not confident, last_call_head 2 == head 2 -> nothing.  End of flush: FINAL [W1, C1] (last_call_head = 2,
            # head = 0; active_tokens[0] still holds W1), context -> [blk, blk], SILENCE
            ("flush", ("C2", -20.0, 10.0)),
            ("chunks", [
                # W1 7.0 vs 10.0: early 1: 9 > 7 -> blank; not equal (context[1] = blk); time_since < 2200 and 7 - dt/3000 > 6 (dt <= 1280 ms:
                # at most 28 + 2 chunks since the last token) -> provisional: last_call_head 2 != head + 1 = 1 -> NOT a repeat although
                # active_tokens[0].token == W1: PARTIAL [W1'(-1.0, WORD_BOUNDARY)], last_call_head = 1, head back to 0
                [("W1", 7.0, 10.0)],
                # the same again: last_call_head 1 == head + 1 and active_tokens[0].token == W1 -> suppressed
                [("W1", 7.0, 10.0)],
            ]),
            ("after", ("W1", 7.0, 10.0)),
        ],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 1)]),
            (F, [_t("W1", 5.0, 1, 1), _t("C1", 5.0, 0, 1)]),
            (S, []),
            (P, [_t("W1", -1.0, 1, ("post", 1))]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (False, ("W1", "C1"), "C1"), (True, ("W1", "C1"), "C1"), "FLUSH",
                (True, (BLK, BLK), None), (True, (BLK, BLK), None)],
        flush_state=((BLK, BLK), None),
    ),

    dict(
        name="exact_ties_and_sentence_check_needs_a_word_boundary",
        doc="blank - early_emit == max is NOT blank (strict `>`, :329-330), in round 0 and later; a continuation token after '.' does not finalise the sentence (:368)",
        phases=[("chunks", [
            # chunk 1 (t = 40)
            [("W1", 5.0, 0.0),       # PARTIAL [W1]
             ("DOT", 7.0, 10.0),     # override (:356-358): PARTIAL [W1, .(7.0, SENTENCE_END)]; context [W1, .]
             ("C1", 7.0, 7.0)],      # r2 early 0: 7.0 > 7.0 is false -> non-blank.  C1 is no word boundary: the sentence check (:368-388) is
                                     #    skipped although the last token is "." -> no FINAL: PARTIAL [W1, ., C1]
            # chunk 2 (t = 80)
            [("C2", 6.0, 7.0),       # r0 early 1: (7 - 1) > 6.0 is false -> non-blank: PARTIAL [W1, ., C1, C2]
             ("C1", -20.0, 10.0)],   # blank, not confident, last_call_head 4 == head 4 -> nothing
        ])],
        events=[
            (P, [_t("W1", 5.0, 1, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("DOT", 7.0, 2, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("DOT", 7.0, 2, 1), _t("C1", 7.0, 0, 1)]),
            (P, [_t("W1", 5.0, 1, 1), _t("DOT", 7.0, 2, 1), _t("C1", 7.0, 0, 1), _t("C2", 6.0, 0, 2)]),
        ],
        rounds=[(False, (BLK, "W1"), "W1"), (False, ("W1", "DOT"), "DOT"), (False, ("DOT", "C1"), "C1"),
                (False, ("C1", "C2"), "C2"), (True, ("C1", "C2"), "C2")],
    ),

    dict(
        name="dot_after_a_single_digit_token",
        doc="the digit rule (:345-351) applies from ONE active token on (active_token_head > 0)",
        phases=[("chunks", [
            # chunk 1: "2" emitted (head = 1).  r1 "." 7.0 vs 10.0: blank by logits; head 1 > 0 and the last token starts with a digit -> no
            # punctuation -> no override -> blank; time_since 0, 7 > 6, not equal -> provisional PARTIAL ["2", .'(-1.0, no SENTENCE_END)]
            [("D2", 5.0, 0.0), ("DOT", 7.0, 10.0)],
        ])],
        events=[
            (P, [_t("D2", 5.0, 0, 1)]),
            (P, [_t("D2", 5.0, 0, 1), _t("DOT", -1.0, 0, 1)]),
        ],
        rounds=[(False, (BLK, "D2"), "D2"), (True, (BLK, "D2"), "D2")],
    ),

]
