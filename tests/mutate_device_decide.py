"""Mutation test of the state-machine fixtures against the DEVICE's copy of the search decision (round 6).

csrc/kernels_misc.hip decide_kernel holds the part of aas_process_logits (src/april_session.c:306-429) that the NEXT network call
depends on -- arg-max with the lower id winning ties, blank / non-blank with the early-emit term, the punctuation override and the digit-dot
rule, the context push, the 2.2 s silence that clears the context -- a third transcription beside the oracle's and the host's.  The same
single edits as in tests/mutate_state_machine.py / mutate_product_state_machine.py, each compiled (hipcc on the one device source, linked
with the library's other objects into its own .so) and run through aprilx_run_decide on the GPU (tests/device_decide_mutant_worker.py,
APRIL_ASR_LIB).  A mutant that passes every case SURVIVES; the GPU test fails unless there are none.

usage: python tests/mutate_device_decide.py [-v]        (tests/test_gpu_decide.py runs it inside the gpu suite)
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "april_asr_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
DEVFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-w", "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]

MUTANTS = [
    ("tie_takes_the_higher_id", "const bool take = (oi >= 0) && (best_i < 0 || ov > best || (ov == best && oi < best_i));", "const bool take = (oi >= 0) && (best_i < 0 || ov > best || (ov == best && oi > best_i));"),
    ("cleared_tests_context0", "const bool cleared = st.ctx1 == a.blank;", "const bool cleared = st.ctx0 == a.blank;"),
    ("same_tests_context0", "const bool same = st.ctx1 == tok;", "const bool same = st.ctx0 == tok;"),
    ("same_keeps_early_emit", "const float ee = same ? 0.0f : a.early_emit;", "const float ee = a.early_emit;"),
    ("blank_test_not_strict", "bool is_blank = (bl - ee) > tv;", "bool is_blank = (bl - ee) >= tv;"),
    ("early_emit_added", "bool is_blank = (bl - ee) > tv;", "bool is_blank = (bl + ee) > tv;"),
    ("comma_is_no_punctuation", "bool punct = (tc & (TKC_SENT_END | TKC_COMMA)) != 0;", "bool punct = (tc & TKC_SENT_END) != 0;"),
    ("digit_rule_for_every_punctuation", "(a.tok_class[st.last_tok] & TKC_DIGIT_START) && (tc & TKC_DOT)) punct = false;", "(a.tok_class[st.last_tok] & TKC_DIGIT_START)) punct = false;"),
    ("digit_rule_dropped", "(a.tok_class[st.last_tok] & TKC_DIGIT_START) && (tc & TKC_DOT)) punct = false;", "(a.tok_class[st.last_tok] & TKC_DIGIT_START) && (tc & TKC_DOT)) punct = punct;"),
    ("override_margin_2_5", "tv > (bl - 3.5f)) is_blank = false;", "tv > (bl - 2.5f)) is_blank = false;"),
    ("override_margin_4_5", "tv > (bl - 3.5f)) is_blank = false;", "tv > (bl - 4.5f)) is_blank = false;"),
    ("override_not_strict", "tv > (bl - 3.5f)) is_blank = false;", "tv >= (bl - 3.5f)) is_blank = false;"),
    ("override_on_cleared_context", "if (!cleared && punct && !same && tv", "if (punct && !same && tv"),
    ("override_on_repeated_token", "if (!cleared && punct && !same && tv", "if (!cleared && punct && tv"),
    ("override_for_every_token", "if (!cleared && punct && !same && tv", "if (!cleared && !same && tv"),
    ("emission_time_not_recorded", "            st.last_emit_ms = now;\n", "            ;\n"),
    ("context_not_shifted", "st.ctx0 = st.ctx1; st.ctx1 = tok;", "st.ctx1 = tok;"),
    ("last_token_not_recorded", "            st.last_tok = tok;\n", "            ;\n"),
    ("silence_after_more_than_2200", "if (now - st.last_emit_ms >= 2200u) {", "if (now - st.last_emit_ms > 2200u) {"),
    ("silence_after_2100", "if (now - st.last_emit_ms >= 2200u) {", "if (now - st.last_emit_ms >= 2100u) {"),
    ("silence_after_2300", "if (now - st.last_emit_ms >= 2200u) {", "if (now - st.last_emit_ms >= 2300u) {"),
    ("silence_keeps_the_last_token", "                st.last_tok = -1;\n", "                ;\n"),
    ("clear_context_tests_context1", "if (st.ctx0 != a.blank) { st.ctx0 = a.blank; st.ctx1 = a.blank; rerun = true; }", "if (st.ctx1 != a.blank) { st.ctx0 = a.blank; st.ctx1 = a.blank; rerun = true; }"),
    ("clear_context_half", "if (st.ctx0 != a.blank) { st.ctx0 = a.blank; st.ctx1 = a.blank; rerun = true; }", "if (st.ctx0 != a.blank) { st.ctx1 = a.blank; rerun = true; }"),
]
EQUIVALENT = [
    # last_tok is -1 or the id of an emitted (non-blank) token; the blank is token 0 in every model here, so `>= 0` and `>= 1` admit the same states
    ("digit_rule_last_token_from_1", "if (punct && st.last_tok >= 0 && (a.tok_class[st.last_tok] & TKC_DIGIT_START) && (tc & TKC_DOT)) punct = false;", "if (punct && st.last_tok >= 1 && (a.tok_class[st.last_tok] & TKC_DIGIT_START) && (tc & TKC_DOT)) punct = false;"),
]


def build_variant(tmp, name, src_text, objs):
    cc = os.path.join(tmp, name + ".hip")
    open(cc, "w").write(src_text)
    obj = os.path.join(tmp, name + ".o")
    r = subprocess.run([HIPCC] + DEVFLAGS + ["-c", cc, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        return None, r.stdout.decode()[-400:]
    so = os.path.join(tmp, "lib_" + name + ".so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", so] + objs + [obj, "-L/opt/rocm/lib", "-lrccl", "-lpthread"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        return None, r.stdout.decode()[-400:]
    return so, ""


def run_mutant(tmp, objs, src, name, old, new, model_path):
    if src.count(old) != 1:
        return "FAILED", "the text to mutate occurs %d times in kernels_misc.hip" % src.count(old)
    so, why = build_variant(tmp, name, src.replace(old, new), objs)
    if so is None:
        return "FAILED", why
    env = dict(os.environ, APRIL_ASR_LIB=so, APRIL_LOG_LEVEL="NONE")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "device_decide_mutant_worker.py"), model_path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    except subprocess.TimeoutExpired:
        return "KILLED", "time-out"
    finally:
        for f in (so, os.path.join(tmp, name + ".o"), os.path.join(tmp, name + ".hip")):
            try:
                os.remove(f)
            except OSError:
                pass
    out = r.stdout.decode()
    if r.returncode == 0 and "SURVIVED" in out:
        return "SURVIVED", ""
    return "KILLED", (out.strip().splitlines() or ["exit %d" % r.returncode])[-1][:200]


def run_all(verbose=False, model_path=None, workers=6):
    """returns (killed, survivors, equivalent_killed, build_failures)"""
    src = open(os.path.join(CSRC, "kernels_misc.hip")).read()
    objs = [o for o in sorted(glob.glob(os.path.join(CSRC, "build", "*.o"))) if os.path.basename(o) != "kernels_misc.o"]
    assert objs, "build the library first (csrc/build/*.o)"
    tmp = tempfile.mkdtemp(prefix="april_dmutants_")
    try:
        if model_path is None:
            sys.path.insert(0, ROOT)
            from april_asr_amd import synth_model as SM
            model_path = os.path.join(tmp, "tiny.april")
            SM.write_model(model_path, SM.TINY_DIMS)
        status, why = run_mutant(tmp, objs, src, "identity", "void launch_decide(", "void launch_decide(", model_path)
        assert status == "SURVIVED", "the unmutated device decision fails the fixtures through this harness: %s" % why
        with ThreadPoolExecutor(workers) as ex:
            res = list(ex.map(lambda m: (m[0],) + run_mutant(tmp, objs, src, m[0], m[1], m[2], model_path), MUTANTS))
            eqr = list(ex.map(lambda m: (m[0],) + run_mutant(tmp, objs, src, "eq_" + m[0], m[1], m[2], model_path), EQUIVALENT))
        killed, survivors, failures = [], [], []
        for name, status, why in res:
            if verbose:
                print("%-42s %s %s" % (name, status, why))
            (killed if status == "KILLED" else survivors if status == "SURVIVED" else failures).append((name, why))
        eq_killed = [(n, w) for n, st, w in eqr if st != "SURVIVED"]
        return killed, survivors, eq_killed, failures
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    k, s, e, f = run_all(verbose="-v" in sys.argv)
    print("%d mutants of decide_kernel: %d killed, %d survived, %d failed to build; %d equivalent mutants, %d of them unexpectedly killed"
          % (len(MUTANTS), len(k), len(s), len(f), len(EQUIVALENT), len(e)))
    for name, _ in s:
        print("SURVIVOR:", name)
    for name, why in f:
        print("BUILD FAILURE:", name, why)
    sys.exit(1 if (s or e or f) else 0)
