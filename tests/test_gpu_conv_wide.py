"""The conv front end has two forms (csrc/kernels_misc.hip): conv12_kernel (channel groups of the second conv as separate workgroups:
what fills the chip at a few chunks) and conv12_wide_kernel (one workgroup per chunk, all channels, packed multiplies and adds: from
APRIL_CONV_WIDE_MIN = 48 chunks per launch).  Every output is the same chain of products in both (input channel, kernel row, kernel
column, bias last), so whole sessions must agree bit for bit whichever form a launch takes.  Two processes run the same sessions with
the wide form off (-1) and forced on for every launch (0); every logit and every callback must be identical.
Reference path: the conv-embed nodes of the encoder graph behind src/april_session.c:431-454."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(path, wide_min, nsess, secs):
    env = dict(os.environ, APRIL_CONV_WIDE_MIN=str(wide_min), APRIL_MAX_SESSIONS="64", APRIL_MAX_BATCH="1024")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "recur_worker.py"), path, str(nsess), str(secs)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("DIGEST")][-1].split()
    return line[1], int(line[2]), int(line[3]), int(line[4])


@pytest.mark.parametrize("which,nsess,secs", [("v0", 3, 3.0), ("tiny", 7, 3.0), ("medium", 3, 2.0), ("large", 2, 2.0)])
def test_wide_conv_front_end_equals_grouped_form(built, tiny_model, medium_model, v0_model, large_model, which, nsess, secs):
    path = {"tiny": tiny_model, "medium": medium_model, "v0": v0_model, "large": large_model}[which]["path"]
    a = run(path, -1, nsess, secs)
    b = run(path, 0, nsess, secs)
    assert a[1] == b[1] and a[1] > 0 and a[3] == 0 and b[3] == 0, (a, b)
    assert a[0] == b[0], "the two forms of the conv front end differ"

