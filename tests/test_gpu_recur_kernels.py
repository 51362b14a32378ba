"""The layer GEMMs at <= 16 rows (the recurrent pair and the block stages of a long feed, every layer GEMM of a few streaming
sessions) run as weight streams (csrc/kernels_recur.hip: six forms) and as general GEMM tiles above; both must produce the same
bits.  Two processes run the same sessions with the stream kernels off and on; every logit and every callback must be identical.
The four models cover the kernels' shapes: 1 ... 24 k blocks per wave, kz = 1, 2 and 8, the block ring wrapping (larger encoder)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(path, mode, nsess, secs, ksplit=0):
    env = dict(os.environ, APRIL_RECUR_KERNELS=str(mode), APRIL_MAX_SESSIONS="64", APRIL_MAX_BATCH="1024", APRIL_RECUR_KSPLIT=str(ksplit))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "recur_worker.py"), path, str(nsess), str(secs)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("DIGEST")][-1].split()
    return line[1], int(line[2]), int(line[3]), int(line[4])


@pytest.mark.parametrize("which,nsess,secs", [("tiny", 1, 4.0), ("tiny", 7, 3.0), ("medium", 3, 3.0), ("v0", 2, 3.0), ("large", 1, 2.0)])
def test_stream_kernels_equal_general_kernels(built, tiny_model, medium_model, v0_model, large_model, which, nsess, secs):
    path = {"tiny": tiny_model, "medium": medium_model, "v0": v0_model, "large": large_model}[which]["path"]
    a = run(path, 0, nsess, secs)
    b = run(path, 1, nsess, secs)
    assert a[1] == b[1] and a[1] > 0 and a[2] == b[2] and a[2] > 0, (a, b)       # same chunks, some of them layer-major
    assert a[3] == 0 and b[3] == 0
    assert a[0] == b[0], "stream kernels and general kernels differ"


@pytest.mark.parametrize("which,nsess,secs,cut", [("v0", 1, 3.0, 1), ("v0", 2, 3.0, 2), ("tiny", 5, 3.0, 1), ("medium", 3, 2.0, 1)])
def test_k_cut_stream_kernels_equal_whole_k(built, tiny_model, medium_model, v0_model, which, nsess, secs, cut):
    """The measurement form APRIL_RECUR_KSPLIT (csrc/kernels_recur.hip recur_ksplit: the projection / FFN-down stream kernels with K cut
    across workgroups and an in-launch hand-over to the last arriver) computes the same slab sums and the same tree: every logit and
    callback equals the whole-K form's, over the layer-major steps of a long feed and the streamed feeds behind it."""
    path = {"tiny": tiny_model, "medium": medium_model, "v0": v0_model}[which]["path"]
    a = run(path, 1, nsess, secs)
    b = run(path, 1, nsess, secs, ksplit=cut)
    assert a[1] == b[1] and a[1] > 0 and a[3] == 0 and b[3] == 0, (a, b)
    assert a[0] == b[0], "K-cut stream kernels differ from the whole-K form"
