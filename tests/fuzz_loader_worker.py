"""Worker of tests/test_loader.py::test_corrupted_models_never_crash: loads `count` corrupted copies of a model with the
product's host-only loader and with the oracle's loader; prints how many each accepted.  A crash kills this process."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import april_asr_amd as A                    # noqa: E402
from oracle import orc_py as O               # noqa: E402

src, tmp, seed, count = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
data = np.frombuffer(open(src, "rb").read(), np.uint8)
rng = np.random.RandomState(seed)
ok_p = ok_o = 0
for it in range(count):
    d = data.copy()
    kind = it % 4
    if kind == 0:                             # a few random byte flips anywhere
        for _ in range(int(rng.randint(1, 8))):
            d[int(rng.randint(d.size))] = int(rng.randint(256))
    elif kind == 1:                           # flips in the first 4 KiB (container header, params, ONNX graph preamble)
        for _ in range(int(rng.randint(1, 16))):
            d[int(rng.randint(min(4096, d.size)))] = int(rng.randint(256))
    elif kind == 2:                           # truncation
        d = d[: int(rng.randint(1, d.size))]
    else:                                     # a run of 0xff (huge varints / lengths)
        o = int(rng.randint(d.size - 16)); d[o:o + int(rng.randint(1, 16))] = 255
    with open(tmp, "wb") as f:
        f.write(d.tobytes())
    try:
        m = A.Model.load_host_only(tmp); m.close(); ok_p += 1
    except Exception:
        pass
    p = O.lib().orc_model_load(tmp.encode())
    if p:
        O.lib().orc_model_free(p); ok_o += 1
print("accepted product=%d oracle=%d of %d" % (ok_p, ok_o, count))
