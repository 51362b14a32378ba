"""Pre-pack a .april model into the cache file the MI355X library loads without parsing ONNX again.

    python -m april_asr_amd.convert model.april model.aprilx          # fp32, MFMA-packed
    python -m april_asr_amd.convert model.april model.aprilx16 --f16  # binary16 matrices for APRIL_PRECISION=f16 (half the size)

The reference parses the three ONNX graphs at every aam_create_model (src/april_model.c:24-107); this library extracts and
packs the weights once (aprilx_model_save_blob / aprilx_model_save_blob_f16) and later loads go through
aprilx_model_load_blob (april_asr_amd.Model.load_blob).  Runs on the host: no GPU is needed to convert.
"""
import argparse
import os
import sys

from . import Model


def convert(src: str, dst: str, f16: bool = False) -> int:
    m = Model.load_host_only(src)          # parse + structural weight extraction + packing, no GPU object
    try:
        m.save_blob(dst, f16=f16)
    finally:
        m.close()
    return os.path.getsize(dst)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m april_asr_amd.convert", description=__doc__.split("\n\n")[0])
    ap.add_argument("model", help="input .april file")
    ap.add_argument("out", help="output cache file")
    ap.add_argument("--f16", action="store_true", help="binary16 weight matrices (for APRIL_PRECISION=f16)")
    a = ap.parse_args(argv)
    try:
        n = convert(a.model, a.out, a.f16)
    except Exception as e:      # the library has logged the reason (which node of which graph it stopped at)
        print("convert: %s" % e, file=sys.stderr)
        return 1
    print("%s: %d bytes (%s)" % (a.out, n, "fp16 matrices" if a.f16 else "fp32"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
