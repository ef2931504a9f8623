"""Adds hash-only goldens of two more full-size frames to tests/golden/cases.json (run here, where /root/reference is built
as oracle/_ref/libcharls_ref.so; make_golden.py made the rest of the file and is not run again by this):

  full_tulips_tiled_0 / _1   the reference's natural test image tiled 8 x 8 to 4096 x 4096 (frames 0 and 1 of bench.py's
                             `tulips` data: its decoder figures for "a natural image" are measured on these)
  full_noise                 4096 x 4096 samples uniform in 0..255 (the encoder's worst case)

Only sizes and SHA-256 of the reference's output are kept (the frames are 16.8 MB each).
Usage: python tests/golden/make_golden_full_frames.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
from charls_amd.capi import CharLSLibrary  # noqa: E402

REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libcharls_ref.so")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    ref = CharLSLibrary(REF_LIB)
    path = os.path.join(HERE, "cases.json")
    with open(path) as f:
        cases = json.load(f)
    new = [dict(name="full_tulips_tiled_0", kind="tulips_tiled", seed=0), dict(name="full_tulips_tiled_1", kind="tulips_tiled", seed=1),
           dict(name="full_noise", kind="noise", seed=2)]
    cases = [c for c in cases if c["name"] not in {n["name"] for n in new}]
    for n in new:
        rec = dict(name=n["name"], width=4096, height=4096, bits_per_sample=8, component_count=1, interleave_mode=0, near_lossless=0,
                   color_transformation=0, preset=None, kind=n["kind"], seed=n["seed"], encoding_options=0)
        img = common.case_input(rec)
        rec["input_sha256"] = sha(img.tobytes())
        # (full-range noise codes to more than the reference's estimated destination size: give it room)
        jls = ref.encode(img, width=4096, height=4096, destination_size=2 * img.size + 1024)
        rec.update(errc=0, jls_size=len(jls), jls_sha256=sha(jls))
        _, px = ref.decode(jls)
        rec["decoded_sha256"] = sha(px.tobytes())
        cases.append(rec)
        print(rec["name"], rec["jls_size"])
    with open(path, "w") as f:
        json.dump(cases, f, indent=1)


if __name__ == "__main__":
    main()
