#!/usr/bin/env python3
"""Generates tests/golden/* by running THE REFERENCE (oracle/_ref/libcharls_ref.so, built from /root/reference by
`make -C oracle ref`) in the build container.  The outputs are data only:

* refdata/   -- data files the reference's own tests hold (test/data/**): ISO conformance images/streams, restart-marker
                streams, corrupt streams.  Copied verbatim (land10 truncated to its header, which is all its test reads).
* small/     -- .jls byte strings the reference produced for seeded synthetic inputs (charls_amd/synth.py) at small sizes.
* cases.json -- one record per case: parameters, input recipe, size + SHA-256 of the reference's .jls, and for the
                BASELINE.json full-size configurations the hash only.

Run:  python tests/golden/make_golden.py          (needs /root/reference; not run on the GPU box)
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from charls_amd import synth  # noqa: E402
from charls_amd.capi import CharLSLibrary, JpegLSError  # noqa: E402

REF_DATA = "/root/reference/test/data"
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libcharls_ref.so")

COPY = [
    "conformance/test8.ppm", "conformance/test8bs2.pgm", "conformance/test16.pgm",
    "conformance/t8c0e0.jls", "conformance/t8c1e0.jls", "conformance/t8c2e0.jls",
    "conformance/t8c0e3.jls", "conformance/t8c1e3.jls", "conformance/t8c2e3.jls",
    "conformance/t8nde0.jls", "conformance/t8nde3.jls", "conformance/t16e0.jls", "conformance/t16e3.jls",
    "tulips-gray-8bit-512-512.pgm", "tulips-gray-8bit-512-512-hp-encoder.jls",
    "test8_ilv_none_rm_7.jls", "test8_ilv_line_rm_7.jls", "test8_ilv_sample_rm_7.jls", "test8_ilv_sample_rm_300.jls",
    "test16_rm_5.jls", "8bit-monochrome-2x2.jls", "2bit_parrot_150x200.pgm", "4bit-monochrome.pgm",
    "16-bit-640-480-many-dots.pgm", "banny.ppm", "banny-hp1.jls", "banny-hp2.jls", "banny-hp3.jls",
    "fuzzy-input-bad-run-mode-golomb-code.jls", "fuzzy-input-no-valid-bits-at-the-end.jls",
    "fuzzy_input_golomb_16.jls", "no_start_byte_after_encoded_scan.jls",
]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    ref = CharLSLibrary(REF_LIB)
    os.makedirs(os.path.join(HERE, "refdata"), exist_ok=True)
    os.makedirs(os.path.join(HERE, "small"), exist_ok=True)
    for rel in COPY:
        shutil.copyfile(os.path.join(REF_DATA, rel), os.path.join(HERE, "refdata", os.path.basename(rel)))
    with open(os.path.join(REF_DATA, "land10-10bit-rgb-hp3-invalid.jls"), "rb") as f:
        head = f.read(4096)
    with open(os.path.join(HERE, "refdata", "land10-10bit-rgb-hp3-invalid.head.jls"), "wb") as f:
        f.write(head)

    cases = []

    def add(name, w, h, bits=8, comps=1, ilv=0, near=0, ct=0, preset=None, kind="gradient", seed=1, options=0,
            keep=True):
        img = synth.frame_numpy(w, h, seed=seed, bits=bits, components=comps, kind=kind, interleaved=(ilv != 0))
        rec = dict(name=name, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=ilv,
                   near_lossless=near, color_transformation=ct, preset=preset, kind=kind, seed=seed,
                   encoding_options=options, input_sha256=sha(img.tobytes()))
        try:
            jls = ref.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, near_lossless=near,
                             interleave_mode=ilv, color_transformation=ct, preset=preset, encoding_options=options)
            rec.update(errc=0, jls_size=len(jls), jls_sha256=sha(jls))
            _, px = ref.decode(jls)
            rec["decoded_sha256"] = sha(px.tobytes())
            if keep:
                rec["file"] = f"small/{name}.jls"
                with open(os.path.join(HERE, rec["file"]), "wb") as f:
                    f.write(jls)
        except JpegLSError as e:
            rec.update(errc=e.errc)
        cases.append(rec)
        print(name, rec.get("jls_size"), rec["errc"])

    # small, bytes committed -------------------------------------------------------------------------------------
    for bits in (2, 4, 7, 8, 10, 12, 15, 16):
        add(f"gray{bits}_64x48", 64, 48, bits=bits, kind="mixed", seed=bits)
        nr = min(3, ((1 << bits) - 1) // 2)
        add(f"gray{bits}_near{nr}_64x48", 64, 48, bits=bits, near=nr, kind="mixed", seed=bits + 20)
    add("gray8_w1", 1, 37, seed=3)
    add("gray8_h1", 53, 1, seed=4)
    add("gray8_1x1", 1, 1, seed=5)
    add("gray8_zero_512", 512, 512, kind="zero")          # 99-byte KAT (test/jpegls_encoder_test.cpp:1807)
    add("gray8_noise_96", 96, 96, kind="noise", seed=6)   # escape codes / FF stuffing
    add("gray16_noise_64", 64, 64, bits=16, kind="noise", seed=7)  # LIMIT=64 escape codes
    add("gray8_hard_128", 128, 128, kind="hard", seed=8)
    add("gray8_runs_300x40", 300, 40, kind="mixed", seed=9)
    add("gray8_custom_pc", 96, 64, preset=(255, 9, 9, 9, 31), kind="mixed", seed=10)   # t8nde* style parameters
    add("gray8_maxval100", 64, 64, preset=(100, 0, 0, 0, 0), kind="zero", seed=11)
    add("gray12_pc_jai", 64, 64, bits=12, options=4, seed=12)
    add("gray16_pc_jai", 64, 64, bits=16, options=4, seed=12)
    add("gray8_even_version", 33, 17, options=3, seed=13)
    for ilv in (0, 1, 2):
        add(f"rgb8_ilv{ilv}_80x60", 80, 60, comps=3, ilv=ilv, kind="mixed", seed=30 + ilv)
        add(f"rgb8_ilv{ilv}_near2_80x60", 80, 60, comps=3, ilv=ilv, near=2, kind="mixed", seed=40 + ilv)
        add(f"rgb16_ilv{ilv}_40x30", 40, 30, bits=16, comps=3, ilv=ilv, kind="mixed", seed=50 + ilv)
    for ct in (1, 2, 3):
        add(f"rgb8_line_hp{ct}", 80, 60, comps=3, ilv=1, ct=ct, kind="mixed", seed=60 + ct)
        add(f"rgb8_sample_hp{ct}", 80, 60, comps=3, ilv=2, ct=ct, kind="mixed", seed=70 + ct)
        add(f"rgb16_sample_hp{ct}", 40, 30, bits=16, comps=3, ilv=2, ct=ct, kind="mixed", seed=80 + ct)
    add("c2_ilv1", 64, 64, comps=2, ilv=1, kind="mixed", seed=90)
    add("c2_ilv2", 64, 64, comps=2, ilv=2, kind="mixed", seed=91)
    add("c4_ilv1", 64, 64, comps=4, ilv=1, kind="mixed", seed=92)
    add("c4_ilv2", 64, 64, comps=4, ilv=2, near=1, kind="mixed", seed=93)
    add("c5_ilv0", 32, 32, comps=5, ilv=0, kind="mixed", seed=94)
    add("config5_literal_rejected", 64, 64, comps=3, ilv=2, ct=1, near=2, seed=95)    # errc 109
    # tiny cases: every coding mode once, cheap enough for the thread-per-lane kernel emulation of the CPU suite
    for ilv in (0, 1, 2):
        add(f"tiny_rgb8_ilv{ilv}", 24, 16, comps=3, ilv=ilv, kind="mixed", seed=110 + ilv)
    add("tiny_rgb8_ilv2_near2", 24, 16, comps=3, ilv=2, near=2, kind="mixed", seed=114)
    add("tiny_rgb8_ilv1_near1", 24, 16, comps=3, ilv=1, near=1, kind="mixed", seed=115)
    add("tiny_rgb16_line_hp3", 24, 16, bits=16, comps=3, ilv=1, ct=3, kind="mixed", seed=116)
    add("tiny_rgb8_sample_hp1", 24, 16, comps=3, ilv=2, ct=1, kind="mixed", seed=117)
    add("tiny_c4_ilv2_near1", 16, 16, comps=4, ilv=2, near=1, kind="mixed", seed=118)
    add("tiny_c2_ilv1", 20, 12, comps=2, ilv=1, kind="mixed", seed=119)
    add("tiny_gray12", 40, 24, bits=12, kind="mixed", seed=120)
    add("tiny_gray16_noise", 24, 24, bits=16, kind="noise", seed=121)
    add("tiny_gray2", 40, 24, bits=2, kind="mixed", seed=122)
    add("tiny_gray8_near3", 40, 24, near=3, kind="mixed", seed=123)
    add("tiny_gray8_noise", 32, 32, kind="noise", seed=124)
    # BASELINE.json configs at crop sizes (bytes) and full size (hash only) -----------------------------------------
    add("cfg1_512", 512, 512, seed=1, keep=False)
    add("cfg2_crop256", 256, 256, seed=2)
    add("cfg3_crop256", 256, 256, bits=16, seed=4)
    add("cfg3b_crop256_12bit", 256, 256, bits=12, seed=4)
    add("cfg5a_crop128", 128, 128, comps=3, ilv=2, ct=1, seed=5)
    add("cfg5b_crop128", 128, 128, comps=3, ilv=2, near=2, seed=5)
    if "--no-full" not in sys.argv:
        add("cfg2_full", 4096, 4096, seed=2, keep=False)
        add("cfg2_full_mixed", 4096, 4096, seed=2, kind="mixed", keep=False)
        add("cfg3_full", 4096, 4096, bits=16, seed=4, keep=False)
        add("cfg3b_full_12bit", 4096, 4096, bits=12, seed=4, keep=False)
        for f in range(4):
            add(f"cfg4_frame{f}", 2048, 2048, seed=100 + f, keep=False)
        add("cfg5a_full", 4096, 4096, comps=3, ilv=2, ct=1, seed=5, keep=False)
        add("cfg5b_full", 4096, 4096, comps=3, ilv=2, near=2, seed=5, keep=False)

    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print(len(cases), "cases written")


if __name__ == "__main__":
    main()
