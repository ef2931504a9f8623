"""Shared helpers for the tests: golden manifest, PNM reader, input recipes."""
import hashlib
import json
import os

import numpy as np

from charls_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return json.load(f)


def tulips_tiled(width, height, roll):
    """The reference's natural test image (refdata/tulips-gray-8bit-512-512.pgm) tiled to width x height and rolled by
    (17 roll, 29 roll) -- frame `roll` of bench.py's `tulips` data (bench.py: data_frames)."""
    tile, _ = read_pnm("tulips-gray-8bit-512-512.pgm")
    big = np.tile(tile, (height // tile.shape[0], width // tile.shape[1]))
    return np.ascontiguousarray(np.roll(big, shift=(17 * roll, 29 * roll), axis=(0, 1)))


def case_input(c):
    if c["kind"] == "tulips_tiled":
        return tulips_tiled(c["width"], c["height"], c["seed"])
    return synth.frame_numpy(c["width"], c["height"], seed=c["seed"], bits=c["bits_per_sample"],
                             components=c["component_count"], kind=c["kind"],
                             interleaved=(c["interleave_mode"] != 0))


def case_kwargs(c):
    return dict(width=c["width"], height=c["height"], bits_per_sample=c["bits_per_sample"],
                component_count=c["component_count"], near_lossless=c["near_lossless"],
                interleave_mode=c["interleave_mode"], color_transformation=c["color_transformation"],
                preset=tuple(c["preset"]) if c["preset"] else None, encoding_options=c["encoding_options"])


def refdata(name) -> bytes:
    with open(os.path.join(GOLDEN, "refdata", name), "rb") as f:
        return f.read()


def read_pnm(name):
    """Minimal binary PGM/PPM reader (P5/P6). 16-bit samples are big-endian in the file, returned native.
    Returns (array (H,W) or (H,W,3), maxval)."""
    data = refdata(name)
    tokens, pos = [], 0
    while len(tokens) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        start = pos
        while not data[pos:pos + 1].isspace():
            pos += 1
        tokens.append(data[start:pos])
    pos += 2 if data[pos:pos + 2] == b"\r\n" else 1
    magic, w, h, maxval = tokens[0], int(tokens[1]), int(tokens[2]), int(tokens[3])
    comps = 3 if magic == b"P6" else 1
    if maxval > 255:
        a = np.frombuffer(data, dtype=">u2", count=w * h * comps, offset=pos).astype(np.uint16)
    else:
        a = np.frombuffer(data, dtype=np.uint8, count=w * h * comps, offset=pos).copy()
    return (a.reshape(h, w, 3) if comps == 3 else a.reshape(h, w)), maxval


def planar(img):
    """(H,W,C) -> (C,H,W) contiguous: the ILV_NONE user layout (src/charls_jpegls_encoder.cpp:209-224)."""
    return np.ascontiguousarray(np.moveaxis(img, 2, 0))
