#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/ (SURVEY.md section 8c, G1..G8).

Inputs come from the seeded generator in j2k_amd/synth.py; expected outputs come from a real
libopenjp2 (2.4.0 from /opt/conda, cross-checked against Pillow's bundled 2.5.4) driven through
oracle/opj_replay.c, i.e. through the exact call sequence of the reference's encode entry point
(reference: src/common/j2k_openjpeg_codec.cpp:598-750).  COM marker segments are stripped before
storing/hashing because they embed the library version string.

Run here (needs libopenjp2 + its header, which the GPU box may lack):
    python tests/golden/make_golden.py [--full]
Small cases store the codestream itself (<name>.j2k); full-size BASELINE configs store only
(length, sha256 of the COM-stripped codestream, sha256 of the decoded planes).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from j2k_amd import synth  # noqa: E402
from oracle.oracle import OpjReplay, find_openjpeg_libs, make_params, strip_com  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (width, height, ncomp, prec, seed, dist, params kwargs, input_shift)
# input_shift: G7 feeds 16-bit container samples that CopyBuffer shifts right by (16 - prec).
SMALL = {
    "g1_64x64_grey_1lvl": (64, 64, 1, 8, 12345, "A", dict(numres=2)),
    "g1_64x64_grey_5lvl": (64, 64, 1, 8, 12345, "A", dict(numres=6)),
    "g2_c1_512_grey_53": (512, 512, 1, 8, 12345, "B", dict(numres=6)),
    "g3_300x200_rgb8_53_rct": (300, 200, 3, 8, 12345, "B", dict(numres=6, mct=True)),
    "g4_300x200_rgb16_53_rct_tile128": (300, 200, 3, 16, 34567, "B", dict(numres=6, mct=True, tile=128)),
    "g5_300x200_rgb8_53_ref_literal": (300, 200, 3, 8, 12345, "B", dict(numres=6, mct=False, layers=12, tile=1024)),
    "g6_300x200_rgb8_97_ict": (300, 200, 3, 8, 12345, "B", dict(numres=6, mct=True, reversible=False)),
    "g6_300x200_rgb16_97_ict": (300, 200, 3, 16, 23456, "B", dict(numres=6, mct=True, reversible=False)),
    "g7_300x200_rgb10_53": (300, 200, 3, 10, 45678, "B", dict(numres=6, mct=True)),
    "g9_300x200_rgba8_53_rct": (300, 200, 4, 8, 777, "B", dict(numres=6, mct=True)),
    "g9_97x61_grey12_97_4lvl": (97, 61, 1, 12, 4242, "A", dict(numres=5, reversible=False)),
    "g9_150x130_rgb8_97_tile64": (150, 130, 3, 8, 99, "A", dict(numres=4, mct=True, reversible=False, tile=64)),
}

# G8: main headers of every BASELINE config at reduced size (same coding parameters)
HEADERS = {
    "g8_c1": (512, 512, 1, 8, dict(numres=6)),
    "g8_c2": (256, 256, 3, 8, dict(numres=6, mct=True, reversible=False)),
    "g8_c3": (256, 256, 3, 16, dict(numres=7, mct=True, reversible=False)),
    "g8_c3_5lvl": (256, 256, 3, 16, dict(numres=6, mct=True, reversible=False)),
    "g8_c4": (512, 512, 3, 16, dict(numres=6, mct=True, tile=128)),
    "g8_c5": (256, 135, 3, 10, dict(numres=6, mct=True, reversible=False)),
}

FULL = {
    "c1_512_grey_53": (512, 512, 1, 8, 12345, "A", dict(numres=6)),
    "c2_4096_rgb8_97": (4096, 4096, 3, 8, 12345, "A", dict(numres=6, mct=True, reversible=False)),
    "c3_8192_rgb16_97_6lvl": (8192, 8192, 3, 16, 23456, "A", dict(numres=7, mct=True, reversible=False)),
    "c3_8192_rgb16_97_5lvl": (8192, 8192, 3, 16, 23456, "A", dict(numres=6, mct=True, reversible=False)),
    "c5_frame0_4096x2160_rgb10_97": (4096, 2160, 3, 10, 45678, "A", dict(numres=6, mct=True, reversible=False)),
    "c4_tile_2048_rgb16_53": (2048, 2048, 3, 16, 34567, "A", dict(numres=6, mct=True)),
}


def sha(b):
    return hashlib.sha256(b).hexdigest()


def psnr(a, b, prec):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10 * np.log10(((1 << prec) - 1) ** 2 / mse)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also (re)generate the full-size hashes")
    args = ap.parse_args()
    libs = find_openjpeg_libs()
    reps = [OpjReplay(l) for l in libs[:1]]
    rep = reps[0]
    others = [OpjReplay(l) for l in libs[1:]]
    meta_path = os.path.join(HERE, "golden.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    meta["_generator"] = dict(library=rep.version, libpath=os.path.basename(rep.libpath),
                              cross_checked=[o.version for o in others],
                              note="COM segments stripped before hashing/storing")

    for name, (w, h, nc, prec, seed, dist, kw) in SMALL.items():
        pl = synth.planes(w, h, nc, prec, seed, dist)
        p = make_params(w, h, nc, prec, **kw)
        cs = strip_com(rep.encode(pl, p))
        for o in others:
            assert strip_com(o.encode(pl, p)) == cs, (name, o.version)
        dec = rep.decode(cs)
        with open(os.path.join(HERE, name + ".j2k"), "wb") as f:
            f.write(cs)
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist=dist, params=kw,
                          length=len(cs), sha256=sha(cs), decoded_sha256=sha(dec.tobytes()),
                          psnr=None if kw.get("reversible", True) else round(psnr(dec, pl, prec), 4))
        if kw.get("reversible", True):
            assert np.array_equal(dec, pl), name
        print(name, len(cs), meta[name]["psnr"])

    for name, (w, h, nc, prec, kw) in HEADERS.items():
        pl = synth.planes(w, h, nc, prec, 1, "B")
        p = make_params(w, h, nc, prec, **kw)
        cs = strip_com(rep.encode(pl, p))
        hdr = cs[:cs.index(b"\xff\x90")]
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, params=kw, main_header_hex=hdr.hex())
        print(name, len(hdr))

    if args.full:
        for name, (w, h, nc, prec, seed, dist, kw) in FULL.items():
            t0 = time.time()
            pl = synth.planes(w, h, nc, prec, seed, dist)
            p = make_params(w, h, nc, prec, **kw)
            cs = strip_com(rep.encode(pl, p, threads=8))
            t1 = time.time()
            dec = rep.decode(cs, threads=8)
            meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist=dist, params=kw,
                              length=len(cs), sha256=sha(cs), decoded_sha256=sha(dec.tobytes()),
                              psnr=None if kw.get("reversible", True) else round(psnr(dec, pl, prec), 4))
            print(name, len(cs), meta[name]["psnr"], f"enc {t1 - t0:.1f}s total {time.time() - t0:.1f}s")
            del pl, dec

    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
