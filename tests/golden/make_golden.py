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

# JP2 wrapper: whole files from libopenjp2's own JP2 writer (OPJ_CODEC_JP2), COM segment
# kept (the test passes the same comment text).  name -> (w, h, ncomp, prec, seed, params kwargs,
# OPJ_COLOR_SPACE, ICC profile length or 0, alpha channel or -1).  Generated with 2.5.x (2.4.0 cannot write
# ICC profiles and writes EnumCS 0 for e-YCC/CMYK); the other cases are cross-checked against 2.4.0.
JP2 = {
    "j1_64x48_rgb8_srgb": (64, 48, 3, 8, 101, dict(numres=3, mct=True), 1, 0, -1),
    "j2_64x48_grey8": (64, 48, 1, 8, 102, dict(numres=3), 2, 0, -1),
    "j3_64x48_rgba8_srgb_alpha": (64, 48, 4, 8, 103, dict(numres=3, mct=True), 1, 0, 3),
    "j4_64x48_rgb16_icc": (64, 48, 3, 16, 104, dict(numres=3, mct=True), 1, 560, -1),
    "j5_64x48_rgb8_sycc_97": (64, 48, 3, 8, 105, dict(numres=3, reversible=False), 3, 0, -1),
    "j6_40x30_greya8": (40, 30, 2, 8, 106, dict(numres=2), 2, 0, 1),
    "j7_40x30_cmyk8": (40, 30, 4, 8, 107, dict(numres=2), 5, 0, -1),
    "j8_40x30_rgb8_unspecified": (40, 30, 3, 8, 108, dict(numres=2), 0, 0, -1),
    "j9_40x30_rgba16_icc_alpha": (40, 30, 4, 16, 109, dict(numres=2, mct=True), 1, 200, 3),
}


# Rate control: whole codestreams from libopenjp2 with tcp_rates / cp_disto_alloc, COM kept.
# name -> (w, h, ncomp, prec, seed, params kwargs, compression ratio per layer)
RATES = {
    "r1_128_grey8_53_r20": (128, 128, 1, 8, 7, dict(numres=3), [20.0]),
    "r2_128_grey8_97_r20": (128, 128, 1, 8, 7, dict(numres=3, reversible=False), [20.0]),
    "r3_300x200_rgb8_97_ict_r40_20_10": (300, 200, 3, 8, 7, dict(numres=6, mct=True, reversible=False), [40.0, 20.0, 10.0]),
    "r4_300x200_rgb8_53_rct_r30_10_0": (300, 200, 3, 8, 7, dict(numres=6, mct=True), [30.0, 10.0, 0.0]),
    "r5_300x200_rgb16_97_ict_tile128_r50_25": (300, 200, 3, 16, 7, dict(numres=6, mct=True, reversible=False, tile=128), [50.0, 25.0]),
    "r6_64_grey8_53_r5_2_1": (64, 64, 1, 8, 7, dict(numres=2), [5.0, 2.0, 1.0]),
    "r7_97x61_grey12_97_r12_6_3": (97, 61, 1, 12, 7, dict(numres=5, reversible=False), [12.0, 6.0, 3.0]),
    "r8_300x200_rgba8_53_rct_7layers": (300, 200, 4, 8, 7, dict(numres=6, mct=True), [100.0, 50.0, 25.0, 12.0, 6.0, 3.0, 0.0]),
    "r9_200x150_rgb10_97_cblk32_r25_8": (200, 150, 3, 10, 9, dict(numres=4, mct=True, reversible=False, cblk=(32, 32)), [25.0, 8.0]),
}


# Progression orders (settings.order = j2k::Order = OPJ_PROG_ORDER): name -> (w, h, nc, prec, seed, kwargs, order, rates or None)
ORDERS = {
    "o1_300x200_rgb8_53_rct_3layers_rlcp": (300, 200, 3, 8, 31, dict(numres=6, mct=True, layers=3), 1, None),
    "o2_300x200_rgb8_97_ict_rpcl_r40_20_8": (300, 200, 3, 8, 32, dict(numres=5, mct=True, reversible=False), 2, [40.0, 20.0, 8.0]),
    "o3_200x300_rgba16_53_tile128_2layers_pcrl": (200, 300, 4, 16, 33, dict(numres=4, tile=128, layers=2), 3, None),
    "o4_97x61_grey12_97_cprl_r12_3": (97, 61, 1, 12, 34, dict(numres=5, reversible=False), 4, [12.0, 3.0]),
}

# Fixed quality: PSNR target (dB) per layer, cp_fixed_quality / tcp_distoratio.  Same tuple layout as RATES.
QUALITY = {
    "q1_128_grey8_53_q35": (128, 128, 1, 8, 7, dict(numres=3), [35.0]),
    "q2_300x200_rgb8_97_ict_q30_38_45": (300, 200, 3, 8, 7, dict(numres=6, mct=True, reversible=False), [30.0, 38.0, 45.0]),
    "q3_300x200_rgb8_53_rct_q32_40_0": (300, 200, 3, 8, 7, dict(numres=6, mct=True), [32.0, 40.0, 0.0]),
    "q4_300x200_rgb16_97_ict_tile128_q40_60": (300, 200, 3, 16, 7, dict(numres=6, mct=True, reversible=False, tile=128), [40.0, 60.0]),
    "q5_239x97_rgba16_97_q43_47": (239, 97, 4, 16, 11, dict(numres=2, mct=True, reversible=False), [42.7, 46.6]),
}


# Files the reference can READ (OpenJPEGCodec::GetFileInfo / ::ReadFile pass anything libopenjp2 decodes) but that no
# WriteFile of the plug-in produces: sub-sampled / signed components, user-defined precincts, code-block styles, origin
# offsets, SOP/EPH.  name -> (w, h, ncomp, prec, seed, encode_ext kwargs).  Stored with the per-component decoded hashes
# (full size and cp_reduce 1), written by libopenjp2's general encoder (oracle/opj_replay.c: opjr_encode_ext).
EXT = {
    "u1_300x200_ycc420_8_53": (300, 200, 3, 8, 61, dict(sub=[(1, 1), (2, 2), (2, 2)], numres=5)),
    "u2_301x199_ycc422_10_97_tile128": (301, 199, 3, 10, 62, dict(sub=[(1, 1), (2, 1), (2, 1)], numres=4, reversible=False, tile=(128, 128))),
    "u3_300x200_rgb8_53_precincts_rpcl": (300, 200, 3, 8, 63, dict(numres=5, mct=True, precincts=[(128, 128), (64, 64)], prog=2)),
    "u4_300x200_rgb8_97_precincts_cprl_2layers": (300, 200, 3, 8, 64, dict(numres=4, mct=True, reversible=False, precincts=[(64, 64), (32, 32)],
                                                                          prog=4, rates=[30.0, 8.0], cblk=(32, 32))),
    "u5_97x61_grey12_signed_53": (97, 61, 1, 12, 65, dict(numres=4, sgnd=True)),
    "u6_200x150_rgb8_53_offset": (200, 150, 3, 8, 66, dict(numres=4, mct=True, x0=37, y0=21, tile=(96, 96), tile_origin=(5, 3))),
    "u7_128_grey8_53_bypass_termall": (128, 128, 1, 8, 67, dict(numres=3, mode=1 | 4)),
    "u8_300x200_rgb8_53_sop_eph_pcrl_precincts": (300, 200, 3, 8, 68, dict(numres=5, precincts=[(128, 64)], prog=3, sop=True, eph=True)),
    # user-defined precincts as the ENCODER's goldens too (plain unsigned components, no SOP/EPH): all five progressions
    "p1_512x270_rgb12_97_cprl_dci_precincts_r10": (512, 270, 3, 12, 71, dict(numres=6, mct=True, reversible=False, precincts=[(256, 256)] * 5 + [(128, 128)],
                                                                             prog=4, rates=[10.0], cblk=(32, 32))),
    "p2_300x200_rgb8_53_pcrl_precincts64_3layers": (300, 200, 3, 8, 72, dict(numres=5, mct=True, precincts=[(64, 64)], prog=3, layers=3)),
    "p3_301x203_grey16_97_rlcp_precincts": (301, 203, 1, 16, 73, dict(numres=4, reversible=False, precincts=[(128, 64), (64, 64), (32, 16)], prog=1)),
    "p4_257x129_rgba8_53_lrcp_precincts_tile128": (257, 129, 4, 8, 74, dict(numres=3, mct=True, precincts=[(32, 32)], tile=(128, 128))),
    # code-block styles (COD SPcod: 1 bypass, 2 reset, 4 termall, 8 vcausal, 16 pterm, 32 segsym), one by one and together
    "s1_300x200_rgb16_53_bypass": (300, 200, 3, 16, 81, dict(numres=4, mct=True, mode=1)),
    "s2_300x200_rgb8_97_reset_vcausal_segsym": (300, 200, 3, 8, 82, dict(numres=4, mct=True, reversible=False, mode=2 | 8 | 32)),
    "s3_200x150_grey12_53_termall_pterm": (200, 150, 1, 12, 83, dict(numres=3, mode=4 | 16)),
    "s4_300x200_rgb16_97_all_styles_2layers_tile128": (300, 200, 3, 16, 84, dict(numres=4, mct=True, reversible=False, mode=63, rates=[40.0, 4.0],
                                                                               tile=(128, 128))),
    "s5_257x131_rgb10_53_bypass_termall_cblk32_rpcl": (257, 131, 3, 10, 85, dict(numres=3, mct=True, mode=1 | 4, cblk=(32, 32), prog=2, precincts=[(64, 64)])),
    # libopenjp2's digital cinema profiles: CPRL, precincts 128 / 256, 32 x 32 blocks, a tile-part per component, TLM; 4K: a
    # progression order change (POC) that puts the 2K resolutions first
    "d1_512x270_rgb12_cinema2k": (512, 270, 3, 12, 86, dict(numres=6, mct=True, reversible=False, rsiz=3, versions_differ=True)),
    "d2_1024x540_rgb12_cinema4k_poc": (1024, 540, 3, 12, 87, dict(numres=7, mct=True, reversible=False, rsiz=4, versions_differ=True)),
    # region of interest by MAXSHIFT (RGN): libopenjp2 lifts a whole component
    "r1_200x150_rgb8_53_roi_comp1_shift5": (200, 150, 3, 8, 91, dict(numres=4, roi=(1, 5))),
    "r2_300x200_rgb10_97_ict_roi_comp0_shift7_r12": (300, 200, 3, 10, 92, dict(numres=5, mct=True, reversible=False, roi=(0, 7), rates=[12.0])),
    # more than four components: the reference's reader takes the first four (src/common/j2k_openjpeg.cpp:278, :530)
    "m1_90x70_6comp8_53_rct": (90, 70, 6, 8, 93, dict(numres=4, mct=True)),
    "m2_150x130_5comp12_97_tile64_2layers": (150, 130, 5, 12, 94, dict(numres=3, reversible=False, tile=(64, 64), rates=[20.0, 5.0])),
    "ua_200x150_grey8_53_cblk128x32": (200, 150, 1, 8, 70, dict(numres=3, cblk=(128, 32))),  # legal (xcb + ycb <= 12), beyond the 64 x 64 of this decoder
    "u9_256_rgb8_53_precincts_lrcp_tile100": (256, 256, 3, 8, 69, dict(numres=4, mct=True, precincts=[(64, 64), (64, 64), (32, 32), (16, 16)], tile=(100, 100))),
}


def ext_entries(meta, reps):
    rep = reps[0]
    for name, (w, h, nc, prec, seed, kw) in EXT.items():
        kw = dict(kw)
        pl = synth.planes(w, h, nc, prec, seed, "B")
        if kw.get("sgnd"):
            pl = pl - (1 << (prec - 1))
        x0, y0 = kw.get("x0", 0), kw.get("y0", 0)
        sub = kw.get("sub", [(1, 1)] * nc)
        comps = []
        for c in range(nc):
            dx, dy = sub[c]
            # component sample (i, j) sits at reference-grid position (i*dx, j*dy): those of the image area [x0, x0+w) x [y0, y0+h)
            cx0, cx1 = -(-x0 // dx), -(-(x0 + w) // dx)
            cy0, cy1 = -(-y0 // dy), -(-(y0 + h) // dy)
            full = synth.planes(x0 + w, y0 + h, nc, prec, seed, "B")[c] - ((1 << (prec - 1)) if kw.get("sgnd") else 0)
            comps.append(np.ascontiguousarray(full[cy0 * dy:(cy1 - 1) * dy + 1:dy, cx0 * dx:(cx1 - 1) * dx + 1:dx]))
        enc_kw = {k: v for k, v in kw.items() if k != "versions_differ"}
        enc_kw.update(x1=x0 + w, y1=y0 + h, prec=prec)
        keep_com = "rates" in kw  # a byte budget also pays for the COM segment: those files keep it (the test passes the same text)
        raws = [r.encode_ext(comps, **enc_kw) for r in reps]
        outs = [x.replace(r.comment.encode(), reps[0].comment.encode()) if keep_com else strip_com(x) for x, r in zip(raws, reps)]
        if not kw.get("versions_differ"):  # (the cinema profiles: 2.5 writes other rate limits / markers than 2.4 -- the file is 2.4.0's, both decode it alike)
            assert all(o == outs[0] for o in outs[1:]), name
        cs = outs[0]
        dec = {}
        for red in (0, 1):
            ds = [r.decode_comps(cs, red) for r in reps]
            for d in ds[1:]:
                assert all(np.array_equal(a["data"], b["data"]) for a, b in zip(ds[0], d)), (name, red)
            dec[str(red)] = [dict(shape=list(c["data"].shape), sha256=sha(c["data"].tobytes()), prec=c["prec"], sgnd=c["sgnd"], dx=c["dx"], dy=c["dy"])
                             for c in ds[0]]
        if kw.get("reversible", True) and "rates" not in kw:
            assert all(np.array_equal(a, b["data"]) for a, b in zip(comps, rep.decode_comps(cs, 0))), name
        os.makedirs(os.path.join(HERE, "ext"), exist_ok=True)  # (a directory of their own: the globs over the plug-in's own files stay as they are)
        with open(os.path.join(HERE, "ext", name + ".j2k"), "wb") as f:
            f.write(cs)
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist="B", ext=kw, length=len(cs), sha256=sha(cs), decoded_comps=dec,
                          comment=reps[0].comment if keep_com else "")
        print(name, len(cs))


def fake_icc(n, seed):
    """Deterministic stand-in for an ICC profile (the box carries it opaquely)."""
    x, out = seed, bytearray()
    for _ in range(n):
        x = (x * 1103515245 + 12345) & 0x7fffffff
        out.append((x >> 16) & 0xff)
    return bytes(out)


FULL = {
    "c1_512_grey_53": (512, 512, 1, 8, 12345, "A", dict(numres=6)),
    "c2_4096_rgb8_97": (4096, 4096, 3, 8, 12345, "A", dict(numres=6, mct=True, reversible=False)),
    "c3_8192_rgb16_97_6lvl": (8192, 8192, 3, 16, 23456, "A", dict(numres=7, mct=True, reversible=False)),
    "c3_8192_rgb16_97_5lvl": (8192, 8192, 3, 16, 23456, "A", dict(numres=6, mct=True, reversible=False)),
    "c5_frame0_4096x2160_rgb10_97": (4096, 2160, 3, 10, 45678, "A", dict(numres=6, mct=True, reversible=False)),
    "c4_tile_2048_rgb16_53": (2048, 2048, 3, 16, 34567, "A", dict(numres=6, mct=True)),
    # one tile row of C4 (8 tiles of 2048^2 at non-zero origins): what one rank of the 8-GPU job encodes
    "c4_slice_16384x2048_rgb16_53_tile2048": (16384, 2048, 3, 16, 34567, "A", dict(numres=6, mct=True, tile=2048)),
}

# Tiled full-size images of which only codestream hashes are kept (no decode: 5/3, the bytes are the claim):
# name -> (w, h, nc, prec, seed, dist, kwargs, tile ranges [(first, count)] whose tile-part bytes are hashed on their own)
FULL_TILED = {
    # BASELINE config 4 whole: what `bench.py --mode c4` encodes at N = 1 (rank 0's seed)
    "c4_16384_rgb16_53_tile2048": (16384, 16384, 3, 16, 34567, "A", dict(numres=6, mct=True, tile=2048), [(0, 8), (8, 16), (56, 8)]),
    # three tile rows: rows 1..2 (tiles 8..23, y-origin 2048) are what a middle rank of the 8-GPU job sees through its
    # base-pointer offset
    "c4_rows_16384x6144_rgb16_53_tile2048": (16384, 6144, 3, 16, 34567, "A", dict(numres=6, mct=True, tile=2048), [(8, 16), (0, 8)]),
}

# Full-size rate control: >= 8192 code-blocks and 16-bit samples, so the frame takes two coder groups and its
# longest decision streams go to the scalar coder (hash only; COM kept like in RATES)
FULL_RATES = {
    "rh1_4096_rgb16_97_r20": (4096, 4096, 3, 16, 23456, "A", dict(numres=6, mct=True, reversible=False), [20.0]),
}


def sha(b):
    return hashlib.sha256(b).hexdigest()


def psnr(a, b, prec):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10 * np.log10(((1 << prec) - 1) ** 2 / mse)


def full_entries(args, meta, rep, newest):
    for name, (w, h, nc, prec, seed, dist, kw) in FULL.items():
        if args.only is not None and name not in args.only:
            continue
        t0 = time.time()
        pl = synth.planes(w, h, nc, prec, seed, dist)
        p = make_params(w, h, nc, prec, **kw)
        cs = strip_com(rep.encode(pl, p, threads=8))
        t1 = time.time()
        dec = rep.decode(cs, threads=8)
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist=dist, params=kw,
                          length=len(cs), sha256=sha(cs), decoded_sha256=sha(dec.tobytes()),
                          psnr=None if kw.get("reversible", True) else round(psnr(dec, pl, prec), 4))
        if kw.get("reversible", True):
            assert np.array_equal(dec, pl), name
        print(name, len(cs), meta[name]["psnr"], f"enc {t1 - t0:.1f}s total {time.time() - t0:.1f}s", flush=True)
        del pl, dec
    for name, (w, h, nc, prec, seed, dist, kw, ranges) in FULL_TILED.items():
        if args.only is None or name not in args.only:  # (minutes and tens of GB each: only on request)
            continue
        t0 = time.time()
        pl = synth.planes(w, h, nc, prec, seed, dist)
        p = make_params(w, h, nc, prec, **kw)
        cs = strip_com(rep.encode(pl, p, threads=8))
        del pl
        # tile-parts: SOT (ff90) Lsot=10 Isot(2) Psot(4) TPsot TNsot; one tile-part per tile, in tile order
        pos = cs.index(b"\xff\x90")
        spans = []
        while cs[pos:pos + 2] == b"\xff\x90":
            isot = int.from_bytes(cs[pos + 4:pos + 6], "big")
            psot = int.from_bytes(cs[pos + 6:pos + 10], "big")
            assert isot == len(spans), name
            spans.append((pos, psot))
            pos += psot
        assert cs[pos:] == b"\xff\xd9", name
        tp = {}
        for first, count in ranges:
            a, b = spans[first][0], spans[first + count - 1][0] + spans[first + count - 1][1]
            tp[f"{first}:{count}"] = dict(length=b - a, sha256=sha(cs[a:b]))
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist=dist, params=kw, length=len(cs), sha256=sha(cs),
                          main_header_len=spans[0][0], tileparts=tp)
        print(name, len(cs), f"total {time.time() - t0:.1f}s", flush=True)
        del cs
    for name, (w, h, nc, prec, seed, dist, kw, rates) in FULL_RATES.items():
        if args.only is not None and name not in args.only:
            continue
        t0 = time.time()
        pl = synth.planes(w, h, nc, prec, seed, dist)
        p = make_params(w, h, nc, prec, layers=len(rates), **kw)
        f = newest.encode_rates(pl, p, rates, threads=8)
        dec = newest.decode(f, threads=8)
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist=dist, params=kw, rates=rates,
                          comment=newest.comment, length=len(f), sha256=sha(f), decoded_sha256=sha(dec.tobytes()),
                          psnr=round(psnr(dec, pl, prec), 4), library=newest.version)
        print(name, len(f), meta[name]["psnr"], f"total {time.time() - t0:.1f}s", flush=True)
        del pl, dec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also (re)generate the full-size hashes")
    ap.add_argument("--reduced", action="store_true",
                    help="only add decoded_reduced_sha256 (libopenjp2 decodes at cp_reduce 1 and 2) to every stored file's entry")
    ap.add_argument("--ext", action="store_true", help="only (re)generate the EXT files (features outside the plug-in's own writer)")
    ap.add_argument("--only", nargs="*", default=None, help="regenerate only these FULL / FULL_RATES entries (implies --full)")
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

    newest = max([rep] + others, key=lambda r: tuple(int(x) for x in r.version.split(".")))
    if args.ext:
        ext_entries(meta, [rep] + others)
        with open(meta_path, "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        return
    if args.reduced:
        # N4 decode: what libopenjp2 returns for cp_reduce = 1, 2 (setup_decoder before read_header, the order OpenJPEG
        # documents; the reference's ReadFile sets it after the header, which only its Grok fork honours)
        import glob
        for path in sorted(glob.glob(os.path.join(HERE, "*.j2k")) + glob.glob(os.path.join(HERE, "*.jp2"))):
            name = os.path.basename(path).rsplit(".", 1)[0]
            data = open(path, "rb").read()
            red = {}
            for r in (1, 2, 3):
                if r > meta[name]["params"].get("numres", 6) - 1:
                    continue
                outs = [o.decode_ref(data, r, 1)[0] for o in [rep] + others]
                assert all(np.array_equal(outs[0], x) for x in outs[1:]), (name, r)
                red[str(r)] = sha(outs[0].tobytes())
            full = [o.decode_ref(data, 0, 1)[0] for o in [rep] + others]
            assert all(np.array_equal(full[0], x) for x in full[1:]), name
            meta[name]["decoded_reduced_sha256"] = red
            meta[name].setdefault("decoded_sha256", sha(full[0].tobytes()))
            assert meta[name]["decoded_sha256"] == sha(full[0].tobytes()), name
            print(name, sorted(red))
        with open(meta_path, "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        return
    if args.only is not None:
        full_entries(args, meta, rep, newest)
        with open(meta_path, "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        return

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

    newest = max([rep] + others, key=lambda r: tuple(int(x) for x in r.version.split(".")))
    for name, (w, h, nc, prec, seed, kw, cspace, icc_len, alpha) in JP2.items():
        pl = synth.planes(w, h, nc, prec, seed, "B")
        p = make_params(w, h, nc, prec, **kw)
        icc = fake_icc(icc_len, seed) if icc_len else None
        f = newest.encode_jp2(pl, p, cspace, icc, alpha)
        for o in [rep] + others:
            if o is newest or icc or cspace in (4, 5):
                continue
            g = o.encode_jp2(pl, p, cspace, icc, alpha)  # differs only in the version text of the COM segment
            assert g.replace(o.comment.encode(), newest.comment.encode()) == f, (name, o.version)
        dec, info = newest.decode_ex(f)
        assert info["jp2"] and info["icc"] == (icc or b""), name
        with open(os.path.join(HERE, name + ".jp2"), "wb") as fh:
            fh.write(f)
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist="B", params=kw, color_space=cspace,
                          icc_len=icc_len, icc_seed=seed, alpha_channel=alpha, comment=newest.comment, length=len(f),
                          sha256=sha(f), decoded_sha256=sha(dec.tobytes()), library=newest.version)
        print(name, len(f))

    for name, (w, h, nc, prec, seed, kw, rates) in RATES.items():
        pl = synth.planes(w, h, nc, prec, seed, "B")
        p = make_params(w, h, nc, prec, layers=len(rates), **kw)
        f = newest.encode_rates(pl, p, rates)
        for o in [rep] + others:
            if o is not newest:
                assert o.encode_rates(pl, p, rates).replace(o.comment.encode(), newest.comment.encode()) == f, (name, o.version)
        dec = newest.decode(f)
        with open(os.path.join(HERE, name + ".j2k"), "wb") as fh:
            fh.write(f)
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist="B", params=kw, rates=rates,
                          comment=newest.comment, length=len(f), sha256=sha(f), decoded_sha256=sha(dec.tobytes()),
                          psnr=round(psnr(dec, pl, prec), 4) if not np.array_equal(dec, pl) else None, library=newest.version)
        print(name, len(f), meta[name]["psnr"])

    for name, (w, h, nc, prec, seed, kw, q) in QUALITY.items():
        pl = synth.planes(w, h, nc, prec, seed, "B")
        p = make_params(w, h, nc, prec, layers=len(q), **kw)
        f = newest.encode_psnr(pl, p, q)
        for o in [rep] + others:
            if o is not newest:
                assert o.encode_psnr(pl, p, q).replace(o.comment.encode(), newest.comment.encode()) == f, (name, o.version)
        dec = newest.decode(f)
        with open(os.path.join(HERE, name + ".j2k"), "wb") as fh:
            fh.write(f)
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist="B", params=kw, psnr_targets=q,
                          comment=newest.comment, length=len(f), sha256=sha(f), decoded_sha256=sha(dec.tobytes()),
                          psnr=round(psnr(dec, pl, prec), 4) if not np.array_equal(dec, pl) else None, library=newest.version)
        print(name, len(f), meta[name]["psnr"])

    for name, (w, h, nc, prec, seed, kw, order, rates) in ORDERS.items():
        pl = synth.planes(w, h, nc, prec, seed, "B")
        kw2 = dict(kw, layers=len(rates)) if rates else dict(kw)
        p = make_params(w, h, nc, prec, prog=order, **kw2)
        outs = []
        for o in [newest, rep] + others:
            o.set_progression(order)
            outs.append((o.encode_rates(pl, p, rates) if rates else o.encode(pl, p)).replace(o.comment.encode(), newest.comment.encode()))
            o.set_progression(0)
        assert all(x == outs[0] for x in outs), name
        f = outs[0]
        dec = newest.decode(f)
        with open(os.path.join(HERE, name + ".j2k"), "wb") as fh:
            fh.write(f)
        meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist="B", params=kw2, order=order, rates=rates,
                          comment=newest.comment, length=len(f), sha256=sha(f), decoded_sha256=sha(dec.tobytes()), library=newest.version)
        print(name, len(f))

    # JP2 wrapper and rate control together: the boxes in front of the codestream count against the budget
    name, (w, h, nc, prec, seed, kw, rates, cspace, alpha) = "jr1_300x200_rgba8_jp2_srgb_alpha_r30_8", (
        300, 200, 4, 8, 21, dict(numres=5, mct=True, reversible=False), [30.0, 8.0], 1, 3)
    pl = synth.planes(w, h, nc, prec, seed, "B")
    p = make_params(w, h, nc, prec, layers=len(rates), **kw)
    f = newest.encode_jp2_rates(pl, p, rates, cspace, None, alpha)
    with open(os.path.join(HERE, name + ".jp2"), "wb") as fh:
        fh.write(f)
    meta[name] = dict(width=w, height=h, ncomp=nc, prec=prec, seed=seed, dist="B", params=kw, rates=rates, color_space=cspace,
                      icc_len=0, icc_seed=seed, alpha_channel=alpha, comment=newest.comment, length=len(f), sha256=sha(f),
                      library=newest.version)
    print(name, len(f))

    if args.full:
        full_entries(args, meta, rep, newest)

    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
