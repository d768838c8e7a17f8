"""One process, several devices: j2k_hip_encode_batch and j2k_hip_encode_tiles_distributed.  The GPU
box has one device, so the device list names it several times -- every worker still has its own handle, its own
streams and its own arenas, which is all the code path knows about a device."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_case
from j2k_amd import api, synth

pytestmark = pytest.mark.gpu


class Sinks:
    def __init__(self, n):
        self.chunks = [[] for _ in range(n)]
        self.ids = (C.c_void_p * n)(*[i + 1 for i in range(n)])  # user pointer = frame number + 1

        @api.WRITE_FN
        def write(user, buf, nbytes):
            self.chunks[user - 1].append(C.string_at(buf, nbytes))
            return nbytes
        self.fn = write

    def data(self, i):
        return b"".join(self.chunks[i])


def test_device_count():
    L = api.load_library()
    assert L.j2k_hip_device_count() >= 1


@pytest.mark.parametrize("ndev,per", [(1, 1), (2, 2), (3, 1)])
def test_encode_batch_equals_frame_by_frame(golden, ndev, per):
    L = api.load_library()
    g, pl0, _, cs0 = golden_case(golden, "g6_300x200_rgb16_97_ict")
    w, h, nc, prec = g["width"], g["height"], g["ncomp"], g["prec"]
    p = api.make_params(w, h, nc, prec, reversible=False, ycc=True, comment="")
    frames = [synth.ae_frame(pl0, prec)[0]] + [synth.ae_frame(synth.planes(w, h, nc, prec, 40 + k, "AB"[k & 1]), prec)[0] for k in range(6)]
    lay = synth.ae_frame(pl0, prec)[1]
    enc = api.Encoder(0)
    singles = [enc.encode_host(f, lay, p) for f in frames]
    enc.close()
    assert singles[0] == cs0
    planes = (api.Plane * (nc * len(frames)))()
    for f, fr in enumerate(frames):
        one = api.planes_from_layout(fr.ctypes.data, lay, nc)
        for c in range(nc):
            planes[f * nc + c] = one[c]
    sinks = Sinks(len(frames))
    devs = (C.c_int * ndev)(*([0] * ndev))
    rc = L.j2k_hip_encode_batch(devs, ndev, per, C.byref(p), planes, len(frames), sinks.fn, sinks.ids)
    assert rc == 0, L.j2k_hip_multi_last_error()
    assert [sinks.data(i) for i in range(len(frames))] == singles


@pytest.mark.parametrize("name,ndev,extra", [("g4_300x200_rgb16_53_rct_tile128", 1, {}), ("g4_300x200_rgb16_53_rct_tile128", 3, {}),
                                             ("g4_300x200_rgb16_53_rct_tile128", 8, {}), ("g9_150x130_rgb8_97_tile64", 4, {}),
                                             ("g4_300x200_rgb16_53_rct_tile128", 2, dict(jp2=True, color_space=1))])
def test_tiles_distributed_equals_single_device(golden, name, ndev, extra):
    """More devices than tiles, uneven splits, a JP2 wrapper around the distributed codestream: always the file one
    device writes -- and for the plain ones libopenjp2's."""
    L = api.load_library()
    g, pl, _, cs = golden_case(golden, name)
    kw = g["params"]
    frame, lay = synth.ae_frame(pl, g["prec"], row_pad_bytes=8)
    p = api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=kw.get("reversible", True), ycc=kw.get("mct", False),
                        tile_size=kw["tile"], num_resolutions=kw.get("numres", 6), comment="", **extra)
    planes = api.planes_from_layout(frame.ctypes.data, lay, g["ncomp"])
    sinks = Sinks(1)
    devs = (C.c_int * ndev)(*([0] * ndev))
    rc = L.j2k_hip_encode_tiles_distributed(devs, ndev, C.byref(p), planes, sinks.fn, 1)
    assert rc == 0, L.j2k_hip_multi_last_error()
    enc = api.Encoder(0)
    whole = enc.encode_host(frame, lay, p)
    enc.close()
    assert sinks.data(0) == whole
    if not extra:
        assert whole == cs


def test_multi_device_errors():
    L = api.load_library()
    pl = synth.planes(64, 64, 1, 8, 1)
    frame, lay = synth.ae_frame(pl, 8)
    planes = api.planes_from_layout(frame.ctypes.data, lay, 1)
    sinks = Sinks(1)
    bad = api.make_params(64, 64, 1, 8, num_resolutions=9)
    devs = (C.c_int * 1)(0)
    assert L.j2k_hip_encode_batch(devs, 1, 1, C.byref(bad), planes, 1, sinks.fn, sinks.ids) != 0
    assert b"resolutions" in L.j2k_hip_multi_last_error()
    assert L.j2k_hip_encode_tiles_distributed(devs, 1, C.byref(bad), planes, sinks.fn, 1) != 0
    nodev = (C.c_int * 1)(99)
    ok = api.make_params(64, 64, 1, 8, num_resolutions=3)
    assert L.j2k_hip_encode_batch(nodev, 1, 1, C.byref(ok), planes, 1, sinks.fn, sinks.ids) != 0
    assert b"device" in L.j2k_hip_multi_last_error()
