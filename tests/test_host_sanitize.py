"""Host-side C++ of libj2k_hip (geometry, Tier-2 planner, JP2 wrapper, rate control, worker threads) under
ASan + UBSan and under ThreadSanitizer.
GPU sanitizers are not available on the pool, so the device-independent logic gets its own CPU build."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("flags", ["address,undefined", "thread"])
def test_host_logic_under_sanitizers(tmp_path, flags):
    csrc = os.path.join(ROOT, "j2k_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")] + \
           [os.path.join(csrc, f) for f in ("geometry.cpp", "tier2.cpp", "jp2.cpp", "rate_control.cpp", "workers.cpp", "bands.cpp")]
    exe = str(tmp_path / "host_sanitize")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + flags, "-fno-sanitize-recover=all",
                            "-I" + os.path.join(ROOT, "include"), *srcs, "-lpthread", "-o", exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert run.stdout.count("ok ") == 59


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_decode_parser_under_sanitizers(tmp_path):
    """decode_plan.cpp (JP2 boxes, headers, packet headers) on every golden file and on truncated / bit-flipped copies."""
    import glob
    csrc = os.path.join(ROOT, "j2k_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "native", "decode_sanitize.cpp")] + [os.path.join(csrc, f) for f in ("decode_plan.cpp", "geometry.cpp")]
    exe = str(tmp_path / "decode_sanitize")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-I" + os.path.join(ROOT, "include"), *srcs, "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.j2k")) + glob.glob(os.path.join(ROOT, "tests", "golden", "*.jp2")) +
                   glob.glob(os.path.join(ROOT, "tests", "golden", "ext", "*.j2k")))  # (ext: precincts, sub-sampling, offsets, code-block styles)
    # packed packet headers (PPT / PPM) and QCC / COC variants: crafted from committed files (tests/test_read_fallback.py)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_read_fallback as rf
    src = open(os.path.join(ROOT, "tests", "golden", "ext", rf.PACKED_SOURCE + ".j2k"), "rb").read()
    g6 = open(os.path.join(ROOT, "tests", "golden", "g6_300x200_rgb16_97_ict.j2k"), "rb").read()
    for name, data in (("ppt.j2k", rf._repack_headers(src, "ppt")), ("ppm.j2k", rf._repack_headers(src, "ppm")),
                       ("qcc.j2k", rf._with_qcc(g6, (0, 1, 2), True)), ("coc.j2k", rf._with_coc(g6, (0, 2)))):
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        files.append(path)
    # 2000 mutations per file (J2K_FUZZ_MUTATIONS; the seed is J2K_FUZZ_SEED and is printed with a failure): the files are dealt to
    # a few processes side by side, so the depth costs about a minute of wall clock
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    nproc = max(1, min(6, (os.cpu_count() or 2) - 1))
    procs = [subprocess.Popen([exe] + files[k::nproc], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for k in range(nproc) if files[k::nproc]]
    for pr in procs:
        out, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, (out[-2000:], err[-4000:])
        assert out.startswith("planned ")
