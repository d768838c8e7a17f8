"""Build the in-tree HIP library (libj2k_hip.so) and the C++ host wrapper for gfx950.

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB = os.path.join(PKG, "libj2k_hip.so")
HOSTLIB = os.path.join(PKG, "libj2k_host.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: HIP's __fmul_rn/__fadd_rn are plain operators, so contraction must be off for the
# 9/7 path to round every product and sum separately (bit-exact with the oracle).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off"]
FLAGS += os.environ.get("J2K_EXTRA_CFLAGS", "").split()  # experiments (-D switches); use with force=True


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _deps_newer(obj: str, src: str, headers: list[str]) -> bool:
    return _stale(obj, [src] + headers)


def build_variant(name: str, cflags: list[str], sources: list[str]) -> str:
    """An A/B build: libj2k_hip_<name>.so = the regular objects with `sources` recompiled under extra flags
    (run with J2K_HIP_LIB=<path>)."""
    build_library()
    objdir = os.path.join(PKG, "build")
    objs = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hip"))):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if os.path.basename(src) in sources:
            obj = os.path.join(objdir, os.path.basename(src) + "." + name + ".o")
            subprocess.check_call([HIPCC] + [f for f in FLAGS if f != "-shared"] + cflags + ["-c", "-o", obj, src])
        objs.append(obj)
    out = os.path.join(PKG, "libj2k_hip_" + name + ".so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    return out


def build_library(force: bool = False, verbose: bool = False) -> str:
    """One object per source (compiled in parallel, rebuilt only when the source or a header changed), one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hip")))
    headers = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(PKG, "..", "include", "j2k_hip.h")]
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _deps_newer(obj, src, headers):
            jobs.append([HIPCC] + cflags + ["-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.basename(src) + ".o") for src in srcs]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    """C++ mirror of the reference's codec interface (HipCodec : j2k::Codec) on top of the C ABI."""
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp")))
    if not srcs:
        return ""
    deps = srcs + glob.glob(os.path.join(HOST, "*.h")) + [os.path.join(PKG, "..", "include", "j2k_hip.h")]
    if force or _stale(HOSTLIB, deps + [LIB]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", os.path.join(PKG, "..", "include"),
               "-o", HOSTLIB] + srcs + ["-L", PKG, "-lj2k_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOSTLIB


if __name__ == "__main__":
    print(build_library(verbose=True))
    print(build_host(verbose=True))
