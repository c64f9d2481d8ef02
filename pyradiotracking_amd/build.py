"""Build the native pieces in-tree (no JIT cache: the .so files travel with the repo snapshot).

* ``librt_analyze.so``   gfx950 HIP kernels + C-ABI (include/rt_analyze.h, rt_match.h, rt_format.h) -- the product
* ``_rt_hostcheck.so``   host build of csrc/rt_core.h's scalar logic          -- unit tests only
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "librt_analyze.so")
HOSTCHECK = os.path.join(PKG, "_rt_hostcheck.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources():
    out = [os.path.join(REPO, "include", "rt_analyze.h")]
    for name in sorted(os.listdir(CSRC)):
        out.append(os.path.join(CSRC, name))
    return out


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def build_library(force=False, verbose=False):
    if not force and _newer(LIB, _sources()):
        return LIB
    cmd = [
        hipcc_path(),
        "-O3",
        "-std=c++17",
        "--offload-arch=gfx950",
        "-shared",
        "-fPIC",
        "-ffp-contract=off",
        "-fno-slp-vectorize",  # SLP packing of the butterflies costs ~25 VGPRs in shuffles, no speed
        "-Wno-unused-value",
        "-Wno-pass-failed",
        "-I" + os.path.join(REPO, "include"),
        "-o",
        LIB,
        os.path.join(CSRC, "rt_analyze.hip"),
        os.path.join(CSRC, "rt_match.cpp"),  # host-only parts of the C-ABI (include/rt_match.h, rt_format.h)
        os.path.join(CSRC, "rt_format.cpp"),
    ]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


def build_hostcheck(force=False, verbose=False):
    src = os.path.join(CSRC, "rt_hostcheck.cpp")
    if not force and _newer(HOSTCHECK, [src, os.path.join(CSRC, "rt_core.h")]):
        return HOSTCHECK
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", HOSTCHECK, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return HOSTCHECK


def build_all(force=False, verbose=False):
    return build_library(force, verbose), build_hostcheck(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
