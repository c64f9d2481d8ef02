"""Build the native pieces in-tree (no JIT cache: the .so files travel with the repo snapshot).

* ``librt_analyze.so``   gfx950 HIP kernels + C-ABI (include/rt_analyze.h, rt_match.h, rt_format.h) -- the product
* ``librt_analyze_diag.so``  the same sources with ``-DRT_DIAG`` (csrc/rt_diag.h): the only build that reads the
  laboratory's environment switches -- the fault-injection test loads it; nothing else does
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
LIB_DIAG = os.path.join(PKG, "librt_analyze_diag.so")
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


def _library_cmd(target, extra=()):
    return [
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
        *extra,
        "-I" + os.path.join(REPO, "include"),
        "-o",
        target,
        os.path.join(CSRC, "rt_analyze.hip"),
        os.path.join(CSRC, "rt_match.cpp"),  # host-only parts of the C-ABI (include/rt_match.h, rt_format.h)
        os.path.join(CSRC, "rt_format.cpp"),
    ]


def build_library(force=False, verbose=False, diag=False):
    """The product library and (``diag=True``: ``build_all``, tools/variant.sh, the fault-injection test) its diagnostic twin,
    compiled side by side (two hipcc processes).  The import path (``_native.load_library``) builds the product alone: a cold
    import must not pay for -- or race other ranks on -- a second 2 MB multi-template build nobody loads.  Each target is
    linked under a temporary name and moved into place (``os.replace``), so a concurrent importer never maps a half-written
    file; when one job fails the other is stopped and waited for before the error is raised (no orphaned hipcc)."""
    jobs = []
    for target, extra in ((LIB, ()), (LIB_DIAG, ("-DRT_DIAG",))):
        if target == LIB_DIAG and not diag:
            continue
        if not force and _newer(target, _sources()):
            continue
        tmp = f"{target}.tmp{os.getpid()}"
        cmd = _library_cmd(tmp, extra)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((cmd, subprocess.Popen(cmd), tmp, target))
    failed = None
    for cmd, pr, tmp, target in jobs:
        if failed is not None:
            if pr.poll() is None:
                pr.terminate()
            try:
                pr.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
        elif pr.wait() != 0:
            failed = (pr.returncode, cmd)
        else:
            os.replace(tmp, target)
            continue
        try:
            os.remove(tmp)
        except OSError:
            pass
    if failed is not None:
        raise subprocess.CalledProcessError(failed[0], failed[1])
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
    """everything: the product, its diagnostic twin and the host check (``__graft_entry__.build``)"""
    return build_library(force, verbose, diag=True), build_hostcheck(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
