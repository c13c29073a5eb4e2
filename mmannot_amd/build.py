"""Build the in-tree shared libraries.

    python -m mmannot_amd.build            # everything
    libmmannot_amd.so   HIP kernels + C ABI (hipcc --offload-arch=gfx950), include/mmannot_amd.h

hipcc cross-compiles gfx950 without a GPU; the .so travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmmannot_amd.so")


def _sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp")) and not f.startswith("cli_"):
            out.append(os.path.join(CSRC, f))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    srcs = _sources()
    deps = srcs + [os.path.join(ROOT, "include", "mmannot_amd.h")] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    if force or _stale(LIB, deps):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-result",
               "-I", os.path.join(ROOT, "include"), "-o", LIB]
        for s in srcs:
            cmd += ["-x", "hip", s]
        cmd += ["-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


def build_cli(force: bool = False, verbose: bool = False):
    """mmannot-amd command-line binary (drop-in for `mmannot`), links the shared library."""
    src = os.path.join(CSRC, "cli_main.cpp")
    if not os.path.exists(src):
        return None
    exe = os.path.join(PKG, "mmannot-amd")
    if force or _stale(exe, [src, LIB]):
        cmd = ["hipcc", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe, src,
               "-L", PKG, "-lmmannot_amd", "-Wl,-rpath,$ORIGIN", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return exe


def build_addnh(force: bool = False, verbose: bool = False):
    """mmannot-amd-addnh: the companion SAM filter (host only)."""
    src = os.path.join(CSRC, "cli_addnh.cpp")
    exe = os.path.join(PKG, "mmannot-amd-addnh")
    if os.path.exists(src) and (force or _stale(exe, [src])):
        cmd = ["g++", "-O2", "-std=c++17", "-o", exe, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return exe


def build_all(force: bool = False, verbose: bool = False):
    build_lib(force, verbose)
    build_cli(force, verbose)
    build_addnh(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
