"""Builds libspectro_hip.so (hand-written HIP for gfx950 + the C-ABI host code) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libspectro_hip.so")
SOURCES = ["plan.hip", "fft2d.hip", "kernels_generic.hip", "kernels_r32x16.hip", "kernels_fft2d.hip", "kernels_c2c1024.hip", "kernels_reg2d.hip", "kernels_q16x32.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libspectro_hip.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "spectro_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", LIB]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    if verbose and r.stdout:
        print(r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
