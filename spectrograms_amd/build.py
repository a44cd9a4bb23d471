"""Builds libspectro_hip.so (hand-written HIP for gfx950 + the C-ABI host code) in-tree with hipcc.

Every translation unit is compiled to its own object under build/obj (in parallel, skipped when it is newer than its source and
the shared headers) and the objects are linked into spectrograms_amd/libspectro_hip.so.  `variant()` builds the same library with
extra -D flags on chosen sources into build/libsgx_<name>.so for A/B timing runs (loaded through SGX_LIB_PATH); the product
library itself reads no environment switches.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libspectro_hip.so")
OBJ = os.path.join(ROOT, "build", "obj")
# the single source list (Makefile and tools/ read it through `python -m spectrograms_amd.build --sources`)
SOURCES = ["plan.hip", "fft2d.hip", "shard.hip", "kernels_generic.hip", "kernels_r32x16.hip", "kernels_r32x32.hip", "kernels_istft2048.hip", "kernels_d32x16.hip", "kernels_istft_d1024.hip", "kernels_r64x32.hip", "kernels_d32x32.hip", "kernels_fft2d.hip",
           "kernels_c2c1024.hip", "kernels_reg2d.hip", "membench.hip", "bluestein.hip", "bigfft.hip"]
ARCH = "gfx950"
LINK_LIBS = ["-ldl"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libspectro_hip.so")


def _cflags() -> list[str]:
    return ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _headers_mtime() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hs.append(os.path.join(ROOT, "include", "spectro_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, obj: str, extra: list[str], force: bool) -> str | None:
    sp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), _headers_mtime()):
        return None
    cmd = [_hipcc(), *_cflags(), *extra, "-c", sp, "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout)
    return r.stdout


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "spectro_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _link(objs: list[str], out: str) -> None:
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out, *objs, *LINK_LIBS]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or min(len(SOURCES), max(1, (os.cpu_count() or 2)))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    with ThreadPoolExecutor(jobs) as ex:
        outs = list(ex.map(lambda so: _compile(so[0], so[1], [], force), zip(SOURCES, objs)))
    if verbose:
        for s, o in zip(SOURCES, outs):
            print(f"{s}: {'up to date' if o is None else 'compiled'}")
    _link(objs, LIB)
    return LIB


def variant(name: str, flags: list[str], sources: tuple[str, ...] = ("kernels_r32x16.hip",)) -> str:
    """A/B build: `flags` applied to `sources` only, everything else taken from the product objects."""
    build()
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    for s in SOURCES:
        if s in sources:
            o = os.path.join(OBJ, s.replace(".hip", f".{name}.o"))
            _compile(s, o, flags, True)
        else:
            o = os.path.join(OBJ, s.replace(".hip", ".o"))
            _compile(s, o, [], False)
        objs.append(o)
    out = os.path.join(ROOT, "build", f"libsgx_{name}.so")
    _link(objs, out)
    return out


if __name__ == "__main__":
    import sys

    if "--sources" in sys.argv:
        print(" ".join(os.path.join("spectrograms_amd", "csrc", s) for s in SOURCES))
    elif len(sys.argv) > 2 and sys.argv[1] == "--variant":
        rest, srcs = sys.argv[3:], []
        while len(rest) >= 2 and rest[0] == "--src":  # --variant NAME [--src file.hip]... flags...
            srcs.append(rest[1])
            rest = rest[2:]
        print(variant(sys.argv[2], rest, tuple(srcs) or ("kernels_r32x16.hip",)))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
