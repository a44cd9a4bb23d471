"""The boundary is a C ABI: a plain-C99 program includes include/spectro_hip.h, links libspectro_hip.so and exercises the
host-only entry points (no GPU)."""
import os
import shutil
import subprocess

import pytest

from spectrograms_amd import build as sgbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_is_plain_c_and_host_entry_points_work(tmp_path):
    lib = sgbuild.build()
    exe = str(tmp_path / "host_only")
    src = os.path.join(ROOT, "tests", "c_abi", "host_only.c")
    libdir = os.path.dirname(lib)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), src,
                        "-o", exe, "-L" + libdir, "-lspectro_hip", "-lm", "-Wl,-rpath," + libdir],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout
    assert "passed" in r.stdout
