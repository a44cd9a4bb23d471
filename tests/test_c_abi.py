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


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_device_smoke(tmp_path):
    """No Python in the data path: a C program drives the library with hipMalloc'ed buffers through the C ABI."""
    lib = sgbuild.build()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "device_smoke")
    src = os.path.join(ROOT, "tests", "c_abi", "device_smoke.c")
    libdir = os.path.dirname(lib)
    r = subprocess.run([hipcc, "-x", "c", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                        "-I/opt/rocm/include", src, "-o", exe, "-L" + libdir, "-lspectro_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                        "-Wl,-rpath," + libdir + ":/opt/rocm/lib"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "device smoke passed" in r.stdout, r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_shard_execute_multi_rank(tmp_path):
    """sgx_comm_create / sgx_shard_execute / sgx_gather with 2 and 3 ranks (host threads sharing the one GPU of the test box), equal
    and ragged shards, in-place and separate shard buffers, against a single launch bit for bit.  The collectives are served by
    tests/c_abi/fake_rccl.c, which the library finds through its own dlsym lookup because the executable exports it."""
    lib = sgbuild.build()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "shard_ranks")
    srcs = [os.path.join(ROOT, "tests", "c_abi", f) for f in ("shard_ranks.c", "fake_rccl.c")]
    libdir = os.path.dirname(lib)
    r = subprocess.run([hipcc, "-x", "c", "-std=gnu99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                        "-I/opt/rocm/include", *srcs, "-o", exe, "-rdynamic", "-L" + libdir, "-lspectro_hip", "-L/opt/rocm/lib",
                        "-lamdhip64", "-lm", "-lpthread", "-Wl,-rpath," + libdir + ":/opt/rocm/lib"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "multi-rank shard test passed" in r.stdout, r.stdout
