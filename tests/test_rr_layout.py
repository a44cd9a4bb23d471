"""LDS layout of the register-tiled transforms (spectrograms_amd/csrc/rr_layout.h): the tabulated XOR swizzles must be as good
as a full parameter search, conflict-free (or one extra cycle) in every pass, and a bijection on the tile.  Host-only: the
header is plain constexpr C++ and is compiled here with g++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spectrograms_amd", "csrc")

BIJECTION_SRC = r"""
#include <cstdio>
#include <vector>
#include "rr_layout.h"
using namespace sgx;
template <unsigned EB, int A, int B, int C> int one() {
    typedef RrLayout<EB, A, B, C> L;
    std::vector<int> seen(A * L::RS, 0);
    int bad = 0;
    for (unsigned k1 = 0; k1 < (unsigned)A; ++k1)
        for (unsigned hi = 0; hi < (unsigned)B; ++hi)
            for (unsigned lo = 0; lo < (unsigned)C; ++lo) {
                const unsigned pass1 = k1 * L::RS + (L::k1_mask(k1) ^ L::hi_part(hi) ^ lo);   // as the passes address it
                const unsigned k = k1 + A * (hi + B * lo);
                if (pass1 >= seen.size() || seen[pass1]++ || pass1 != L::of_output(k)) ++bad;
                if (C > 1 && pass1 != rr_index(rr_swizzle(EB, A, B, C), B, C, k1, hi, lo)) ++bad;
            }
    if (bad) printf("bad layout: elem %u (%d,%d,%d)\n", EB, A, B, C);
    return bad;
}
int main() {
    int bad = 0;
    bad += one<8, 8, 8, 8>() + one<8, 16, 8, 8>() + one<8, 16, 16, 8>() + one<8, 16, 16, 16>();
    bad += one<16, 8, 4, 4>() + one<16, 8, 8, 4>() + one<16, 8, 8, 8>() + one<16, 16, 8, 8>() + one<16, 16, 16, 8>() + one<16, 16, 16, 16>();
    bad += one<8, 16, 16, 1>() + one<8, 25, 8, 1>() + one<16, 8, 8, 1>() + one<8, 32, 30, 1>();
    return bad ? 1 : 0;
}
"""


def _run(tmp_path, name, src_path=None, src_text=None):
    if src_text is not None:
        src_path = tmp_path / (name + ".cpp")
        src_path.write_text(src_text)
    exe = tmp_path / name
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", CSRC, str(src_path), "-o", str(exe)], check=True)
    return subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)


def test_tabulated_swizzles_match_the_search(tmp_path):
    r = _run(tmp_path, "rr_layout_check", src_path=os.path.join(ROOT, "tools", "ubench", "rr_layout_check.cpp"))
    assert r.returncode == 0, r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("elem")]
    assert len(lines) == 10
    for l in lines:
        # "... plain  88  table {...}  50  search {...}  50  ideal 48"
        plain = int(l.split("plain")[1].split()[0])
        table = int(l.split("}")[1].split()[0])
        ideal = int(l.split("ideal")[1])
        assert table < plain, l
        assert table <= ideal + ideal // 6, l  # at most one extra cycle in one of the patterns


def test_layout_is_a_bijection_and_consistent(tmp_path):
    r = _run(tmp_path, "rr_bijection", src_text=BIJECTION_SRC)
    assert r.returncode == 0, r.stdout
