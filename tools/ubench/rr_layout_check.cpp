// Prints, for every three-pass split the register-tiled kernels instantiate, the LDS conflict cost (rr_cost) of the plain
// layout, of the tabulated swizzle (rr_swizzle) and of the best swizzle a full search finds:
//   g++ -O2 -std=c++17 -I spectrograms_amd/csrc tools/ubench/rr_layout_check.cpp -o /tmp/rr_layout_check && /tmp/rr_layout_check
// Exit code 1 if a tabulated entry is worse than the search result (tests/test_rr_layout.py runs this).
#include <cstdio>
#include "rr_layout.h"
using namespace sgx;
int main() {
    struct { unsigned eb, a, b, c; } cases[] = {{8, 8, 8, 8}, {8, 16, 8, 8}, {8, 16, 16, 8}, {8, 16, 16, 16}, {16, 8, 4, 4}, {16, 8, 8, 4},
                                               {16, 8, 8, 8}, {16, 16, 8, 8}, {16, 16, 16, 8}, {16, 16, 16, 16}};
    int bad = 0;
    for (auto &cs : cases) {
        const unsigned U = 256 / cs.eb;
        const RrSwz plain{cs.b * cs.c + 1, 0, 0, 0};
        const RrSwz tab = rr_swizzle(cs.eb, cs.a, cs.b, cs.c);
        const RrSwz srch = rr_search(cs.eb, cs.a, cs.b, cs.c);
        const unsigned cp = rr_cost(plain, U, cs.a, cs.b, cs.c), ct = rr_cost(tab, U, cs.a, cs.b, cs.c), cb = rr_cost(srch, U, cs.a, cs.b, cs.c);
        const unsigned ideal = 2 * (64 / U) * 6;
        printf("elem %2u B  (%2u,%2u,%2u): plain %3u  table {rs %3u mh %u sh %u ml %2u} %3u  search {rs %3u mh %u sh %u ml %2u} %3u  ideal %u\n", cs.eb, cs.a,
               cs.b, cs.c, cp, tab.rs, tab.mh, tab.sh, tab.ml, ct, srch.rs, srch.mh, srch.sh, srch.ml, cb, ideal);
        if (ct > cb) bad = 1;
    }
    return bad;
}
