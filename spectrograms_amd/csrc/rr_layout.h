// rr_layout.h — where element (k1, hi, lo) of a frame's A x (B C) tile lives in LDS for the register-tiled transforms.
//
// Plain constexpr C++ (no HIP): the kernels, the host-side LDS sizing and tools/ubench/rr_layout_check.cpp share it.
//
// A frame's tile is A rows (k1) of B C complex elements; inside a row the position is p = hi * C + lo.  The passes walk it
// with different lanes-to-elements maps (a wave's lanes, fastest index first):
//   pass 1 writes   p (= hi C + lo) consecutive, k1 in the instruction stream
//   pass 2 r/w      lo, then k1               (hi in the instruction stream)
//   pass 3 r/w      hi, then k1               (lo in the instruction stream)
//   split reads     frame, then k = k1 + A (hi + B lo)
// LDS serves a 16-byte access 16 lanes at a time and an 8-byte access 32 lanes at a time (256 bytes = the 64 banks once):
// a group of U = 256 / sizeof(element) consecutive lanes is conflict-free iff its element indices differ mod U.  With the
// plain layout k1 * (B C + 1) + p, pass 3 of the three-pass f64 transforms puts 4 lanes on every bank quad and pass 2 two
// (measured: n_fft 1024 f64, 63 % of all LDS cycles were bank conflicts).  Padding rows costs LDS (frames per tile), so the
// three-pass splits use an XOR swizzle instead:
//     S(k1, hi, lo) = k1 * RS + ((hi ^ hx(k1)) * C  +  (lo ^ lx(k1, hi))),
//     hx(k1) = (k1 * mh) mod B,   lx(k1, hi) = ((hi >> sh) ^ (k1 * ml)) mod C,   RS = B C + rpad
// whose parameters (mh, sh, ml, rpad) are picked per (U, A, B, C) by counting the conflict cycles of the four access patterns
// above (rr_cost; rr_search tries them all, rr_swizzle holds the results as a table that tests/test_rr_layout.py checks against
// the search).  In the kernels a swizzled access is one XOR of a lane value with a compile-time
// constant, because the fields do not overlap.  Two-pass splits (C = 1) keep k1 * (B + 1) + p, which is conflict-free.
#pragma once

namespace sgx {

struct RrSwz {
    unsigned rs;  // row stride (elements)
    unsigned mh, sh, ml;
};

constexpr unsigned rr_hx(const RrSwz &z, unsigned B, unsigned k1) { return (k1 * z.mh) & (B - 1); }
constexpr unsigned rr_lx(const RrSwz &z, unsigned C, unsigned k1, unsigned hi) { return ((hi >> z.sh) ^ (k1 * z.ml)) & (C - 1); }
// position inside the frame tile (without the frame offset)
constexpr unsigned rr_index(const RrSwz &z, unsigned B, unsigned C, unsigned k1, unsigned hi, unsigned lo) {
    return k1 * z.rs + (((hi ^ rr_hx(z, B, k1)) * C) | (lo ^ rr_lx(z, C, k1, hi)));
}
constexpr unsigned rr_frame_stride(unsigned A, unsigned rs) { return (A * rs) | 1u; }  // odd: lanes over frames spread over the banks

// conflict cycles of one U-lane group: the largest number of lanes on one residue mod U
template <typename F>
constexpr unsigned rr_group_cycles(unsigned U, unsigned lane0, F index_of_lane) {
    unsigned worst = 0;
    for (unsigned res = 0; res < U; ++res) {
        unsigned n = 0;
        for (unsigned l = 0; l < U; ++l)
            if (index_of_lane(lane0 + l) % U == res) ++n;
        if (n > worst) worst = n;
    }
    return worst;
}

// LDS cycles (in units of a conflict-free group access) of one wave's pass over the four patterns, weighted by how often
// each runs per element: pass 1 once, passes 2 and 3 twice (read + write), the split once
constexpr unsigned rr_cost(const RrSwz &z, unsigned U, unsigned A, unsigned B, unsigned C) {
    const unsigned fs = rr_frame_stride(A, z.rs);
    unsigned cost = 0;
    for (unsigned g = 0; g < 64 / U; ++g) {
        const unsigned l0 = g * U;
        // instruction-stream indices sampled at a few values (the patterns repeat)
        for (unsigned c = 0; c < 2; ++c) {
            const unsigned k1c = c ? A - 1 : 1 % A, hic = c ? B - 1 : 1 % B, loc = c ? C - 1 : 1 % C;
            cost += rr_group_cycles(U, l0, [&](unsigned l) { const unsigned p = l % (B * C), f = l / (B * C); return f * fs + rr_index(z, B, C, k1c, p / C, p % C); });
            cost += 2 * rr_group_cycles(U, l0, [&](unsigned l) { const unsigned q = l % (A * C), f = l / (A * C); return f * fs + rr_index(z, B, C, q / C, hic, q % C); });
            cost += 2 * rr_group_cycles(U, l0, [&](unsigned l) { const unsigned q = l % (A * B), f = l / (A * B); return f * fs + rr_index(z, B, C, q / B, q % B, loc); });
            // split: 8 frames per tile is the common case; bins k and the mirrored m - k
            const unsigned k0 = c ? 37u % (A * B * C / 2) : 1u;
            cost += rr_group_cycles(U, l0, [&](unsigned l) {
                const unsigned f = l & 7u, k = (k0 + (l >> 3)) % (A * B * C), q = k / A;
                return f * fs + rr_index(z, B, C, k % A, q % B, q / B);
            });
        }
    }
    return cost;
}

// full parameter search (host tools only: too many steps for a constant expression in the kernels)
inline RrSwz rr_search(unsigned elem_bytes, unsigned A, unsigned B, unsigned C) {
    if (C <= 1) return RrSwz{B * C + 1, 0, 0, 0};
    const unsigned U = 256 / elem_bytes;
    RrSwz best{B * C + 1, 0, 0, 0};
    unsigned best_cost = rr_cost(best, U, A, B, C);
    unsigned lb = 0;
    while ((1u << lb) < B) ++lb;
    for (unsigned rpad = 0; rpad < 2; ++rpad)
        for (unsigned mh = 0; mh < (B < 4 ? B : 4); ++mh)
            for (unsigned sh = 0; sh <= lb; ++sh)
                for (unsigned ml = 0; ml < C; ++ml) {
                    const RrSwz z{B * C + rpad, mh, sh, ml};
                    const unsigned c = rr_cost(z, U, A, B, C);
                    if (c < best_cost) {
                        best = z;
                        best_cost = c;
                    }
                }
    return best;
}

// the layout of an (A, B, C) split for elements of `elem_bytes` (8: f32 pairs, 16: f64 pairs): rr_search's results for the
// three-pass splits the kernels instantiate (reg_radix.h), the plain padded rows for everything else
constexpr RrSwz rr_swizzle(unsigned elem_bytes, unsigned A, unsigned B, unsigned C) {
    struct Entry { unsigned eb, a, b, c; RrSwz z; };
    constexpr Entry table[] = {
        {8, 8, 8, 8, {64, 1, 0, 0}},     {8, 16, 8, 8, {64, 1, 0, 0}},    {8, 16, 16, 8, {128, 3, 1, 0}},  {8, 16, 16, 16, {256, 1, 0, 0}},
        {16, 8, 4, 4, {16, 1, 0, 0}},    {16, 8, 8, 4, {32, 1, 1, 1}},    {16, 8, 8, 8, {64, 1, 0, 0}},    {16, 16, 8, 8, {64, 1, 0, 0}},
        {16, 16, 16, 8, {128, 1, 1, 0}}, {16, 16, 16, 16, {256, 0, 0, 8}},
    };
    for (const Entry &e : table)
        if (e.eb == elem_bytes && e.a == A && e.b == B && e.c == C) return e.z;
    return RrSwz{B * C + 1, 0, 0, 0};
}

// The layout of one kernel instance.  index = k1 RS + (k1_mask(k1) ^ hi_part(hi) ^ lo): in each pass two of the three terms are
// lane values built once per work item and the third is a compile-time constant of the unrolled loop.
template <unsigned ElemBytes, int A, int B, int C>
struct RrLayout {
    // scalars, read by value: a struct member passed by reference would be an odr-use, and device code then loads the
    // parameters from memory at run time instead of folding them
    static constexpr unsigned RS = rr_swizzle(ElemBytes, A, B, C).rs;  // row stride (complex elements)
    static constexpr unsigned MH = rr_swizzle(ElemBytes, A, B, C).mh, SH = rr_swizzle(ElemBytes, A, B, C).sh, ML = rr_swizzle(ElemBytes, A, B, C).ml;
    static constexpr unsigned FS = rr_frame_stride(A, RS);             // frame / sequence stride
    static constexpr unsigned k1_mask(unsigned k1) { return (((k1 * MH) & (B - 1)) * C) | ((k1 * ML) & (C - 1)); }  // the row's share of the swizzle
    static constexpr unsigned hi_part(unsigned hi) { return (hi * C) | ((hi >> SH) & (C - 1)); }
    // element k = k1 + A (hi + B lo) of the finished transform
    static constexpr unsigned of_output(unsigned k) {
        if (C <= 1) return (k % A) * RS + k / A;  // two passes: plain rows
        return (k % A) * RS + (k1_mask(k % A) ^ hi_part((k / A) % B) ^ (k / (A * B)));
    }
};

}  // namespace sgx
