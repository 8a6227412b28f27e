// nchmm_kmer.hpp -- 6-mer algebra (2 bits per base, first base in the top bits).
// Same functions as the reference's Kmer<6> (src/nanocall/Kmer.hpp:41-148), written as closed-form
// bit operations instead of lazily initialised tables, so they are constexpr-friendly and need no
// mutex (the reference guards its static tables with one, Kmer.hpp:119-146).
#ifndef NCHMM_KMER_HPP
#define NCHMM_KMER_HPP

#include <cstdint>

namespace nchmm {

struct Kmer6 {
    static constexpr unsigned k = 6;
    static constexpr unsigned n_states = 1u << 12;

    // Kmer::prefix / suffix, Kmer.hpp:69-76
    static constexpr unsigned prefix(unsigned i, unsigned len) { return i >> (2 * (k - len)); }
    static constexpr unsigned suffix(unsigned i, unsigned len) { return i & ((1u << (2 * len)) - 1); }

    // Kmer::to_string, Kmer.hpp:41-50
    static void to_chars(unsigned v, char out[6])
    {
        for (unsigned j = 0; j < k; ++j) out[j] = "ACGT"[(v >> (2 * (k - 1 - j))) & 3];
    }

    // Kmer::to_int, Kmer.hpp:12-35; returns n_states on a letter outside ACGT
    static unsigned from_chars(const char* s, unsigned len = k)
    {
        unsigned v = 0;
        for (unsigned j = 0; j < len; ++j) {
            unsigned b;
            switch (s[j]) { case 'A': b = 0; break; case 'C': b = 1; break;
                            case 'G': b = 2; break; case 'T': b = 3; break; default: return n_states; }
            v = (v << 2) | b;
        }
        return v;
    }

    // Kmer::min_skip, Kmer.hpp:51-68: smallest shift d with suffix(k1, 6-d) == prefix(k2, 6-d)
    static constexpr unsigned min_skip(unsigned k1, unsigned k2)
    {
        if (k1 == k2) return 0;
        for (unsigned d = 1; d < k; ++d)
            if (suffix(k1, k - d) == prefix(k2, k - d)) return d;
        return k;
    }

    // Kmer::max_self_overlap, Kmer.hpp:81-110
    static constexpr unsigned max_self_overlap(unsigned i)
    {
        for (unsigned len = k - 1; len >= 1; --len)
            if (suffix(i, len) == prefix(i, len)) return len;
        return 0;
    }

    // Kmer::neighbour_list(i, 1)[b] and (i, 2)[bb], Kmer.hpp:128-142
    static constexpr unsigned step_succ(unsigned i, unsigned b) { return ((i & 0x3FFu) << 2) | b; }
    static constexpr unsigned skip_succ(unsigned i, unsigned bb) { return ((i & 0xFFu) << 4) | bb; }
    // the inverse direction used by the DP: predecessors of j
    static constexpr unsigned step_pred(unsigned j, unsigned x) { return (x << 10) | (j >> 2); }
    static constexpr unsigned skip_pred(unsigned j, unsigned xy) { return (xy << 8) | (j >> 4); }
};

}  // namespace nchmm
#endif
