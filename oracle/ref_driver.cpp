// ref_driver.cpp -- thin extern "C" driver over the parts of the REAL reference that compile from
// their own sources with no stand-ins: src/nanocall/Kmer.hpp (needs only the C++ standard
// library) and src/nanocall/Builtin_Model.cpp (+ src/builtin_models/*.inl).
// Built by `make -C oracle ref` into oracle/_ref/ (git-ignored; never copied into the repo).
// Everything else on the hot path (#include "logger.hpp" / "logsumset.hpp" / "fast5.hpp" /
// "alg.hpp" from the empty hpptools / fast5 submodules) is unbuildable here -- see DESIGN.md.
// TEST INFRASTRUCTURE ONLY.
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "Kmer.hpp"
#include "Builtin_Model.hpp"

typedef Kmer<6> K;

extern "C" {
unsigned ref_kmer_n_states() { return K::n_states; }
unsigned ref_kmer_min_skip(unsigned a, unsigned b) { return K::min_skip(a, b); }
unsigned ref_kmer_prefix(unsigned i, unsigned k) { return K::prefix(i, k); }
unsigned ref_kmer_suffix(unsigned i, unsigned k) { return K::suffix(i, k); }
unsigned ref_kmer_max_self_overlap(unsigned i) { return K::max_self_overlap(i); }
unsigned ref_kmer_to_int(const char* s) { return (unsigned)K::to_int(std::string(s)); }
void ref_kmer_to_string(unsigned k, char* out) { std::string s = K::to_string(k); std::memcpy(out, s.c_str(), s.size() + 1); }
unsigned ref_kmer_neighbour_list(unsigned i, unsigned d, unsigned* out)
{
    const std::vector<unsigned>& v = K::neighbour_list(i, d);
    for (size_t k = 0; k < v.size(); ++k) out[k] = v[k];
    return (unsigned)v.size();
}
unsigned ref_builtin_num() { return Builtin_Model::num; }
unsigned ref_builtin_strand(unsigned i) { return Builtin_Model::strands[i]; }
const char* ref_builtin_name(unsigned i) { return Builtin_Model::names[i].c_str(); }
unsigned ref_builtin_size(unsigned i) { return (unsigned)Builtin_Model::init_lists[i].size(); }
const float* ref_builtin_table(unsigned i) { return Builtin_Model::init_lists[i].data(); }
}
