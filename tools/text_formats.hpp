// text_formats.hpp -- readers for the three text formats of the reference's debug harness, shared by
// tools/run-viterbi.cpp and tools/run-fwbw.cpp:
//   model        kmer level_mean level_stdv sd_mean sd_stdv      (Pore_Model operator>>, Pore_Model.hpp:251-287)
//   transitions  kmer_i kmer_j log_p                             (State_Transitions operator>>, State_Transitions.hpp:237-252)
//   events       mean stdv start length                          (Event operator>>, Event.hpp:59-68)
#pragma once
#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "nanocall_amd/nanocall_amd.hpp"

namespace text_formats {

using namespace nanocall_amd;

// rows may come in any order; '#' and header lines are skipped (Pore_Model.hpp:262-266)
inline std::vector<float> read_model_table(const std::string& fn)
{
    std::vector<float> table(4096 * 4, 0.f);
    std::ifstream is(fn);
    if (!is) throw std::runtime_error("cannot open " + fn);
    std::string line;
    unsigned n = 0;
    while (std::getline(is, line)) {
        std::istringstream iss(line);
        std::string k;
        iss >> k;
        if (k.empty() || k[0] == '#' || line.find("kmer") != std::string::npos) continue;
        const size_t j = Kmer<6>::to_int(k);
        if (j >= 4096) throw std::runtime_error("bad kmer " + k);
        iss >> table[4 * j] >> table[4 * j + 1] >> table[4 * j + 2] >> table[4 * j + 3];
        ++n;
    }
    if (n != 4096) throw std::runtime_error("unexpected number of states");
    return table;
}

// arcs (i -> j, log p) into device transition slot `slot`; the device layer wants from_v order: by destination,
// predecessors ascending
inline void put_transitions_file(const std::string& fn, int slot)
{
    std::ifstream is(fn);
    if (!is) throw std::runtime_error("cannot open " + fn);
    std::string ki, kj;
    float p;
    std::vector<std::tuple<unsigned, unsigned, float>> arcs;
    while (is >> ki >> kj >> p) arcs.emplace_back((unsigned)Kmer<6>::to_int(kj), (unsigned)Kmer<6>::to_int(ki), p);
    std::sort(arcs.begin(), arcs.end());
    std::vector<uint32_t> rp(4097, 0);
    std::vector<uint16_t> pred;
    std::vector<float> w;
    for (auto& a : arcs) { rp[std::get<0>(a) + 1]++; pred.push_back((uint16_t)std::get<1>(a)); w.push_back(std::get<2>(a)); }
    for (unsigned j = 0; j < 4096; ++j) rp[j + 1] += rp[j];
    check(nchmm_put_transitions(Device::instance().ctx(), slot, rp.data(), pred.data(), w.data()), "nchmm_put_transitions");
}

inline Event_Sequence<float, 6> read_events(const std::string& fn)
{
    Event_Sequence<float, 6> ev;
    std::ifstream is(fn);
    if (!is) throw std::runtime_error("cannot open " + fn);
    Event<float, 6> e;
    while (is >> e.mean >> e.stdv >> e.start >> e.length) { e.corrected_mean = e.mean; e.update_logs(); ev.push_back(e); }
    if (ev.empty()) throw std::runtime_error("no events");
    return ev;
}

}  // namespace text_formats
