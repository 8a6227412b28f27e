// run-viterbi.cpp -- the reference's debug harness (src/nanocall/run-viterbi.cpp:38-57) on top of the
// nanocall_amd host classes: read a scaled pore model, state transitions and events in the
// reference's text formats, run Viterbi (on the GPU), print the base sequence.
//
//   run-viterbi -p model.tsv (-s transitions.tsv | --pr-skip P --pr-stay Q) -e events.tsv [--fasta NAME]
//
//   model.tsv        kmer level_mean level_stdv sd_mean sd_stdv      (Pore_Model operator>>, Pore_Model.hpp:251-287)
//   transitions.tsv  kmer_i kmer_j log_p                             (State_Transitions operator>>, :237-252)
//   events.tsv       mean stdv start length                          (Event operator>>, Event.hpp:59-68)
//
// Build: g++ -std=c++17 -O2 -Iinclude tools/run-viterbi.cpp -Lnanocall_amd -lnanocall_hip -Wl,-rpath,... -o run-viterbi
#include <algorithm>
#include <fstream>
#include <iostream>
#include <sstream>

#include "nanocall_amd/nanocall_amd.hpp"

using namespace nanocall_amd;
typedef Pore_Model<float, 6> Pore_Model_Type;
typedef State_Transitions<float, 6> State_Transitions_Type;
typedef Event<float, 6> Event_Type;
typedef Event_Sequence<float, 6> Event_Sequence_Type;
typedef Viterbi<float, 6> Viterbi_Type;

int main(int argc, char* argv[])
{
    std::string pm_fn, st_fn, ev_fn, fasta_name;
    float pr_skip = .3f, pr_stay = .1f;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string a = argv[i], v = argv[i + 1];
        if (a == "-p") pm_fn = v; else if (a == "-s") st_fn = v; else if (a == "-e") ev_fn = v;
        else if (a == "--pr-skip") pr_skip = std::stof(v); else if (a == "--pr-stay") pr_stay = std::stof(v);
        else if (a == "--fasta") fasta_name = v;
        else { std::cerr << "unknown option " << a << std::endl; return 2; }
    }
    if (pm_fn.empty() || ev_fn.empty()) { std::cerr << "usage: run-viterbi -p model -e events [-s transitions | --pr-skip P --pr-stay Q]\n"; return 2; }
    try {
        // model: rows may come in any order; '#' and header lines are skipped (Pore_Model.hpp:262-266)
        std::vector<float> table(4096 * 4, 0.f);
        {
            std::ifstream is(pm_fn);
            std::string line; unsigned n = 0;
            while (std::getline(is, line)) {
                std::istringstream iss(line); std::string k;
                iss >> k;
                if (k.empty() || k[0] == '#' || line.find("kmer") != std::string::npos) continue;
                size_t j = Kmer<6>::to_int(k);
                if (j >= 4096) { std::cerr << "bad kmer " << k << std::endl; return 1; }
                iss >> table[4 * j] >> table[4 * j + 1] >> table[4 * j + 2] >> table[4 * j + 3];
                ++n;
            }
            if (n != 4096) { std::cerr << "unexpected number of states" << std::endl; return 1; }
        }
        Pore_Model_Type pm;
        pm.load_from_vector(table);   // "scaled pore model file": used as is, like the reference tool
        State_Transitions_Type st;
        if (st_fn.empty()) {
            st.compute_transitions_fast(pr_skip, pr_stay);
        } else {
            // arcs (i -> j, log p); the device layer needs from_v order: by destination, predecessors ascending
            std::ifstream is(st_fn);
            std::string ki, kj; float p;
            std::vector<std::tuple<unsigned, unsigned, float>> arcs;
            while (is >> ki >> kj >> p) arcs.emplace_back((unsigned)Kmer<6>::to_int(kj), (unsigned)Kmer<6>::to_int(ki), p);
            std::sort(arcs.begin(), arcs.end());
            std::vector<uint32_t> rp(4097, 0); std::vector<uint16_t> pred; std::vector<float> w;
            for (auto& a : arcs) { rp[std::get<0>(a) + 1]++; pred.push_back((uint16_t)std::get<1>(a)); w.push_back(std::get<2>(a)); }
            for (unsigned j = 0; j < 4096; ++j) rp[j + 1] += rp[j];
            check(nchmm_put_transitions(Device::instance().ctx(), 1, rp.data(), pred.data(), w.data()), "nchmm_put_transitions");
        }
        Event_Sequence_Type ev;
        {
            std::ifstream is(ev_fn);
            Event_Type e;
            while (is >> e.mean >> e.stdv >> e.start >> e.length) { e.corrected_mean = e.mean; e.update_logs(); ev.push_back(e); }
        }
        if (ev.empty()) { std::cerr << "no events" << std::endl; return 1; }
        Viterbi_Type vit;
        if (st_fn.empty()) {
            vit.fill(pm, st, ev);
        } else {
            // transitions came from a file into slot 1: run the batch form against that slot
            pm.put(1);
            std::vector<uint64_t> off{0, ev.size()};
            std::vector<float> cm, sd, ls;
            detail::soa(ev, cm, sd, ls);
            std::vector<uint16_t> states(ev.size()); float pp; int32_t status, slot = 1;
            check(nchmm_viterbi(Device::instance().ctx(), 1, off.data(), cm.data(), sd.data(), ls.data(), &slot, &slot, states.data(), &pp, &status), "nchmm_viterbi");
            for (size_t i = 0; i < ev.size(); ++i) {
                ev[i].model_state_idx = states[i];
                ev[i].set_model_state(Kmer<6>::to_string(states[i]));
                ev[i].move = i ? (int)Kmer<6>::min_skip(states[i - 1], states[i]) : 0;
            }
        }
        const std::string seq = ev.get_base_seq();
        if (fasta_name.empty()) {
            std::cout << seq << std::endl;
        } else {
            std::vector<char> buf(seq.size() + seq.size() / 80 + fasta_name.size() + 16);
            size_t n = 0;
            check(nchmm_write_fasta(fasta_name.c_str(), seq.c_str(), 80, buf.data(), buf.size(), &n), "nchmm_write_fasta");
            std::cout.write(buf.data(), n);
        }
        std::cerr << "path_probability " << vit.path_probability() << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
