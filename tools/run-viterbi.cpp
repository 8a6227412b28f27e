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
#include "text_formats.hpp"

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
        Pore_Model_Type pm;
        pm.load_from_vector(text_formats::read_model_table(pm_fn));   // "scaled pore model file": used as is, like the reference tool
        State_Transitions_Type st;
        if (st_fn.empty()) st.compute_transitions_fast(pr_skip, pr_stay);
        else text_formats::put_transitions_file(st_fn, 1);
        Event_Sequence_Type ev = text_formats::read_events(ev_fn);
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
