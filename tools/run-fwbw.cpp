// run-fwbw.cpp -- the reference's second debug harness (src/nanocall/run-fwbw.cpp:43-98) on top of the
// nanocall_amd host classes: scaled pore model, state transitions and events in the reference's text formats,
// forward-backward on the GPU, then what the reference prints: every k-mer whose posterior at the MIDDLE event
// is >= 0.1, highest first, as "kmer<TAB>posterior".  log Pr(data) goes to stderr; -o dumps the matrices as
// "event state alpha beta" lines.
//
//   run-fwbw -p model.tsv (-s transitions.tsv | --pr-skip P --pr-stay Q) -e events.tsv [-o matrices.tsv]
#include <cmath>
#include <fstream>
#include <iostream>
#include <set>

#include "nanocall_amd/nanocall_amd.hpp"
#include "text_formats.hpp"

using namespace nanocall_amd;
typedef Pore_Model<float, 6> Pore_Model_Type;
typedef State_Transitions<float, 6> State_Transitions_Type;
typedef Event_Sequence<float, 6> Event_Sequence_Type;
typedef Forward_Backward<float, 6> Forward_Backward_Type;

int main(int argc, char* argv[])
{
    std::string pm_fn, st_fn, ev_fn, out_fn;
    float pr_skip = .3f, pr_stay = .1f;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string a = argv[i], v = argv[i + 1];
        if (a == "-p") pm_fn = v; else if (a == "-s") st_fn = v; else if (a == "-e") ev_fn = v; else if (a == "-o") out_fn = v;
        else if (a == "--pr-skip") pr_skip = std::stof(v); else if (a == "--pr-stay") pr_stay = std::stof(v);
        else { std::cerr << "unknown option " << a << std::endl; return 2; }
    }
    if (pm_fn.empty() || ev_fn.empty()) { std::cerr << "usage: run-fwbw -p model -e events [-s transitions | --pr-skip P --pr-stay Q] [-o out]\n"; return 2; }
    try {
        Pore_Model_Type pm;
        pm.load_from_vector(text_formats::read_model_table(pm_fn));
        State_Transitions_Type st;
        Event_Sequence_Type ev = text_formats::read_events(ev_fn);
        Forward_Backward_Type fwbw;
        if (st_fn.empty()) {
            st.compute_transitions_fast(pr_skip, pr_stay);
            fwbw.fill(pm, st, ev);
        } else {
            text_formats::put_transitions_file(st_fn, 1);
            fwbw.fill_with_slot(pm, ev, 1);         // transitions already sit in device slot 1
        }
        // all kmers with posterior >= .1 for the middle event, highest first (run-fwbw.cpp:71-88)
        std::multiset<std::pair<float, unsigned>> s;
        for (unsigned j = 0; j < Forward_Backward_Type::n_states; ++j) {
            const float v = std::exp(fwbw.log_posterior((unsigned)(ev.size() / 2), j));
            if (v >= .1f) s.insert(std::make_pair(v, j));
        }
        for (auto it = s.rbegin(); it != s.rend(); ++it) std::cout << Kmer<6>::to_string(it->second) << '\t' << it->first << std::endl;
        std::cerr << "log_pr_data " << fwbw.log_pr_data() << std::endl;
        if (!out_fn.empty()) {
            std::ofstream os(out_fn);
            os.precision(9);
            for (unsigned i = 0; i < ev.size(); ++i)
                for (unsigned j = 0; j < Forward_Backward_Type::n_states; ++j)
                    os << i << '\t' << j << '\t' << fwbw.cell(i, j).alpha << '\t' << fwbw.cell(i, j).beta << '\n';
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
