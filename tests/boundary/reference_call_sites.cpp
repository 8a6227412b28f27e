// reference_call_sites.cpp -- BOUNDARY PROOF (test program): the two places where nanocall's driver enters the HMM
// core, written against include/nanocall_amd/nanocall_amd.hpp exactly as the reference writes them against its own
// headers -- same types, same member calls, same argument lists:
//
//   * the 2D training round loop            src/nanocall/nanocall.cpp:360-426   (Parameter_Trainer::train_one_round)
//   * the basecall_strand functor           src/nanocall/nanocall.cpp:645-690   (Pore_Model::scale, State_Transitions::
//                                            compute_transitions_fast, apply_drift_correction, Viterbi::fill, path_probability)
//
// If this file compiles and its output matches the oracle (tests/test_cpp_layer_gpu.py), a maintainer who swaps the
// reference headers for this one keeps those call sites as they are.  Only what the reference takes from elsewhere is
// stubbed here: `opts::` values, LOG, and a read_summary holding the members those lines touch.
//
//   reference_call_sites <template.events> <complement.events> <model_0> <model_1> <num_events> <max_rounds> <train_drift>
//   (event files: "mean stdv start length" per line, Event operator>>, Event.hpp:59-68)
#include <array>
#include <fstream>
#include <iostream>
#include <string>
#include <tuple>

#include "nanocall_amd/nanocall_amd.hpp"

using namespace std;
using namespace nanocall_amd;

#define FLOAT_TYPE float
#define KMER_SIZE 6
typedef State_Transitions<FLOAT_TYPE, KMER_SIZE> State_Transitions_Type;
typedef State_Transition_Parameters<FLOAT_TYPE> State_Transition_Parameters_Type;
typedef Pore_Model<FLOAT_TYPE, KMER_SIZE> Pore_Model_Type;
typedef Pore_Model_Dict<FLOAT_TYPE, KMER_SIZE> Pore_Model_Dict_Type;
typedef Pore_Model_Parameters<FLOAT_TYPE> Pore_Model_Parameters_Type;
typedef Event<FLOAT_TYPE, KMER_SIZE> Event_Type;
typedef Event_Sequence<FLOAT_TYPE, KMER_SIZE> Event_Sequence_Type;
typedef Parameter_Trainer<FLOAT_TYPE, KMER_SIZE> Parameter_Trainer_Type;
typedef Viterbi<FLOAT_TYPE, KMER_SIZE> Viterbi_Type;

#define LOG(...) if (true) {} else std::clog
enum { debug, info, warning };

namespace opts {
bool no_train_scaling = false, no_train_transitions = false;
unsigned scaling_max_rounds = 10, scaling_num_events = 200;
float scaling_min_progress = 1.0;
}  // namespace opts

// the members of Fast5_Summary the two call sites use
struct Read_Summary {
    string read_id = "read";
    array<Event_Sequence_Type, 2> ev;
    map<array<string, 2>, Pore_Model_Parameters_Type> pm_params_m;
    map<array<string, 2>, array<State_Transition_Parameters_Type, 2>> st_params_m;
    const Event_Sequence_Type& events(unsigned st) const { return ev[st]; }
};

static void print_hex(const char* tag, const Pore_Model_Parameters_Type& p, const array<State_Transition_Parameters_Type, 2>& s)
{
    cout << tag << hexfloat << " " << p.scale << " " << p.shift << " " << p.drift << " " << p.var << " " << p.scale_sd << " " << p.var_sd << " "
         << s[0].p_stay << " " << s[0].p_skip << " " << s[1].p_stay << " " << s[1].p_skip << defaultfloat;
}

int main(int argc, char* argv[])
{
    if (argc != 8) { cerr << "usage: reference_call_sites ev0 ev1 model0 model1 num_events max_rounds train_drift" << endl; return 2; }
    Read_Summary read_summary;
    for (unsigned st = 0; st < 2; ++st) {
        ifstream is(argv[1 + st]);
        Event_Type e;
        while (is >> e) read_summary.ev[st].push_back(e);
    }
    const string m_name_0 = argv[3], m_name_1 = argv[4];
    opts::scaling_num_events = (unsigned)atoi(argv[5]);
    opts::scaling_max_rounds = (unsigned)atoi(argv[6]);
    Parameter_Trainer_Type::pm_train_drift() = (unsigned)atoi(argv[7]);
    State_Transition_Parameters_Type::default_p_stay() = .1f;   // nanocall.cpp:923-924 with the option defaults :84-85
    State_Transition_Parameters_Type::default_p_skip() = .3f;

    // init_models, nanocall.cpp:157-170
    Pore_Model_Dict_Type models;
    for (unsigned i = 0; i < Builtin_Model::num(); ++i) {
        Pore_Model_Type pm;
        pm.load_from_vector(Builtin_Model::init_lists(i));
        pm.strand() = Builtin_Model::strands(i);
        models[Builtin_Model::names(i)] = move(pm);
    }
    // init_transitions, :189
    State_Transitions_Type default_transitions;
    default_transitions.compute_transitions_fast(.3f, .1f);
    Parameter_Trainer_Type::init();   // :280

    // train_event_seqs, :327-338
    array<vector<Event_Sequence_Type>, 2> train_event_seqs;
    for (unsigned st = 0; st < 2; ++st) {
        unsigned num_train_events = min((size_t)opts::scaling_num_events, read_summary.events(st).size());
        train_event_seqs[st].emplace_back(read_summary.events(st).begin(), read_summary.events(st).begin() + num_train_events / 2);
        train_event_seqs[st].emplace_back(read_summary.events(st).end() - num_train_events / 2, read_summary.events(st).end());
    }
    vector<pair<const Event_Sequence_Type*, unsigned>> train_event_seq_ptrs;   // :345-352
    for (unsigned st = 0; st < 2; ++st)
        for (const auto& events : train_event_seqs[st]) train_event_seq_ptrs.push_back(make_pair(&events, st));
    array<string, 2> m_name_key = {{m_name_0, m_name_1}};
    read_summary.pm_params_m[m_name_key] = Pore_Model_Parameters_Type();
    read_summary.st_params_m[m_name_key][0] = State_Transition_Parameters_Type();
    read_summary.st_params_m[m_name_key][1] = State_Transition_Parameters_Type();
    map<array<string, 2>, FLOAT_TYPE> model_fit;

    // ---------------- the 2D round loop, nanocall.cpp:360-426 ----------------
    {
        string m_name = m_name_0 + "+" + m_name_1;
        unsigned round = 0;
        auto& crt_pm_params = read_summary.pm_params_m.at(m_name_key);
        auto& crt_st_params = read_summary.st_params_m.at(m_name_key);
        auto& crt_fit = model_fit[m_name_key];
        crt_fit = -INFINITY;
        while (true) {
            Pore_Model_Parameters_Type old_pm_params(crt_pm_params);
            std::array<State_Transition_Parameters_Type, 2> old_st_params(crt_st_params);
            auto old_fit = crt_fit;
            bool done;

            Parameter_Trainer_Type::train_one_round(
                train_event_seq_ptrs,
                {{&models.at(m_name_0), &models.at(m_name_1)}},
                default_transitions,
                old_pm_params, old_st_params,
                crt_pm_params, crt_st_params, crt_fit, done,
                not opts::no_train_scaling, not opts::no_train_transitions);

            LOG(debug)
                << "scaling_round read [" << read_summary.read_id << "] strand [" << 2 << "] model [" << m_name
                << "] old_pm_params [" << old_pm_params << "] old_st_params [" << old_st_params[0] << "," << old_st_params[1]
                << "] old_fit [" << old_fit << "] crt_pm_params [" << crt_pm_params
                << "] crt_st_params [" << crt_st_params[0] << "," << crt_st_params[1]
                << "] crt_fit [" << crt_fit << "] round [" << round << "]" << endl;
            print_hex("round", crt_pm_params, crt_st_params);
            cout << " " << hexfloat << crt_fit << defaultfloat << " " << done << endl;

            if (done) {
                // singularity detected; stop
                break;
            }

            if (crt_fit < old_fit) {
                crt_pm_params = old_pm_params;
                crt_st_params = old_st_params;
                crt_fit = old_fit;
                break;
            }

            ++round;
            // stop condition
            if (round >= 2u * opts::scaling_max_rounds or (round > 1 and crt_fit < old_fit + opts::scaling_min_progress)) {
                break;
            }
        };   // while true
        print_hex("result", crt_pm_params, crt_st_params);
        cout << " " << hexfloat << crt_fit << defaultfloat << " " << round << endl;
    }

    // ---------------- basecall_strand, nanocall.cpp:645-690 ----------------
    auto basecall_strand = [&](unsigned st, string m_name, const Pore_Model_Parameters_Type& pm_params,
                               const State_Transition_Parameters_Type& st_params) {
        // scale model
        Pore_Model_Type pm(models.at(m_name));
        pm.scale(pm_params);
        State_Transitions_Type custom_transitions;
        const State_Transitions_Type* transitions_ptr;
        if (not st_params.is_default()) {
            custom_transitions.compute_transitions_fast(st_params);
            transitions_ptr = &custom_transitions;
        } else {
            transitions_ptr = &default_transitions;
        }
        LOG(info) << "basecalling read [" << read_summary.read_id << "] strand [" << st << "] model [" << m_name << "] pm_params ["
                  << pm_params << "] st_params [" << st_params << "]" << endl;
        LOG(debug) << "mean_stdv read [" << read_summary.read_id << "] strand [" << st << "] model_mean [" << pm.mean()
                   << "] model_stdv [" << pm.stdv() << "]" << endl;
        // correct drift
        Event_Sequence_Type corrected_events = read_summary.events(st);
        corrected_events.apply_drift_correction(pm_params.drift);
        Viterbi_Type vit;
        vit.fill(pm, *transitions_ptr, corrected_events);
        return std::make_tuple(vit.path_probability(), std::move(corrected_events));
    };
    for (unsigned st = 0; st < 2; ++st) {   // :718-724
        auto r = basecall_strand(st, m_name_key[st], read_summary.pm_params_m.at(m_name_key), read_summary.st_params_m.at(m_name_key)[st]);
        cout << "strand " << st << " " << hexfloat << get<0>(r) << defaultfloat << " " << get<1>(r).get_base_seq() << endl;
    }
    return 0;
}
