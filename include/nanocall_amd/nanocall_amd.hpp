// nanocall_amd.hpp -- C++ host layer over the C ABI (nanocall_hip.h): drop-in counterparts of the
// reference's hot-path classes, same names, same member functions, same argument meaning.
//
//   reference (src/nanocall/)                      here (namespace nanocall_amd)
//   Kmer<6>                    Kmer.hpp            Kmer<6>
//   Event / Event_Sequence     Event.hpp           Event / Event_Sequence
//   Pore_Model_Parameters,
//   Pore_Model_State, Pore_Model   Pore_Model.hpp  same (load_from_vector, scale, state(i), ...)
//   State_Transition_Parameters,
//   State_Transitions          State_Transitions.hpp   same (compute_transitions_fast, neighbours(i).from_v)
//   Viterbi                    Viterbi.hpp         Viterbi::fill(pm, st, ev) / path_probability()
//   Forward_Backward           Forward_Backward.hpp    fill(pm, st, ev) / cell(i,j) / log_posterior / log_pr_data
//   Parameter_Trainer          Parameter_Trainer.hpp   init() / train_one_round(...)
//
// The DP itself runs on the GPU through libnanocall_hip.so; everything that goes through libm (model
// logs, transition weights, event logs) is computed by the library's host functions so that it is
// bit-identical to the reference.  A nanocall maintainer switches by including this header instead
// of the reference headers and adding `using namespace nanocall_amd;`.
//
// Throughput note: the reference calls fill() once per strand from a pfor worker thread.  That works here
// as it stands: Viterbi::fill hands the strand to nchmm_viterbi_strand, which combines the calls that are
// in progress on all threads into batched launches (about half the calling threads' strands per launch) --
// give the pfor as many worker threads as strands should be in flight.  Viterbi::fill_batch (many strands,
// one model) and Parameter_Trainer::train_one_round (which batches its 2-4 windows) are the explicit forms.
#ifndef NANOCALL_AMD_HPP
#define NANOCALL_AMD_HPP

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "nanocall_hip.h"

namespace nanocall_amd {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char* where) : std::runtime_error(std::string(where) + ": " + nchmm_strerror(c)), code(c) {}
};
inline void check(int rc, const char* where) { if (rc != NCHMM_OK) throw Error(rc, where); }

// One device context per host thread (the reference has one DP object per pfor worker,
// nanocall.cpp:611-621).  Slots 0..63 are managed by the classes below.
class Device {
public:
    static Device& instance()
    {
        static thread_local Device d;
        return d;
    }
    // the GPU this thread's context lives on: per thread, so that a host with one worker thread per device (the
    // read-parallel pfor of nanocall.cpp:282,611, one GPU per worker) can drive every GPU of the node.  Set it before
    // the thread's first ctx() call.
    static int& device_id() { static thread_local int id = 0; return id; }
    nchmm_ctx* ctx()
    {
        if (!_ctx) check(nchmm_create(&_ctx, device_id()), "nchmm_create");
        return _ctx;
    }
    ~Device() { if (_ctx) nchmm_destroy(_ctx); }
    // One context per GPU shared by every thread of the process, used ONLY through nchmm_viterbi_strand (which is thread-safe and
    // combines the strands of concurrent callers into batched launches): what Viterbi::fill runs on.
    static nchmm_ctx* shared_ctx(int dev)
    {
        static std::mutex m;
        static std::vector<Device*> per_dev;      // (kept until exit: worker threads may still be inside a call at any time)
        std::lock_guard<std::mutex> g(m);
        if (dev < 0) throw Error(NCHMM_E_INVALID, "Device::shared_ctx");
        if ((size_t)dev >= per_dev.size()) per_dev.resize((size_t)dev + 1, nullptr);
        if (!per_dev[dev]) {
            Device* d = new Device();
            check(nchmm_create(&d->_ctx, dev), "nchmm_create");
            per_dev[dev] = d;
        }
        return per_dev[dev]->_ctx;
    }
private:
    nchmm_ctx* _ctx = nullptr;
};

// ---------------------------------------------------------------------------------------------
// Kmer (Kmer.hpp)
// ---------------------------------------------------------------------------------------------
template <unsigned Kmer_Size = 6>
class Kmer {
    static_assert(Kmer_Size == 6, "the pore HMM is a 6-mer model (Kmer.hpp:119 hard-codes 4096 too)");
public:
    static const unsigned n_states = 1u << (2 * Kmer_Size);
    static size_t to_int(const std::string& s)
    {
        size_t r = 0;
        for (char c : s) { r <<= 2; r += (c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : size_t(-1)); }
        return r;
    }
    static std::string to_string(size_t k)
    {
        std::string r(Kmer_Size, 'A');
        for (unsigned j = 0; j < Kmer_Size; ++j) r[j] = "ACGT"[(k >> (2 * (Kmer_Size - j - 1))) & 3];
        return r;
    }
    static unsigned prefix(unsigned i, unsigned k) { return i >> (2 * (Kmer_Size - k)); }
    static unsigned suffix(unsigned i, unsigned k) { return i & ((1u << (2 * k)) - 1); }
    static unsigned min_skip(unsigned k1, unsigned k2)
    {
        if (k1 == k2) return 0;
        for (unsigned d = 1; d < Kmer_Size; ++d)
            if (suffix(k1, Kmer_Size - d) == prefix(k2, Kmer_Size - d)) return d;
        return Kmer_Size;
    }
    static unsigned max_self_overlap(unsigned i)
    {
        for (unsigned k = Kmer_Size - 1; k >= 1; --k)
            if (suffix(i, k) == prefix(i, k)) return k;
        return 0;
    }
    static std::vector<unsigned> neighbour_list(unsigned i, unsigned d)
    {
        std::vector<unsigned> v;
        for (unsigned b1 = 0; b1 < 4; ++b1) {
            unsigned i1 = (suffix(i, Kmer_Size - 1) << 2) + b1;
            if (d == 1) { v.push_back(i1); continue; }
            for (unsigned b2 = 0; b2 < 4; ++b2) v.push_back((suffix(i1, Kmer_Size - 1) << 2) + b2);
        }
        return v;
    }
};

// ---------------------------------------------------------------------------------------------
// Event / Event_Sequence (Event.hpp)
// ---------------------------------------------------------------------------------------------
template <typename Float_Type = float, unsigned Kmer_Size = 6>
class Event {
public:
    Float_Type mean = 0, corrected_mean = 0, stdv = 0, start = 0, length = 0;
    Float_Type log_mean = 0, log_corrected_mean = 0, log_stdv = 0;
    Float_Type orig_mean = 0, p_model_state = 0;
    std::array<char, Kmer_Size> model_state{};
    unsigned model_state_idx = 0;
    int move = 0;
    void update_logs()   // Event.hpp:35-45
    {
        log_mean = std::log(mean);
        log_corrected_mean = std::log(corrected_mean);
        if (stdv == 0.0) stdv = static_cast<Float_Type>(0.01);
        log_stdv = std::log(stdv);
    }
    void set_model_state(const std::string& s) { std::copy_n(s.begin(), Kmer_Size, model_state.begin()); }
    friend std::ostream& operator<<(std::ostream& os, const Event& ev)   // Event.hpp:51-58
    {
        os << ev.mean << '\t' << ev.stdv << '\t' << ev.start << '\t' << ev.length;
        return os;
    }
    friend std::istream& operator>>(std::istream& is, Event& ev)   // Event.hpp:59-68
    {
        is >> ev.mean >> ev.stdv >> ev.start >> ev.length;
        ev.corrected_mean = ev.mean;
        ev.update_logs();
        return is;
    }
};

template <typename Float_Type = float, unsigned Kmer_Size = 6>
struct Event_Sequence : std::vector<Event<Float_Type, Kmer_Size>> {
    typedef std::vector<Event<Float_Type, Kmer_Size>> Base;
    using Base::Base;
    void apply_drift_correction(Float_Type drift)   // Event.hpp:77-84
    {
        for (auto& e : *this) {
            e.corrected_mean -= drift * e.start;
            e.log_corrected_mean = std::log(e.corrected_mean);
        }
    }
    std::string get_base_seq() const   // Event.hpp:85-99
    {
        const Base& v = *this;
        if (v.empty()) return std::string();
        std::string res(v[0].model_state.begin(), v[0].model_state.end());
        for (size_t i = 1; i < v.size(); ++i) {
            unsigned a = std::min<unsigned>((unsigned)v[i].move, Kmer_Size), b = Kmer_Size - a;
            res.append(v[i].model_state.begin() + b, v[i].model_state.end());
        }
        return res;
    }
};

// ---------------------------------------------------------------------------------------------
// Pore_Model (Pore_Model.hpp)
// ---------------------------------------------------------------------------------------------
template <typename Float_Type = float>
struct Pore_Model_Parameters {
    Float_Type scale = 1, shift = 0, drift = 0, var = 1, scale_sd = 1, var_sd = 1;
    friend std::ostream& operator<<(std::ostream& os, const Pore_Model_Parameters& p)   // Pore_Model.hpp:66-71
    {
        os << "[scale=" << p.scale << " shift=" << p.shift << " drift=" << p.drift << " var=" << p.var << " scale_sd=" << p.scale_sd
           << " var_sd=" << p.var_sd << "]";
        return os;
    }
    void write_tsv(std::ostream& os) const   // Pore_Model.hpp:72-76 (the manipulators stay set on the stream, as there)
    {
        os << std::fixed << std::setprecision(5) << scale << '\t' << shift << '\t' << drift << '\t' << var << '\t' << scale_sd << '\t'
           << var_sd;
    }
};

template <typename Float_Type = float, unsigned Kmer_Size = 6>
struct Pore_Model_State {   // field order is the library's S x 10 layout (= Pore_Model.hpp:85-96)
    Float_Type level_mean, level_stdv, sd_mean, sd_stdv, sd_lambda;
    Float_Type log_level_mean, log_level_stdv, log_sd_mean, log_sd_stdv, log_sd_lambda;
};

template <typename Float_Type = float, unsigned Kmer_Size = 6>
class Pore_Model {
    static_assert(std::is_same<Float_Type, float>::value, "FLOAT_TYPE is float (nanocall.cpp:33-35)");
public:
    typedef Pore_Model_State<Float_Type, Kmer_Size> Pore_Model_State_Type;
    typedef Pore_Model_Parameters<Float_Type> Pore_Model_Parameters_Type;
    typedef Event<Float_Type, Kmer_Size> Event_Type;
    static const unsigned n_states = 1u << (2 * Kmer_Size);

    Pore_Model() : _strand(2) {}
    void clear() { _state.clear(); _origin.reset(); _n_scale = 0; }
    const Pore_Model_State_Type& state(unsigned i) const { return _state.at(i); }
    Pore_Model_State_Type& state(unsigned i) { _origin.reset(); return _state.at(i); }   // (edited by hand: no longer "a loaded model scaled once")
    const std::vector<Pore_Model_State_Type>& get_state_vector() const { return _state; }   // Pore_Model.hpp:183
    // the library's S x 10 layout (nchmm_put_models_scaled, nchmm_train_reads, ... take tables in this form)
    const float* data() const { return reinterpret_cast<const float*>(_state.data()); }
    const unsigned& strand() const { return _strand; }
    unsigned& strand() { return _strand; }
    Float_Type mean() const { return _mean; }
    Float_Type stdv() const { return _stdv; }

    template <typename V_Float_Type>
    void load_from_vector(const std::vector<V_Float_Type>& v)   // Pore_Model.hpp:220-239
    {
        if (v.size() != n_states * 4) throw Error(NCHMM_E_INVALID, "Pore_Model::load_from_vector");
        std::vector<float> t(v.begin(), v.end());
        _state.resize(n_states);
        check(nchmm_model_load(t.data(), reinterpret_cast<float*>(_state.data())), "nchmm_model_load");
        update_statistics();
        // the loaded states, shared by every copy of this model: a copy that is then scaled once -- `pm = models.at(name);
        // pm.scale(pm_params)`, nanocall.cpp:653-657 -- can tell the library "this table, these parameters" (provenance())
        _origin = std::make_shared<const std::vector<Pore_Model_State_Type>>(_state);
        _n_scale = 0;
    }
    void scale(const Pore_Model_Parameters_Type& p)   // Pore_Model.hpp:190-201
    {
        const float par[6] = {p.scale, p.shift, p.drift, p.var, p.scale_sd, p.var_sd};
        check(nchmm_model_scale(reinterpret_cast<float*>(_state.data()), par), "nchmm_model_scale");
        update_statistics();
        if (_n_scale++ == 0) std::memcpy(_scaled_by, par, sizeof(par));
    }
    // (unscaled S x 10 states, parameters) when this model is a loaded model scaled at most once; false otherwise
    bool provenance(const float** base, float p6[6]) const
    {
        if (!_origin || _n_scale > 1 || _origin->size() != _state.size()) return false;
        *base = reinterpret_cast<const float*>(_origin->data());
        const float ident[6] = {1, 0, 0, 1, 1, 1};
        std::memcpy(p6, _n_scale ? _scaled_by : ident, sizeof(ident));
        return true;
    }
    // Pore_Model.hpp:295-299 (host evaluation, for callers outside the DP such as debug dumps)
    Float_Type log_pr_corrected_emission(unsigned i, const Event_Type& e) const
    {
        const auto& s = state(i);
        static const Float_Type log_2pi = std::log(2.0 * M_PI);
        Float_Type a = (e.corrected_mean - s.level_mean) / s.level_stdv;
        Float_Type n = -s.log_level_stdv - (log_2pi + a * a) / static_cast<Float_Type>(2.0);
        Float_Type b = (e.stdv - s.sd_mean) / s.sd_mean;
        Float_Type ig = (s.log_sd_lambda - log_2pi - static_cast<Float_Type>(3.0) * e.log_stdv - s.sd_lambda * b * b / e.stdv)
                        / static_cast<Float_Type>(2.0);
        return n + ig;
    }
    // Pore_Model.hpp:241-249: one line per state, "kmer level_mean level_stdv sd_mean sd_stdv"
    friend std::ostream& operator<<(std::ostream& os, const Pore_Model& pm)
    {
        for (unsigned i = 0; i < pm._state.size(); ++i) {
            const auto& s = pm._state[i];
            os << Kmer<Kmer_Size>::to_string(i) << '\t' << s.level_mean << '\t' << s.level_stdv << '\t' << s.sd_mean << '\t' << s.sd_stdv
               << std::endl;
        }
        return os;
    }
    // Pore_Model.hpp:251-287: rows in any order ('#' and header lines skipped), sorted by k-mer; anything but 4096
    // distinct k-mers is an error (the reference logs and exits; this throws)
    friend std::istream& operator>>(std::istream& is, Pore_Model& pm)
    {
        std::vector<float> table(n_states * 4, 0.f);
        std::vector<char> seen(n_states, 0);
        unsigned n = 0;
        std::string line;
        while (std::getline(is, line)) {
            std::istringstream iss(line);
            std::string k;
            iss >> k;
            if (k.empty() || k[0] == '#') continue;
            if (line.find("kmer") != std::string::npos) continue;
            const size_t j = k.size() == Kmer_Size ? Kmer<Kmer_Size>::to_int(k) : n_states;
            if (j >= n_states || seen[j]) throw Error(NCHMM_E_INVALID, "Pore_Model >>: unexpected k-mer");
            seen[j] = 1;
            iss >> table[4 * j] >> table[4 * j + 1] >> table[4 * j + 2] >> table[4 * j + 3];
            ++n;
        }
        if (n != n_states) throw Error(NCHMM_E_INVALID, "Pore_Model >>: unexpected number of states");
        pm.load_from_vector(table);
        return is;
    }
    // the S x 10 states as the library lays them out (nchmm_model_load)
    const float* states_Sx10() const { return reinterpret_cast<const float*>(_state.data()); }
    // the library's S x 6 table of this model
    void pack6(float* t6) const { check(nchmm_model_pack6(reinterpret_cast<const float*>(_state.data()), t6), "nchmm_model_pack6"); }
    // upload as the library's S x 6 table into `slot`
    void put(int slot) const
    {
        std::vector<float> t6(n_states * 6);
        pack6(t6.data());
        check(nchmm_put_model(Device::instance().ctx(), slot, t6.data()), "nchmm_put_model");
    }
private:
    std::vector<Pore_Model_State_Type> _state;
    std::shared_ptr<const std::vector<Pore_Model_State_Type>> _origin;
    float _scaled_by[6] = {1, 0, 0, 1, 1, 1};
    unsigned _n_scale = 0;
    Float_Type _mean = 0, _stdv = 0;
    unsigned _strand;
    void update_statistics()   // Pore_Model.hpp:307-313; alg::mean_stdv_of is hpptools (absent): nchmm_mean_stdv states what is assumed
    {
        std::vector<float> lv(_state.size());
        for (size_t i = 0; i < _state.size(); ++i) lv[i] = _state[i].level_mean;
        check(nchmm_mean_stdv(lv.size(), lv.data(), &_mean, &_stdv), "nchmm_mean_stdv");
    }
};

// Builtin_Model (Builtin_Model.hpp:7-13): the same four members, served by the library
struct Builtin_Model {
    static unsigned num() { return (unsigned)nchmm_builtin_count(); }
    static std::string names(unsigned i) { const char* n = nchmm_builtin_name((int)i); return n ? n : ""; }
    static unsigned strands(unsigned i) { return (unsigned)nchmm_builtin_strand((int)i); }
    static std::vector<float> init_lists(unsigned i)
    {
        const float* t = nchmm_builtin_table((int)i);
        return t ? std::vector<float>(t, t + 4096 * 4) : std::vector<float>();
    }
};
template <typename Float_Type, unsigned Kmer_Size>
using Pore_Model_Dict = std::map<std::string, Pore_Model<Float_Type, Kmer_Size>>;

// ---------------------------------------------------------------------------------------------
// State_Transitions (State_Transitions.hpp)
// ---------------------------------------------------------------------------------------------
template <typename Float_Type = float>
struct State_Transition_Parameters {
    Float_Type p_stay, p_skip;
    static Float_Type& default_p_stay() { static Float_Type v = .09; return v; }
    static Float_Type& default_p_skip() { static Float_Type v = .28; return v; }
    State_Transition_Parameters() : p_stay(default_p_stay()), p_skip(default_p_skip()) {}
    bool is_default() const { return p_stay == default_p_stay() && p_skip == default_p_skip(); }
    friend std::ostream& operator<<(std::ostream& os, const State_Transition_Parameters& stp)   // State_Transitions.hpp:39-44
    {
        os << "[p_stay=" << stp.p_stay << " p_skip=" << stp.p_skip << "]";
        return os;
    }
    void write_tsv(std::ostream& os) const   // :45-50
    {
        os << std::fixed << std::setprecision(5) << p_stay << '\t' << p_skip;
    }
};

template <typename Float_Type = float>
struct State_Neighbours {
    std::vector<std::pair<unsigned, Float_Type>> from_v, to_v;
};

template <typename Float_Type = float, unsigned Kmer_Size = 6>
class State_Transitions {
public:
    typedef State_Neighbours<Float_Type> State_Neighbours_Type;
    typedef State_Transition_Parameters<Float_Type> State_Transition_Parameters_Type;
    static const unsigned n_states = 1u << (2 * Kmer_Size);
    void clear() { _row_ptr.clear(); _pred.clear(); _logw.clear(); _nb.clear(); }
    bool empty() const { return _row_ptr.empty(); }

    void compute_transitions_fast(Float_Type p_skip_default, Float_Type p_stay)   // State_Transitions.hpp:181-224
    {
        _row_ptr.resize(n_states + 1); _pred.resize(NCHMM_MAX_ARCS); _logw.resize(NCHMM_MAX_ARCS);
        uint32_t n = 0;
        check(nchmm_transitions_fast(p_skip_default, p_stay, _row_ptr.data(), _pred.data(), _logw.data(), &n),
              "nchmm_transitions_fast");
        _pred.resize(n); _logw.resize(n); _nb.clear();
        _p_skip = p_skip_default; _p_stay = p_stay;
    }
    void compute_transitions_fast(const State_Transition_Parameters_Type& stp) { compute_transitions_fast(stp.p_skip, stp.p_stay); }

    // neighbours(i).from_v / .to_v, built on first use
    const State_Neighbours_Type& neighbours(unsigned i) const
    {
        if (_nb.empty()) {
            _nb.resize(n_states);
            for (unsigned j = 0; j < n_states; ++j)
                for (uint32_t a = _row_ptr[j]; a < _row_ptr[j + 1]; ++a) {
                    _nb[j].from_v.emplace_back(_pred[a], _logw[a]);
                    _nb[_pred[a]].to_v.emplace_back(j, _logw[a]);
                }
        }
        return _nb.at(i);
    }
    void put(int slot) const
    {
        check(nchmm_put_transitions(Device::instance().ctx(), slot, _row_ptr.data(), _pred.data(), _logw.data()),
              "nchmm_put_transitions");
    }
    // State_Transitions.hpp:226-236: "kmer_i kmer_j log_p" per arc, by source state
    friend std::ostream& operator<<(std::ostream& os, const State_Transitions& st)
    {
        for (unsigned i = 0; i < n_states && !st.empty(); ++i)
            for (const auto& p : st.neighbours(i).to_v)
                os << Kmer<Kmer_Size>::to_string(i) << '\t' << Kmer<Kmer_Size>::to_string(p.first) << '\t' << p.second << std::endl;
        return os;
    }
    // State_Transitions.hpp:237-252 (+ update_fields :79-104: from_v by ascending source).  The arcs are kept as read;
    // the device accepts them (put) only if they form the stay/step/skip-1 graph (NCHMM_E_TOPOLOGY otherwise).
    friend std::istream& operator>>(std::istream& is, State_Transitions& st)
    {
        std::vector<std::tuple<unsigned, unsigned, Float_Type>> arcs;   // (destination, source, log p)
        std::string ki, kj;
        Float_Type p;
        while (is >> ki >> kj >> p) arcs.emplace_back((unsigned)Kmer<Kmer_Size>::to_int(kj), (unsigned)Kmer<Kmer_Size>::to_int(ki), p);
        std::stable_sort(arcs.begin(), arcs.end(), [](const std::tuple<unsigned, unsigned, Float_Type>& a,
                                                      const std::tuple<unsigned, unsigned, Float_Type>& b) {
            return std::get<0>(a) != std::get<0>(b) ? std::get<0>(a) < std::get<0>(b) : std::get<1>(a) < std::get<1>(b);
        });
        st._row_ptr.assign(n_states + 1, 0); st._pred.clear(); st._logw.clear(); st._nb.clear();
        for (const auto& a : arcs) {
            if (std::get<0>(a) >= n_states || std::get<1>(a) >= n_states) throw Error(NCHMM_E_INVALID, "State_Transitions >>: bad k-mer");
            st._row_ptr[std::get<0>(a) + 1]++;
            st._pred.push_back((uint16_t)std::get<1>(a));
            st._logw.push_back(std::get<2>(a));
        }
        for (unsigned j = 0; j < n_states; ++j) st._row_ptr[j + 1] += st._row_ptr[j];
        return is;
    }
    // the (p_skip, p_stay) this table was computed from by compute_transitions_fast (NaN after operator>>)
    Float_Type p_skip() const { return _p_skip; }
    Float_Type p_stay() const { return _p_stay; }
private:
    Float_Type _p_skip = std::numeric_limits<Float_Type>::quiet_NaN(), _p_stay = std::numeric_limits<Float_Type>::quiet_NaN();
    std::vector<uint32_t> _row_ptr;
    std::vector<uint16_t> _pred;
    std::vector<float> _logw;
    mutable std::vector<State_Neighbours_Type> _nb;
};

namespace detail {
template <typename ES>
inline void soa(const ES& ev, std::vector<float>& cm, std::vector<float>& sd, std::vector<float>& ls)
{
    for (const auto& e : ev) { cm.push_back(e.corrected_mean); sd.push_back(e.stdv); ls.push_back(e.log_stdv); }
}
// f(begin, end) over [0, n) on the host cores: the AoS <-> SoA loops around a batch launch are per-strand independent
// (the reference runs them inside its pfor workers, nanocall.cpp:611-621)
template <typename F>
inline void parallel_for(size_t n, F&& f)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt ? std::min<unsigned>(nt, 32) : 4;
    if (n < 4 || nt < 2) { f((size_t)0, n); return; }
    nt = (unsigned)std::min<size_t>(nt, n);
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; ++i) th.emplace_back([&, i] { f(n * i / nt, n * (i + 1) / nt); });
    for (auto& t : th) t.join();
}
}  // namespace detail

// ---------------------------------------------------------------------------------------------
// Viterbi (Viterbi.hpp)
// ---------------------------------------------------------------------------------------------
template <typename Float_Type = float, unsigned Kmer_Size = 6>
class Viterbi {
public:
    typedef Kmer<Kmer_Size> Kmer_Type;
    typedef Pore_Model<Float_Type, Kmer_Size> Pore_Model_Type;
    typedef State_Transitions<Float_Type, Kmer_Size> State_Transitions_Type;
    typedef Event_Sequence<Float_Type, Kmer_Size> Event_Sequence_Type;
    static const unsigned n_states = Pore_Model_Type::n_states;

    unsigned n_events() const { return _n_events; }
    Float_Type path_probability() const { return _path_probability; }

    // Viterbi.hpp:44-99: fills ev[i].model_state_idx / model_state / move.  One strand per call, as the reference's
    // basecall_strand makes it from every pfor worker thread (nanocall.cpp:645-690, :611-621): concurrent calls -- from any
    // number of threads, each with its own model and transitions -- are combined into batched launches on the GPU's shared
    // context (nchmm_viterbi_strand); the call returns when this strand is decoded.  Run the pfor with as many worker threads
    // as you want strands in flight (they sleep here): ~1000 fill the GPU.
    void fill(const Pore_Model_Type& pm, const State_Transitions_Type& st, Event_Sequence_Type& ev)
    {
        _n_events = (unsigned)ev.size();
        if (std::isnan(st.p_skip()) || std::isnan(st.p_stay())) {
            // transitions read from a file: no (p_skip, p_stay) to hand over -- the thread's own context, a launch to itself
            std::vector<Event_Sequence_Type*> evs{&ev};
            _path_probability = fill_batch(pm, st, evs)[0];
            return;
        }
        const size_t n = ev.size();
        std::vector<float> soa(3 * n);
        float* const cm = soa.data(); float* const sd = cm + n; float* const ls = sd + n;
        for (size_t i = 0; i < n; ++i) { cm[i] = ev[i].corrected_mean; sd[i] = ev[i].stdv; ls[i] = ev[i].log_stdv; }
        std::vector<uint16_t> s(n);
        float pp = 0;
        nchmm_ctx* const ctx = Device::shared_ctx(Device::device_id());
        const float* base = nullptr; float p6[6];
        int rc;
        if (pm.provenance(&base, p6)) {
            // a loaded model scaled once: the table and the parameters travel, Pore_Model::scale runs on the device
            rc = nchmm_viterbi_strand_scaled(ctx, base, p6, st.p_skip(), st.p_stay(), n, cm, sd, ls, s.data(), &pp);
        } else {
            std::vector<float> t6(n_states * 6);
            pm.pack6(t6.data());
            rc = nchmm_viterbi_strand(ctx, t6.data(), st.p_skip(), st.p_stay(), n, cm, sd, ls, s.data(), &pp);
        }
        if (rc != NCHMM_OK && rc != NCHMM_E_NUMERIC) check(rc, "nchmm_viterbi_strand");
        _path_probability = pp;
        if (rc == NCHMM_E_NUMERIC) return;   // reference has undefined behaviour here (Viterbi.hpp:125-141); leave events untouched
        for (size_t i = 0; i < n; ++i) {     // fill_state_seq / fill_move_seq write-back (Viterbi.hpp:134-150)
            const unsigned j = s[i];
            ev[i].model_state_idx = j;
            for (unsigned c = 0; c < Kmer_Size; ++c) ev[i].model_state[c] = "ACGT"[(j >> (2 * (Kmer_Size - 1 - c))) & 3u];   // Kmer::to_string
            ev[i].move = i > 0 ? (int)Kmer_Type::min_skip(s[i - 1], j) : 0;
        }
    }
    // many strands that share one scaled model and one transition table, one launch
    static std::vector<Float_Type> fill_batch(const Pore_Model_Type& pm, const State_Transitions_Type& st,
                                              const std::vector<Event_Sequence_Type*>& evs, int slot = 0)
    {
        pm.put(slot); st.put(slot);
        const size_t n = evs.size();
        std::vector<uint64_t> off(n + 1, 0);
        for (size_t r = 0; r < n; ++r) off[r + 1] = off[r] + evs[r]->size();
        const size_t total = (size_t)off[n];
        // SoA staging (the three fields the emission reads), filled per strand in parallel.  The buffers live with the
        // calling thread and only grow: no zero fill, no page-fault storm on every batch.
        struct Staging { std::unique_ptr<float[]> cm, sd, ls; std::unique_ptr<uint16_t[]> states; size_t cap = 0; };
        static thread_local Staging stg;
        if (stg.cap < total + 1) {
            stg.cap = total + 1 + total / 8;
            stg.cm.reset(new float[stg.cap]); stg.sd.reset(new float[stg.cap]); stg.ls.reset(new float[stg.cap]);
            stg.states.reset(new uint16_t[stg.cap]);
        }
        float* const cm = stg.cm.get(); float* const sd = stg.sd.get(); float* const ls = stg.ls.get();
        uint16_t* const states = stg.states.get();
        detail::parallel_for(n, [&](size_t lo, size_t hi) {
            for (size_t r = lo; r < hi; ++r) {
                size_t k = (size_t)off[r];
                for (const auto& e : *evs[r]) { cm[k] = e.corrected_mean; sd[k] = e.stdv; ls[k] = e.log_stdv; ++k; }
            }
        });
        std::vector<int32_t> slots(n, slot), status(n);
        std::vector<Float_Type> pp(n);
        int rc = nchmm_viterbi(Device::instance().ctx(), n, off.data(), cm, sd, ls, slots.data(), slots.data(), states, pp.data(),
                               status.data());
        if (rc != NCHMM_OK && rc != NCHMM_E_NUMERIC) check(rc, "nchmm_viterbi");
        // fill_state_seq / fill_move_seq write-back (Viterbi.hpp:134-150), per strand in parallel
        detail::parallel_for(n, [&](size_t lo, size_t hi) {
            for (size_t r = lo; r < hi; ++r) {
                auto& ev = *evs[r];
                if (status[r] != 0) continue;   // reference has undefined behaviour here (Viterbi.hpp:125-141); leave events untouched
                const uint16_t* s = states + off[r];
                for (size_t i = 0; i < ev.size(); ++i) {
                    const unsigned j = s[i];
                    ev[i].model_state_idx = j;
                    for (unsigned c = 0; c < Kmer_Size; ++c) ev[i].model_state[c] = "ACGT"[(j >> (2 * (Kmer_Size - 1 - c))) & 3u];   // Kmer::to_string
                    ev[i].move = i > 0 ? (int)Kmer_Type::min_skip(s[i - 1], j) : 0;
                }
            }
        });
        return pp;
    }
private:
    Float_Type _path_probability = 0;
    unsigned _n_events = 0;
};

// ---------------------------------------------------------------------------------------------
// Forward_Backward (Forward_Backward.hpp)
// ---------------------------------------------------------------------------------------------
template <typename Float_Type = float, unsigned Kmer_Size = 6>
class Forward_Backward {
public:
    typedef Pore_Model<Float_Type, Kmer_Size> Pore_Model_Type;
    typedef State_Transitions<Float_Type, Kmer_Size> State_Transitions_Type;
    typedef Event_Sequence<Float_Type, Kmer_Size> Event_Sequence_Type;
    struct Matrix_Entry { Float_Type alpha, beta; };
    static const unsigned n_states = Pore_Model_Type::n_states;

    void clear() { _alpha.clear(); _beta.clear(); }
    unsigned n_events() const { return (unsigned)(_alpha.size() / n_states); }
    Matrix_Entry cell(unsigned i, unsigned j) const { return Matrix_Entry{_alpha[(size_t)i * n_states + j], _beta[(size_t)i * n_states + j]}; }
    Float_Type log_posterior(unsigned i, unsigned j) const { return _alpha[(size_t)i * n_states + j] + _beta[(size_t)i * n_states + j] - _log_pr_data; }
    Float_Type log_pr_data() const { return _log_pr_data; }

    void fill(const Pore_Model_Type& pm, const State_Transitions_Type& st, const Event_Sequence_Type& ev, int slot = 0)
    {
        st.put(slot);
        fill_with_slot(pm, ev, slot);
    }
    // the transitions already sit in device slot `slot` (e.g. nchmm_put_transitions from a transitions file)
    void fill_with_slot(const Pore_Model_Type& pm, const Event_Sequence_Type& ev, int slot)
    {
        pm.put(slot);
        std::vector<float> cm, sd, ls;
        detail::soa(ev, cm, sd, ls);
        const uint64_t off[2] = {0, cm.size()};
        const int32_t s = slot;
        _alpha.assign(cm.size() * n_states, 0); _beta.assign(cm.size() * n_states, 0);
        check(nchmm_fwbw(Device::instance().ctx(), 1, off, cm.data(), sd.data(), ls.data(), &s, nullptr, &s, nullptr, &_log_pr_data,
                         nullptr, nullptr, _alpha.data(), _beta.data()), "nchmm_fwbw");
    }
private:
    std::vector<Float_Type> _alpha, _beta;
    Float_Type _log_pr_data = 0;
};

// ---------------------------------------------------------------------------------------------
// Parameter_Trainer (Parameter_Trainer.hpp)
// ---------------------------------------------------------------------------------------------
template <typename Float_Type = float, unsigned Kmer_Size = 6>
struct Parameter_Trainer {
    typedef Pore_Model<Float_Type, Kmer_Size> Pore_Model_Type;
    typedef Pore_Model_Parameters<Float_Type> Pore_Model_Parameters_Type;
    typedef State_Transitions<Float_Type, Kmer_Size> State_Transitions_Type;
    typedef State_Transition_Parameters<Float_Type> State_Transition_Parameters_Type;
    typedef Event_Sequence<Float_Type, Kmer_Size> Event_Sequence_Type;

    static void init()   // Parameter_Trainer.hpp:30-57
    {
        std::vector<uint16_t> k(4096); uint32_t n = 0;
        check(nchmm_st_train_kmers(k.data(), &n), "nchmm_st_train_kmers");
        st_train_kmers().assign(k.begin(), k.begin() + n);
    }
    static std::vector<unsigned>& st_train_kmers() { static std::vector<unsigned> v; return v; }
    static unsigned& pm_train_drift() { static unsigned v = 1; return v; }

    // Parameter_Trainer.hpp:541-579.
    static void train_one_round(const std::vector<std::pair<const Event_Sequence_Type*, unsigned>>& event_seq_ptrs,
                                const std::array<const Pore_Model_Type*, 2>& model_ptrs,
                                const State_Transitions_Type& default_transitions,
                                const Pore_Model_Parameters_Type& crt_pm_params,
                                const std::array<State_Transition_Parameters_Type, 2>& crt_st_params,
                                Pore_Model_Parameters_Type& new_pm_params,
                                std::array<State_Transition_Parameters_Type, 2>& new_st_params, Float_Type& fit, bool& done,
                                bool train_scaling, bool train_transitions)
    {
        done = false;
        // fill_train_data :99-155.  One call per read, from every pfor worker (nanocall.cpp:282-579): the windows of the calls
        // that are in progress on all threads are combined into batched launches on the GPU's shared context (nchmm_fwbw_windows).
        bool have[2] = {false, false};
        for (const auto& p : event_seq_ptrs) have[p.second] = true;
        const float* unscaled[2] = {nullptr, nullptr};
        float tr_skip[2] = {0, 0}, tr_stay[2] = {0, 0};
        int model_of[2] = {-1, -1};
        size_t n_models = 0;
        bool parametric = true;
        State_Transitions_Type custom[2];
        for (unsigned st = 0; st < 2; ++st) {
            if (!have[st]) continue;
            const State_Transitions_Type* tr = &default_transitions;
            if (!crt_st_params[st].is_default()) { custom[st].compute_transitions_fast(crt_st_params[st]); tr = &custom[st]; }
            parametric = parametric && !std::isnan(tr->p_skip()) && !std::isnan(tr->p_stay());
            // (the library scales the model by crt_pm_params on the device: Pore_Model::scale, bit for bit)
            unscaled[n_models] = model_ptrs[st]->states_Sx10(); tr_skip[n_models] = tr->p_skip(); tr_stay[n_models] = tr->p_stay();
            model_of[st] = (int)n_models++;
        }
        const float pm6[6] = {crt_pm_params.scale, crt_pm_params.shift, crt_pm_params.drift, crt_pm_params.var, crt_pm_params.scale_sd,
                              crt_pm_params.var_sd};
        std::vector<uint64_t> off{0};
        std::vector<float> cm, sd, ls, mean, start;
        std::vector<int32_t> s_slot, w_model;
        std::vector<float> stp, w_pm;
        for (const auto& p : event_seq_ptrs) {
            Event_Sequence_Type corrected(*p.first);
            corrected.apply_drift_correction(crt_pm_params.drift);
            detail::soa(corrected, cm, sd, ls);
            for (const auto& e : *p.first) { mean.push_back(e.mean); start.push_back(e.start); }
            off.push_back(cm.size());
            s_slot.push_back(62 + (int)p.second);
            w_model.push_back(model_of[p.second]);
            for (float v : {crt_pm_params.scale, crt_pm_params.shift, crt_pm_params.drift, crt_pm_params.var, crt_pm_params.scale_sd,
                            crt_pm_params.var_sd})
                w_pm.push_back(v);
            stp.push_back(crt_st_params[p.second].p_stay); stp.push_back(crt_st_params[p.second].p_skip);
        }
        const size_t n_win = event_seq_ptrs.size(), total = cm.size();
        std::vector<float> lpd(n_win), pm_sums(6 * total), st_sums(3 * n_win);
        if (parametric) {
            check(nchmm_fwbw_windows(Device::shared_ctx(Device::device_id()), n_models, unscaled, pm6, tr_skip, tr_stay, n_win, off.data(), cm.data(),
                                     sd.data(), ls.data(), w_model.data(), stp.data(), lpd.data(), pm_sums.data(), st_sums.data()),
                  "nchmm_fwbw_windows");
        } else {
            // default transitions read from a file: no (p_skip, p_stay) to hand over -- the thread's own context.  Slots 62/63.
            nchmm_ctx* ctx = Device::instance().ctx();
            for (unsigned st = 0; st < 2; ++st) {
                if (!have[st]) continue;
                Pore_Model_Type scaled(*model_ptrs[st]);
                scaled.scale(crt_pm_params);
                scaled.put(62 + (int)st);
                (crt_st_params[st].is_default() ? default_transitions : custom[st]).put(62 + (int)st);
            }
            check(nchmm_fwbw(ctx, n_win, off.data(), cm.data(), sd.data(), ls.data(), s_slot.data(), w_pm.data(), s_slot.data(),
                             stp.data(), lpd.data(), pm_sums.data(), st_sums.data(), nullptr, nullptr), "nchmm_fwbw");
        }
        fit = 0;
        for (float v : lpd) fit += v;   // :154
        if (train_scaling) {
            const float crt[6] = {crt_pm_params.scale, crt_pm_params.shift, crt_pm_params.drift, crt_pm_params.var,
                                  crt_pm_params.scale_sd, crt_pm_params.var_sd};
            float np[6]; int d = 0;
            check(nchmm_train_pm_finish(total, pm_sums.data(), mean.data(), sd.data(), start.data(), (int)pm_train_drift(), crt,
                                        np, &d), "nchmm_train_pm_finish");
            new_pm_params.scale = np[0]; new_pm_params.shift = np[1]; new_pm_params.drift = np[2];
            new_pm_params.var = np[3]; new_pm_params.scale_sd = np[4]; new_pm_params.var_sd = np[5];
            if (d) { done = true; new_st_params = crt_st_params; return; }
        }
        if (train_transitions) {
            for (unsigned st = 0; st < 2; ++st) {
                std::vector<float> mine;
                for (size_t w = 0; w < n_win; ++w)
                    if (event_seq_ptrs[w].second == st) mine.insert(mine.end(), st_sums.begin() + 3 * w, st_sums.begin() + 3 * w + 3);
                check(nchmm_train_st_finish(mine.size() / 3, mine.data(), &new_st_params[st].p_stay, &new_st_params[st].p_skip),
                      "nchmm_train_st_finish");
            }
        }
    }
};

}  // namespace nanocall_amd
#endif
