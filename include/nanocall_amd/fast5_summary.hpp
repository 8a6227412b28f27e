// fast5_summary.hpp -- counterpart of the reference's Fast5_Summary (src/nanocall/Fast5_Summary.hpp): per-read
// summary (strand bounds, abasic level, initial scaling parameters per candidate model), event loading, and the
// --stats TSV row.  Same member names and meaning; the arithmetic lives in the library (nchmm_read_summarize,
// nchmm_read_load_events, nchmm_initial_scaling, nchmm_mean_stdv) and the file access in nanocall_fast5.h.
//
// Inputs: FAST5 files (HDF5, through the library's run-time HDF5 binding) and, additionally, the text form of an
// EventDetection table ("#nanocall-events" header, see read_events_table) for machines or tests without HDF5.
// --write-fast5 (add_basecall_*, Fast5_Summary.hpp:379-437) is not provided.
#ifndef NANOCALL_AMD_FAST5_SUMMARY_HPP
#define NANOCALL_AMD_FAST5_SUMMARY_HPP

#include <atomic>
#include <fstream>
#include <set>

#include "nanocall_amd/nanocall_amd.hpp"
#include "nanocall_fast5.h"

namespace nanocall_amd {

// ---- the EventDetection table of one read, from either source -------------------------------------------------
struct Ed_Table {
    bool have_sampling_rate = false, have_events = false;
    double sampling_rate = 0;
    std::string read_id;
    std::vector<nchmm_ed_event> events;
};

// "#nanocall-events" text table: '#key value' header lines (sampling_rate, read_id), then "mean stdv start length"
// per event with start / length in samples -- the columns of the FAST5 EventDetection dataset.
inline bool is_events_table(const std::string& fn)
{
    std::ifstream is(fn);
    std::string line;
    return is && std::getline(is, line) && line.compare(0, 16, "#nanocall-events") == 0;
}

inline Ed_Table read_events_table(const std::string& fn)
{
    Ed_Table t;
    std::ifstream is(fn);
    if (!is) throw Error(NCHMM_E_IO, fn.c_str());
    std::string line;
    while (std::getline(is, line)) {
        if (line.empty()) continue;
        if (line[0] == '#') {
            std::istringstream ls(line.substr(1));
            std::string key;
            ls >> key;
            if (key == "sampling_rate") { ls >> t.sampling_rate; t.have_sampling_rate = true; }
            else if (key == "read_id") ls >> t.read_id;
            continue;
        }
        std::istringstream ls(line);
        nchmm_ed_event e;
        if (ls >> e.mean >> e.stdv >> e.start >> e.length) t.events.push_back(e);
    }
    t.have_events = true;
    return t;
}

inline Ed_Table read_ed_table(const std::string& fn, const std::string& ed_group)
{
    // (a *.fast5 name is tried as HDF5 straight away: every extra open of a file costs as much as reading its table)
    const bool named_fast5 = fn.size() >= 6 && fn.compare(fn.size() - 6, 6, ".fast5") == 0;
    if (!(named_fast5 && nchmm_fast5_available()) && is_events_table(fn)) return read_events_table(fn);
    nchmm_fast5_read r;
    const int rc = nchmm_fast5_load(fn.c_str(), ed_group.c_str(), &r);
    if (rc != NCHMM_OK) throw Error(rc, nchmm_fast5_last_error());   // the reference's hdf5_tools::Exception
    Ed_Table t;
    t.have_sampling_rate = r.have_sampling_rate != 0;
    t.have_events = r.have_events != 0;
    t.sampling_rate = r.sampling_rate;
    t.read_id = r.read_id;
    t.events.assign(r.events, r.events + r.n_events);
    nchmm_fast5_release(&r);
    return t;
}

// fast5::File::is_valid_file as the driver uses it (nanocall.cpp:212,225,247), extended to the text form
inline bool is_valid_read_file(const std::string& fn) { return nchmm_fast5_is_valid_file(fn.c_str()) != 0 || is_events_table(fn); }

// ---- Fast5_Summary ---------------------------------------------------------------------------------------------
template <typename Float_Type = float, unsigned Kmer_Size = 6>
class Fast5_Summary {
public:
    typedef Pore_Model<Float_Type, Kmer_Size> Pore_Model_Type;
    typedef Pore_Model_Dict<Float_Type, Kmer_Size> Pore_Model_Dict_Type;
    typedef Pore_Model_Parameters<Float_Type> Pore_Model_Parameters_Type;
    typedef Event<Float_Type, Kmer_Size> Event_Type;
    typedef Event_Sequence<Float_Type, Kmer_Size> Event_Sequence_Type;
    typedef State_Transition_Parameters<Float_Type> State_Transition_Parameters_Type;

    std::string file_name, base_file_name, read_id;
    std::array<std::array<std::string, 2>, 3> preferred_model;
    std::map<std::array<std::string, 2>, Pore_Model_Parameters_Type> pm_params_m;
    std::map<std::array<std::string, 2>, std::array<State_Transition_Parameters_Type, 2>> st_params_m;
    std::array<unsigned, 4> strand_bounds{{0, 0, 0, 0}};
    std::array<Float_Type, 2> time_length{{0, 0}};
    unsigned num_ed_events = 0;
    Float_Type sampling_rate = 0;
    Float_Type abasic_level = 0;
    bool valid = false;
    bool scale_strands_together = false;

    std::unique_ptr<std::vector<nchmm_ed_event>> ed_events_ptr;
    std::array<std::unique_ptr<Event_Sequence_Type>, 2> events_ptr;

    const Event_Sequence_Type& events(unsigned st) const { return *events_ptr.at(st); }
    Event_Sequence_Type& events(unsigned st) { return *events_ptr.at(st); }

    // option singletons, Fast5_Summary.hpp:74-132 (pushed from the command line at nanocall.cpp:925-964)
    static unsigned& min_ed_events() { static unsigned v = 10; return v; }
    static unsigned& max_ed_events() { static unsigned v = 100000; return v; }
    static std::string& eventdetection_group() { static std::string v = "000"; return v; }
    static double& abasic_level_top_percent() { static double v = 1.0; return v; }
    static double& abasic_level_top_offset() { static double v = 0.0; return v; }
    static unsigned& hairpin_island_window_size() { static unsigned v = 10; return v; }   // (read by find_hairpin_islands only,
    static unsigned& hairpin_island_window_load() { static unsigned v = 5; return v; }    //  which detect_strands does not call, :661)
    static unsigned& template_only() { static unsigned v = 0; return v; }
    static std::array<unsigned, 4>& trim_margins() { static std::array<unsigned, 4> v = {{50u, 50u, 50u, 50u}}; return v; }
    // The reference drops the EventDetection table after summarize() and reads the file again in load_events()
    // (Fast5_Summary.hpp:317-318,329-347) to bound memory.  Opening and reading a FAST5 costs ~0.7 ms and HDF5 serialises
    // its calls, so a driver that is about to process the reads may let summarize() keep up to this many bytes of tables
    // (0 = the reference's behaviour); load_events() then uses the kept table and releases it.
    static size_t& ed_cache_budget() { static size_t v = 0; return v; }
    static std::atomic<size_t>& ed_cache_bytes() { static std::atomic<size_t> v{0}; return v; }

    Fast5_Summary() = default;
    Fast5_Summary(const std::string fn, const Pore_Model_Dict_Type& models, bool sst) { summarize(fn, models, sst); }

    static nchmm_segment_opts segment_opts()
    {
        nchmm_segment_opts o;
        o.min_ed_events = min_ed_events(); o.max_ed_events = max_ed_events();
        o.abasic_level_top_percent = abasic_level_top_percent(); o.abasic_level_top_offset = abasic_level_top_offset();
        o.template_only = template_only();
        for (int k = 0; k < 4; ++k) o.trim_margins[k] = trim_margins()[(size_t)k];
        return o;
    }

    // Fast5_Summary.hpp:138-319.  `preloaded`: the file's EventDetection table when the caller has read it already (a driver
    // that reads the files on one thread -- HDF5 serialises anyway -- and summarises on many); it is consumed.
    void summarize(const std::string& fn, const Pore_Model_Dict_Type& models, bool sst, Ed_Table* preloaded = nullptr)
    {
        valid = true;
        file_name = fn;
        const auto pos = file_name.find_last_of('/');
        base_file_name = pos != std::string::npos ? file_name.substr(pos + 1) : file_name;
        if (base_file_name.size() >= 6 && base_file_name.substr(base_file_name.size() - 6) == ".fast5")
            base_file_name.resize(base_file_name.size() - 6);
        read_id = base_file_name;
        strand_bounds = {{0, 0, 0, 0}};
        time_length = {{0, 0}};
        num_ed_events = 0;
        abasic_level = 0;
        try {
            Ed_Table t = preloaded ? std::move(*preloaded) : read_ed_table(file_name, eventdetection_group());
            do {
                if (!t.have_sampling_rate) { log_info(file_name + ": missing sampling rate"); break; }
                sampling_rate = static_cast<Float_Type>(t.sampling_rate);
                if (sampling_rate < 1000.0 || sampling_rate > 10000.0) { log_info(file_name + ": unexpected sampling rate"); break; }
                if (!t.have_events) { log_info(file_name + ": missing eventdetection events"); break; }
                if (!t.read_id.empty()) read_id = t.read_id;
                nchmm_read_summary s;
                const nchmm_segment_opts o = segment_opts();
                check(nchmm_read_summarize(&o, t.events.size(), t.events.data(), sampling_rate, sst ? 1 : 0, &s), "nchmm_read_summarize");
                abasic_level = s.abasic_level;
                for (int k = 0; k < 4; ++k) strand_bounds[(size_t)k] = s.strand_bounds[k];
                num_ed_events = s.num_ed_events;
                if (num_ed_events == 0) { log_info(file_name + ": read skipped (too few events, abasic level too low or no template strand)"); break; }
                scale_strands_together = s.scale_strands_together != 0;
                time_length = {{s.time_length[0], s.time_length[1]}};
                // initial model scalings, :223-278
                t.events.resize(num_ed_events);
                ed_events_ptr.reset(new std::vector<nchmm_ed_event>(std::move(t.events)));
                // The reference loads the strands' Event objects here (load_events(), three logs per event) only to take the mean and
                // stdv of their levels (:223-234) and drops them again (:317).  The same numbers come from the filtered level means
                // alone -- nchmm_read_load_events is what load_events() fills its Events from -- at a third of the summary pass's cost.
                std::array<std::array<float, 2>, 2> r{};
                std::array<size_t, 2> n_events{{0, 0}};
                for (unsigned st = 0; st < 2; ++st) {
                    const size_t cap = strand_bounds[2 * st + 1] > strand_bounds[2 * st] ? strand_bounds[2 * st + 1] - strand_bounds[2 * st] : 0;
                    if (cap == 0) continue;
                    std::vector<float> buf(4 * cap);
                    check(nchmm_read_load_events(&s, ed_events_ptr->data(), sampling_rate, (int)st, buf.data(), buf.data() + cap, buf.data() + 2 * cap,
                                                 buf.data() + 3 * cap, &n_events[st]), "nchmm_read_load_events");
                    if (n_events[st] < min_ed_events()) continue;
                    check(nchmm_mean_stdv(n_events[st], buf.data(), &r[st][0], &r[st][1]), "nchmm_mean_stdv");
                }
                if (scale_strands_together) {
                    for (const auto& p0 : models) {
                        if (!(p0.second.strand() == 0 || p0.second.strand() == 2)) continue;
                        for (const auto& p1 : models) {
                            if (!(p1.second.strand() == 1 || p1.second.strand() == 2)) continue;
                            const std::array<std::string, 2> m_name = {{p0.first, p1.first}};
                            const float m0[2] = {p0.second.mean(), p0.second.stdv()}, m1[2] = {p1.second.mean(), p1.second.stdv()};
                            Pore_Model_Parameters_Type pm;
                            check(nchmm_initial_scaling(1, r[0].data(), r[1].data(), m0, m1, &pm.scale, &pm.shift), "nchmm_initial_scaling");
                            pm_params_m[m_name] = pm;
                            st_params_m[m_name][0] = State_Transition_Parameters_Type();
                            st_params_m[m_name][1] = State_Transition_Parameters_Type();
                        }
                    }
                } else {
                    for (unsigned st = 0; st < 2; ++st) {
                        if (n_events[st] < min_ed_events()) continue;
                        for (const auto& p : models) {
                            if (!(p.second.strand() == st || p.second.strand() == 2)) continue;
                            std::array<std::string, 2> m_name;
                            m_name[st] = p.first;
                            const float m[2] = {p.second.mean(), p.second.stdv()};
                            Pore_Model_Parameters_Type pm;
                            check(nchmm_initial_scaling(0, r[st].data(), nullptr, m, nullptr, &pm.scale, &pm.shift), "nchmm_initial_scaling");
                            pm_params_m[m_name] = pm;
                            st_params_m[m_name][st] = State_Transition_Parameters_Type();
                        }
                    }
                }
            } while (false);
        } catch (const Error& e) {   // :311-315
            std::clog << "warning: " << file_name << ": HDF5 error: " << e.what() << std::endl;
            num_ed_events = 0;
        }
        drop_events();
        if (ed_events_ptr && num_ed_events > 0) {
            const size_t bytes = ed_events_ptr->size() * sizeof(nchmm_ed_event);
            if (ed_cache_bytes().fetch_add(bytes) + bytes <= ed_cache_budget()) {
                _ed_cached = true;                       // load_events() takes it from here
                return;
            }
            ed_cache_bytes().fetch_sub(bytes);
        }
        ed_events_ptr.reset();
    }

    // Fast5_Summary.hpp:321-370
    void load_events()
    {
        drop_events();
        if (num_ed_events == 0) return;
        const bool must_load = !ed_events_ptr;
        const bool release_cached = _ed_cached;
        _ed_cached = false;
        if (must_load) {
            Ed_Table t = read_ed_table(file_name, eventdetection_group());
            t.events.resize(std::min<size_t>(t.events.size(), num_ed_events));
            ed_events_ptr.reset(new std::vector<nchmm_ed_event>(std::move(t.events)));
        }
        nchmm_read_summary s;
        s.num_ed_events = num_ed_events; s.abasic_level = abasic_level; s.scale_strands_together = scale_strands_together ? 1 : 0;
        for (int k = 0; k < 4; ++k) s.strand_bounds[k] = strand_bounds[(size_t)k];
        s.time_length[0] = time_length[0]; s.time_length[1] = time_length[1];
        for (unsigned st = 0; st < 2; ++st) {
            events_ptr[st].reset(new Event_Sequence_Type());
            const size_t cap = strand_bounds[2 * st + 1] > strand_bounds[2 * st] ? strand_bounds[2 * st + 1] - strand_bounds[2 * st] : 0;
            if (cap == 0) continue;
            std::vector<float> buf(4 * cap);
            size_t n = 0;
            check(nchmm_read_load_events(&s, ed_events_ptr->data(), sampling_rate, (int)st, buf.data(), buf.data() + cap, buf.data() + 2 * cap,
                                         buf.data() + 3 * cap, &n), "nchmm_read_load_events");
            events(st).resize(n);
            for (size_t i = 0; i < n; ++i) {
                Event_Type& e = events(st)[i];
                e.mean = buf[i]; e.corrected_mean = e.mean; e.stdv = buf[cap + i]; e.start = buf[2 * cap + i]; e.length = buf[3 * cap + i];
                e.update_logs();
            }
        }
        if (must_load || release_cached) {
            if (release_cached) ed_cache_bytes().fetch_sub(ed_events_ptr->size() * sizeof(nchmm_ed_event));
            ed_events_ptr.reset();
        }
    }
    void drop_events()
    {
        for (unsigned st = 0; st < 2; ++st) events_ptr[st].reset();
    }

    friend std::ostream& operator<<(std::ostream& os, const Fast5_Summary& fs)   // :439-458
    {
        os << "[base_file_name=" << fs.base_file_name << " valid=" << fs.valid;
        if (fs.valid) {
            os << " num_ed_events=" << fs.num_ed_events;
            if (fs.num_ed_events > 0)
                os << " read_id=" << fs.read_id << " abasic_level=" << fs.abasic_level << " strand_bounds=[" << fs.strand_bounds[0] << ","
                   << fs.strand_bounds[1] << "," << fs.strand_bounds[2] << "," << fs.strand_bounds[3] << "] time_length=[" << fs.time_length[0]
                   << "," << fs.time_length[1] << "]";
        }
        os << "]";
        return os;
    }

    static void write_tsv_header(std::ostream& os)   // :460-477
    {
        os << "file_name" << "\tread_name" << "\tnum_ed_events" << "\tabasic_level" << "\ttemplate_start_idx" << "\ttemplate_end_idx"
           << "\tcomplement_start_idx" << "\tcomplement_end_idx";
        for (unsigned st = 0; st < 2; ++st)
            os << "\tn" << st << "_model_name" << "\tn" << st << "_scale" << "\tn" << st << "_shift" << "\tn" << st << "_drift" << "\tn" << st
               << "_var" << "\tn" << st << "_scale_sd" << "\tn" << st << "_var_sd" << "\tn" << st << "_p_stay" << "\tn" << st << "_p_skip";
    }

    void write_tsv(std::ostream& os) const   // :479-502
    {
        os << base_file_name << '\t' << read_id << '\t' << num_ed_events << '\t' << abasic_level << '\t' << strand_bounds[0] << '\t'
           << strand_bounds[1] << '\t' << strand_bounds[2] << '\t' << strand_bounds[3];
        for (unsigned st = 0; st < 2; ++st) {
            os << '\t';
            if (!preferred_model[st][st].empty()) {
                os << preferred_model[st][st] << '\t';
                pm_params_m.at(preferred_model[st]).write_tsv(os);
                os << '\t';
                st_params_m.at(preferred_model[st])[st].write_tsv(os);
            } else {
                os << ".\t";
                Pore_Model_Parameters_Type().write_tsv(os);
                os << '\t';
                State_Transition_Parameters_Type().write_tsv(os);
            }
        }
    }

    static bool& verbose() { static bool v = false; return v; }
private:
    bool _ed_cached = false;
    static void log_info(const std::string& msg)
    {
        if (verbose()) std::clog << "info: " << msg << std::endl;
    }
};

}  // namespace nanocall_amd
#endif
