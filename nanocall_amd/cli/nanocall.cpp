// nanocall.cpp -- the `nanocall` command line on the MI355X-native HMM core.
//
// Counterpart of the reference driver (src/nanocall/nanocall.cpp): same option table (:50-95), pore presets
// (:943-970), option checks (:995-1059), model / transition / file / read initialisation (:97-273), training,
// basecalling, FASTA (:584-591, names :764-768,837-842) and --stats TSV (:893-903).  What differs is HOW the two
// read loops run: the reference gives one read at a time to each of `-t` worker threads (pfor, :282,611); here the
// reads of a chunk are sharded over the GPUs of the node (one host thread + context per device, nchmm_pool_*) and every
// device decodes its shard in batched launches.  Output order is input order.
//
// Not provided (fails with a message): --write-fast5 (HDF5 write-back), -s/--trans (the device tables, and the
// transition statistics of the EM rounds, are built from (pr_skip, pr_stay)).
// Extra options: --gpus N (devices to use, default all), --chunk-events N (events decoded per batch and device).
//
// More than one GPU: ONE WORKER PROCESS PER GPU (fan_out below).  The reference's unit of parallelism is a pfor worker inside one
// process (:282,611); one process cannot feed eight MI355X (its host stages -- summaries, event loading and packing, FASTA -- keep up
// with two or three), so the input files are partitioned over N children forked BEFORE anything in this process or in a child has
// touched the HIP runtime; each child runs the whole pipeline on its share with its own reader processes, host threads and one
// device, and streams its records, tagged with the input index, up a pipe; the parent writes them in input order (:859-861).
// The only exchange between the workers is the counter reduction: one RCCL all-reduce across the processes (ncclCommInitRank, the
// unique id relayed over the pipes), with the host sum as the fall-back.  --single-process keeps every GPU in this process.
#include <dirent.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <list>
#include <mutex>
#include <set>
#include <sstream>
#include <string>

#include "nanocall_amd/fast5_summary.hpp"
#include "nanocall_amd/nanocall_amd.hpp"

using namespace nanocall_amd;

#ifndef NANOCALL_AMD_VERSION
#define NANOCALL_AMD_VERSION "0.7.4-amd"
#endif

typedef State_Transitions<float, 6> State_Transitions_Type;
typedef State_Transition_Parameters<float> State_Transition_Parameters_Type;
typedef Pore_Model<float, 6> Pore_Model_Type;
typedef Pore_Model_Dict<float, 6> Pore_Model_Dict_Type;
typedef Pore_Model_Parameters<float> Pore_Model_Parameters_Type;
typedef Event<float, 6> Event_Type;
typedef Event_Sequence<float, 6> Event_Sequence_Type;
typedef Fast5_Summary<float, 6> Fast5_Summary_Type;

// ---------------------------------------------------------------------------------------------------------------
// logging: LOG(level) << ... as the reference uses it (hpptools logger.hpp is un-vendored; only the level filter and
// the message bodies are reproduced)
// ---------------------------------------------------------------------------------------------------------------
namespace logger {
enum level { error = 0, warning, info, debug, debug1, debug2 };
inline int& threshold() { static int t = info; return t; }
inline std::mutex& mutex() { static std::mutex m; return m; }
inline const char* name(int l) { static const char* n[] = {"error", "warning", "info", "debug", "debug1", "debug2"}; return n[std::min(l, 5)]; }
inline std::string& tag() { static std::string t; return t; }      // "[w3] " in worker process 3 of a multi-GPU run
struct Line {
    std::ostringstream os;
    int lvl;
    explicit Line(int l) : lvl(l) { os << "= nanocall " << tag() << name(l) << ": "; }
    // (one write per line: the worker processes of a multi-GPU run share this stream)
    ~Line() { std::lock_guard<std::mutex> g(mutex()); const std::string t = os.str(); std::clog.write(t.data(), (std::streamsize)t.size()); std::clog.flush(); }
};
inline int parse_level(const std::string& s)
{
    for (int l = 0; l <= debug2; ++l) if (s == name(l)) return l;
    try { return std::stoi(s); } catch (...) { return info; }
}
}  // namespace logger
#define LOG(l) if (logger::l > logger::threshold()) {} else logger::Line(logger::l).os

// ---------------------------------------------------------------------------------------------------------------
// options: the reference's TCLAP table (nanocall.cpp:50-95) on a small parser with TCLAP's conventions
// (`--name value`, `-x value`, switches, repeatable args, positional inputs, `--`, -h/--help, --version)
// ---------------------------------------------------------------------------------------------------------------
namespace opts {
struct Arg_Base;
inline std::vector<Arg_Base*>& registry() { static std::vector<Arg_Base*> r; return r; }
struct Arg_Base {
    std::string flag, name, desc, type_desc;
    bool is_switch = false, is_set = false;
    Arg_Base(const std::string& f, const std::string& n, const std::string& d, const std::string& t, bool sw)
        : flag(f), name(n), desc(d), type_desc(t), is_switch(sw) { registry().push_back(this); }
    virtual ~Arg_Base() {}
    virtual void assign(const std::string& v) = 0;
    bool isSet() const { return is_set; }
};
template <typename T> bool convert(const std::string& s, T& out)
{
    std::istringstream is(s);
    is >> out;
    return !is.fail() && is.eof();
}
template <> inline bool convert<std::string>(const std::string& s, std::string& out) { out = s; return true; }
template <typename T> struct ValueArg : Arg_Base {
    T value;
    ValueArg(const std::string& f, const std::string& n, const std::string& d, bool, T dflt, const std::string& t)
        : Arg_Base(f, n, d, t, false), value(dflt) {}
    void assign(const std::string& v) override
    {
        if (is_set) throw std::runtime_error("Argument already set: --" + name);
        if (!convert(v, value)) throw std::runtime_error("Couldn't read argument value from string '" + v + "' for --" + name);
        is_set = true;
    }
    T& get() { return value; }
    const T& get() const { return value; }
    operator const T&() const { return value; }
};
struct SwitchArg : Arg_Base {
    bool value = false;
    SwitchArg(const std::string& f, const std::string& n, const std::string& d) : Arg_Base(f, n, d, "", true) {}
    void assign(const std::string&) override { value = true; is_set = true; }
    bool get() const { return value; }
    void set(bool v) { value = v; }
    operator bool() const { return value; }
};
template <typename T> struct MultiArg : Arg_Base {
    std::vector<T> values;
    MultiArg(const std::string& f, const std::string& n, const std::string& d, bool, const std::string& t) : Arg_Base(f, n, d, t, false) {}
    void assign(const std::string& v) override
    {
        T x;
        if (!convert(v, x)) throw std::runtime_error("Couldn't read argument value from string '" + v + "' for --" + name);
        values.push_back(x);
        is_set = true;
    }
    const std::vector<T>& get() const { return values; }
    typename std::vector<T>::const_iterator begin() const { return values.begin(); }
    typename std::vector<T>::const_iterator end() const { return values.end(); }
};

std::string description = "Call bases in Oxford Nanopore reads.";
std::string program_name = "nanocall", orig_argv;
//
ValueArg<std::string> ed_group("", "ed-group", "EventDetection group to use. (default: smallest available)", false, "", "000|001|...");
ValueArg<unsigned> chunk_size("", "chunk-size", "Thread chunk size.", false, 1, "int");
MultiArg<std::string> log_level("", "log", "Log level. (default: info)", false, "string");
ValueArg<std::string> stats_fn("", "stats", "Stats.", false, "", "file");
ValueArg<std::string> train_drift("", "train-drift", "Train drift parameter. (default: yes for R73, no for R9)", false, "", "0|1");
ValueArg<unsigned> trim_ed_hp_end("", "trim-ed-hp-end", "Number of events to trim after hairpin end.", false, 50, "int");
ValueArg<unsigned> trim_ed_hp_start("", "trim-ed-hp-start", "Number of events to trim before hairpin start.", false, 50, "int");
ValueArg<unsigned> trim_ed_sq_end("", "trim-ed-sq-end", "Number of events to trim before sequence end.", false, 50, "int");
ValueArg<unsigned> trim_ed_sq_start("", "trim-ed-sq-start", "Number of events to trim after sequence start.", false, 50, "int");
ValueArg<unsigned> max_ed_events("", "max-ed-events", "Maximum EventDetection events.", false, 100000, "int");
ValueArg<unsigned> min_ed_events("", "min-ed-events", "Minimum EventDetection events.", false, 10, "int");
ValueArg<unsigned> fasta_line_width("", "fasta-line-width", "Maximum fasta line width.", false, 80, "int");
//
ValueArg<float> scaling_select_threshold("", "scaling-select-threshold", "Select best model per strand during scaling if log score better by threshold.", false, 20.0, "float");
ValueArg<float> scaling_min_progress("", "scaling-min-progress", "Minimum scaling fit progress.", false, 1.0, "float");
ValueArg<unsigned> scaling_max_rounds("", "scaling-max-rounds", "Maximum scaling rounds.", false, 10, "int");
ValueArg<unsigned> scaling_num_events("", "scaling-num-events", "Number of events used for model scaling.", false, 200, "int");
//
SwitchArg template_only("", "1d", "Interpret entire read as 1D template only.");
SwitchArg single_strand_scaling("", "single-strand-scaling", "Train scaling parameters per strand.");
SwitchArg double_strand_scaling("", "double-strand-scaling", "Train scaling parameters per read. (default)");
SwitchArg no_train_transitions("", "no-train-transitions", "Do not train state transitions.");
SwitchArg no_train_scaling("", "no-train-scaling", "Do not train pore model scaling.");
SwitchArg train("", "train", "Enable training. (default)");
SwitchArg no_train("", "no-train", "Disable all training.");
SwitchArg basecall("", "basecall", "Enable basecalling (default).");
SwitchArg no_basecall("", "no-basecall", "Disable basecalling.");
//
ValueArg<float> pr_skip("", "pr-skip", "Transition probability of skipping at least 1 state.", false, .3, "float");
ValueArg<float> pr_stay("", "pr-stay", "Transition probability of staying in the same state.", false, .1, "float");
ValueArg<std::string> trans_fn("s", "trans", "Custom initial state transitions.", false, "", "file");
ValueArg<std::string> model_fofn("", "model-fofn", "File of pore models.", false, "", "file");
MultiArg<std::string> model_fn("m", "model", "Custom pore model for strand (0=template, 1=complement, 2=both).", false, "strand:file");
//
ValueArg<std::string> pore("", "pore", "Pore name, used to select builtin pore model.", false, "r9", "r73|r9");
SwitchArg write_fast5("", "write-fast5", "Write basecalls to fast5 files.");
ValueArg<std::string> output_fn("o", "output", "Output.", false, "", "file");
ValueArg<unsigned> num_threads("t", "threads", "Number of parallel threads.", false, 1, "int");
// MI355X additions
ValueArg<int> gpus("", "gpus", "Number of GPUs to shard the reads over, one worker process each. (default: all visible)", false, 0, "int");
ValueArg<unsigned long> chunk_events("", "chunk-events", "Events decoded per batch and GPU.", false, 32000000ul, "int");
ValueArg<unsigned long> ed_cache_mb("", "ed-cache-mb", "Memory (MiB) in which event tables read by the summary pass are kept for the basecalling pass instead of re-reading the files.", false, 4096ul, "int");
ValueArg<int> reader_procs("", "reader-procs", "Processes that read the input files (HDF5 serialises its calls inside one process). 0: read in this process. (default: min(threads, 16))", false, -1, "int");
SwitchArg single_process("", "single-process", "Drive every GPU from this process (one host thread per GPU) instead of starting one worker process per GPU.");
SwitchArg serial_chunks("", "serial-chunks", "Take one chunk of reads at a time through event loading, the GPUs and FASTA writing instead of running the three side by side on consecutive chunks.");
ValueArg<std::string> dump_params_fn("", "dump-params", "Write the exact (hex float) parameters and path log-probability of every basecalled strand.", false, "", "file");
std::vector<std::string> input_fn;   // UnlabeledMultiArg "inputs"

void usage(std::ostream& os)
{
    os << "\nUSAGE:\n\n   " << program_name << "  [options] <path> ...\n\nWhere:\n\n";
    for (const Arg_Base* a : registry()) {
        os << "   ";
        if (!a->flag.empty()) os << "-" << a->flag << (a->is_switch ? "" : " <" + a->type_desc + ">") << ",  ";
        os << "--" << a->name << (a->is_switch ? "" : " <" + a->type_desc + ">") << "\n     " << a->desc << "\n\n";
    }
    os << "   <path>  (accepted multiple times)\n     (required)  Inputs: directories, fast5 files, or files of fast5 file names (use \"-\" to read fofn from stdin).\n\n"
       << "   " << description << "\n\n";
}

// returns false when the program should exit (code in *rc)
bool parse(int argc, char* argv[], int* rc)
{
    program_name = argv[0];
    for (int i = 0; i < argc; ++i) orig_argv += std::string(i ? " " : "") + argv[i];
    try {
        bool rest_positional = false;
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            if (rest_positional || a == "-" || a.empty() || a[0] != '-') { input_fn.push_back(a); continue; }
            if (a == "--") { rest_positional = true; continue; }
            if (a == "-h" || a == "--help") { usage(std::cout); *rc = EXIT_SUCCESS; return false; }
            if (a == "--version") { std::cout << "\n" << program_name << "  version: " << NANOCALL_AMD_VERSION << "\n\n"; *rc = EXIT_SUCCESS; return false; }
            Arg_Base* hit = nullptr;
            for (Arg_Base* r : registry())
                if ((a.size() > 2 && a[1] == '-' && a.substr(2) == r->name) || (a.size() == 2 && !r->flag.empty() && a.substr(1) == r->flag)) hit = r;
            if (!hit) throw std::runtime_error("Couldn't find match for argument: " + a);
            if (hit->is_switch) { hit->assign(""); continue; }
            if (i + 1 >= argc) throw std::runtime_error("Missing a value for this argument: " + a);
            hit->assign(argv[++i]);
        }
        if (input_fn.empty()) throw std::runtime_error("Required argument missing: inputs");
    } catch (const std::exception& e) {
        std::cerr << "PARSE ERROR: " << e.what() << "\n\nFor complete USAGE and HELP type: \n   " << program_name << " --help\n\n";
        *rc = EXIT_FAILURE;
        return false;
    }
    return true;
}
}  // namespace opts

// wall-clock per stage of the run, reported with the counters (the reference logs user CPU seconds of its two loops,
// nanocall.cpp:580-581,867-868)
struct Stage_Clock {
    std::mutex m;
    std::map<std::string, double> secs;
    std::vector<std::string> order;
    struct Scope {
        Stage_Clock& c; std::string name; std::chrono::steady_clock::time_point t0;
        Scope(Stage_Clock& cl, const std::string& n) : c(cl), name(n), t0(std::chrono::steady_clock::now()) {}
        ~Scope()
        {
            std::lock_guard<std::mutex> g(c.m);
            if (!c.secs.count(name)) c.order.push_back(name);
            c.secs[name] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    };
    std::string str() const
    {
        std::ostringstream os;
        for (const auto& n : order) os << " " << n << "=" << secs.at(n);
        return os.str();
    }
};
static Stage_Clock stage_clock;
#define STAGE(name) Stage_Clock::Scope stage_scope_##__LINE__(stage_clock, name)

// ---------------------------------------------------------------------------------------------------------------
// file-system helpers (fs_support.hpp:15-45)
// ---------------------------------------------------------------------------------------------------------------
static bool is_directory(const std::string& fn)
{
    struct stat st;
    return stat(fn.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
static std::vector<std::string> list_directory(const std::string& fn)
{
    std::vector<std::string> res;
    if (DIR* d = opendir(fn.c_str())) {
        while (struct dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (n != "." && n != "..") res.push_back(n);
        }
        closedir(d);
    }
    return res;   // readdir order, as the reference (compare FASTA per record, or feed a fofn)
}

// ---------------------------------------------------------------------------------------------------------------
// init_models / init_transitions / init_files / init_reads  (nanocall.cpp:97-273)
// ---------------------------------------------------------------------------------------------------------------
// A text file that may be gzip-compressed (the reference opens its model files and the model fofn through zstr,
// nanocall.cpp:122,144): zlib's gz layer passes plain files through unchanged.
static std::string read_text_or_gzip(const std::string& fn, const char* what)
{
    gzFile gz = gzopen(fn.c_str(), "rb");
    if (!gz) { LOG(error) << "cannot open " << what << " [" << fn << "]" << std::endl; std::exit(EXIT_FAILURE); }
    std::string text;
    char buf[1 << 16];
    int n;
    while ((n = gzread(gz, buf, sizeof(buf))) > 0) text.append(buf, (size_t)n);
    const int rc = gzclose(gz);
    if (n < 0 || rc != Z_OK) { LOG(error) << "cannot read " << what << " [" << fn << "]: damaged gzip stream" << std::endl; std::exit(EXIT_FAILURE); }
    return text;
}

// One "<strand>:<file>" model argument (strand 0, 1, or 2 = both) -- the reference's syntax and message (nanocall.cpp:99-109).
struct Model_Arg { unsigned strand; std::string file; };
static Model_Arg model_arg(const std::string& spec)
{
    const bool well_formed = spec.size() >= 3 && spec[1] == ':' && spec[0] >= '0' && spec[0] <= '2';
    if (!well_formed) {
        LOG(error) << "could not parse model name: \"" << spec << "\"; format should be \"[0|1|2]:<file>\"" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    return Model_Arg{(unsigned)(spec[0] - '0'), spec.substr(2)};
}

static void log_loaded(const char* kind, const std::string& name, unsigned strand, const Pore_Model_Type& pm)
{
    LOG(info) << "loaded " << kind << " [" << name << "] for strand [" << strand << "] statistics [mean=" << pm.mean() << ", stdv=" << pm.stdv() << "]" << std::endl;
}

// Which pore models the run uses (what init_models does, nanocall.cpp:97-176): the files named by --model / --model-fofn when
// there are any -- for both strands or for neither -- else the builtin tables of the chosen pore.
static void init_models(Pore_Model_Dict_Type& models)
{
    std::vector<Model_Arg> args;
    for (const std::string& spec : opts::model_fn.get()) args.push_back(model_arg(spec));
    if (!opts::model_fofn.get().empty()) {
        std::istringstream lines(read_text_or_gzip(opts::model_fofn.get(), "model fofn"));
        for (std::string spec; std::getline(lines, spec);) args.push_back(model_arg(spec));
    }
    if (args.empty()) {
        const std::string prefix = opts::pore.get() + ".";
        for (unsigned i = 0; i < Builtin_Model::num(); ++i) {
            const std::string name = Builtin_Model::names(i);
            if (name.compare(0, prefix.size(), prefix) != 0) continue;
            Pore_Model_Type& pm = models[name];
            pm.load_from_vector(Builtin_Model::init_lists(i));
            pm.strand() = Builtin_Model::strands(i);
            log_loaded("builtin module", name, pm.strand(), pm);
        }
        if (models.empty()) {
            LOG(error) << "no builtin models found for pore [" << opts::pore.get() << "]" << std::endl;
            std::exit(EXIT_FAILURE);
        }
        return;
    }
    size_t per_strand[3] = {0, 0, 0};
    for (const Model_Arg& a : args) ++per_strand[a.strand];
    if (per_strand[2] == 0 && (per_strand[0] == 0) != (per_strand[1] == 0)) {
        LOG(error) << "models were specified only for strand " << (int)(per_strand[0] == 0) << "! give models for both strands, or for neither." << std::endl;
        std::exit(EXIT_FAILURE);
    }
    // strand by strand, each strand's files in the order they were given (the reference's map of lists)
    for (unsigned strand = 0; strand < 3; ++strand)
        for (const Model_Arg& a : args) {
            if (a.strand != strand) continue;
            std::istringstream text(read_text_or_gzip(a.file, "model file"));
            Pore_Model_Type pm;
            try { text >> pm; } catch (const std::exception& x) { LOG(error) << a.file << ": " << x.what() << std::endl; std::exit(EXIT_FAILURE); }
            pm.strand() = strand;
            log_loaded("module", a.file, strand, pm);
            models[a.file] = std::move(pm);
        }
}

static void init_transitions(State_Transitions_Type& transitions)
{
    if (!opts::trans_fn.get().empty()) {
        LOG(error) << "custom initial state transitions (-s/--trans) are not supported by the GPU core: its transition tables are "
                      "built from (--pr-skip, --pr-stay)" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    transitions.compute_transitions_fast(opts::pr_skip, opts::pr_stay);
    LOG(info) << "init_state_transitions pr_skip=[" << opts::pr_skip.get() << "], pr_stay=[" << opts::pr_stay.get() << "]" << std::endl;
}

// is_valid_read_file for many paths.  One open per file, serialised by the HDF5 lock: 0.29 s for 8000 files in this process.
// From 256 candidates on the check is spread over forked children (the same reasoning, and the same moment of the run, as the
// reader processes below): child c answers for candidates c, c + K, ... with one byte each.
static std::vector<char> valid_read_files(const std::vector<std::string>& cand)
{
    std::vector<char> ok(cand.size(), 0);
    int k = opts::reader_procs.get();
    if (k < 0) k = (int)std::min<unsigned>(opts::num_threads, 16u);
    if (cand.size() < 256 || k < 2) {
        for (size_t i = 0; i < cand.size(); ++i) ok[i] = is_valid_read_file(cand[i]) ? 1 : 0;
        return ok;
    }
    struct Kid { pid_t pid; int fd; };
    std::vector<Kid> kids;
    for (int c = 0; c < k; ++c) {
        int fd[2];
        if (pipe(fd) != 0) break;
        const pid_t pid = fork();
        if (pid < 0) { close(fd[0]); close(fd[1]); break; }
        if (pid == 0) {
            signal(SIGPIPE, SIG_DFL);
            close(fd[0]);
            for (const Kid& o : kids) close(o.fd);
            const char* trap = std::getenv("NANOCALL_TEST_VALIDATE_ABORT");      // test hook: a file that takes its checker down
            for (size_t i = (size_t)c; i < cand.size(); i += (size_t)k) {
                if (trap && *trap && cand[i].find(trap) != std::string::npos) abort();
                const char b = is_valid_read_file(cand[i]) ? '1' : '0';
                ssize_t r;
                while ((r = write(fd[1], &b, 1)) < 0 && errno == EINTR) {}
                if (r != 1) _exit(1);
            }
            _exit(0);
        }
        close(fd[1]);
        kids.push_back(Kid{pid, fd[0]});
    }
    std::vector<char> have(cand.size(), 0);
    if ((int)kids.size() == k) {
        for (int c = 0; c < k; ++c) {
            for (size_t i = (size_t)c; i < cand.size(); i += (size_t)k) {
                char b = 0;
                ssize_t r;
                while ((r = read(kids[(size_t)c].fd, &b, 1)) < 0 && errno == EINTR) {}
                if (r != 1) break;                 // the child died (a file libhdf5 cannot survive): the rest is checked here
                ok[i] = b == '1'; have[i] = 1;
            }
        }
    }
    for (const Kid& o : kids) close(o.fd);
    for (const Kid& o : kids) { int st = 0; while (waitpid(o.pid, &st, 0) < 0 && errno == EINTR) {} }
    // whatever no child answered for: in this process -- except the first unanswered candidate of a child that died, which
    // is taken to be the file that killed it
    std::vector<char> suspect(cand.size(), 0);
    if ((int)kids.size() == k)
        for (int c = 0; c < k; ++c)
            for (size_t i = (size_t)c; i < cand.size(); i += (size_t)k)
                if (!have[i]) { suspect[i] = 1; break; }
    for (size_t i = 0; i < cand.size(); ++i) {
        if (have[i]) continue;
        if (suspect[i]) { LOG(warning) << cand[i] << ": the process checking this file died; file ignored" << std::endl; ok[i] = 0; }
        else ok[i] = is_valid_read_file(cand[i]) ? 1 : 0;
    }
    return ok;
}

static void init_files(std::list<std::string>& files)
{
    // explicit arguments first (their validity decides between "a read file" and "a file of file names")
    std::vector<std::string> args;
    for (const auto& f : opts::input_fn) if (f != "-" && !is_directory(f)) args.push_back(f);
    const std::vector<char> arg_ok = valid_read_files(args);
    size_t arg_at = 0;
    for (const auto& f : opts::input_fn) {
        if (is_directory(f)) {
            std::vector<std::string> cand;
            for (const auto& g : list_directory(f)) {
                const std::string f2 = f + (f[f.size() - 1] != '/' ? "/" : "") + g;
                if (is_directory(f2)) LOG(info) << "ignoring subdirectory [" << f2 << "]" << std::endl;
                else cand.push_back(f2);
            }
            const std::vector<char> ok = valid_read_files(cand);
            for (size_t i = 0; i < cand.size(); ++i) {
                if (ok[i]) { files.push_back(cand[i]); LOG(info) << "adding input file [" << cand[i] << "]" << std::endl; }
                else LOG(info) << "ignoring file [" << cand[i] << "]" << std::endl;
            }
        } else if (f != "-" && arg_ok[arg_at++]) {
            files.push_back(f);
            LOG(info) << "adding input file [" << f << "]" << std::endl;
        } else {
            LOG(info) << "interpreting [" << f << "] as fofn" << std::endl;
            std::ifstream ifs;
            std::istream* is_p = &std::cin;
            if (f != "-") {
                ifs.open(f);
                if (!ifs) { LOG(error) << "cannot open [" << f << "]" << std::endl; std::exit(EXIT_FAILURE); }
                is_p = &ifs;
            }
            std::vector<std::string> cand;
            std::string g;
            while (std::getline(*is_p, g)) cand.push_back(g);
            const std::vector<char> ok = valid_read_files(cand);
            for (size_t i = 0; i < cand.size(); ++i)
                if (ok[i]) { files.push_back(cand[i]); LOG(info) << "adding input file [" << cand[i] << "]" << std::endl; }
        }
    }
    if (files.empty()) {
        LOG(error) << "no fast5 files to process" << std::endl;
        std::exit(EXIT_FAILURE);
    }
}

// f(i) for i in [0, n) on `nt` host threads (the summaries / event loads of different reads are independent)
template <typename F> static void host_parallel(size_t n, unsigned nt, F&& f)
{
    nt = std::max(1u, std::min<unsigned>(nt, (unsigned)std::max<size_t>(n, 1)));
    if (nt == 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    // An exception leaving a std::thread body is std::terminate (a core dump with the GPUs held): keep the first one,
    // let the other workers run dry, rethrow on the caller's thread -- where main() turns it into LOG(error) + EXIT_FAILURE
    // as the reference does (nanocall.cpp: `LOG(error) << ...; exit(EXIT_FAILURE)`).
    std::atomic<size_t> next{0};
    std::exception_ptr first;
    std::mutex first_mutex;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&] {
            try {
                for (size_t i; (i = next++) < n;) f(i);
            } catch (...) {
                std::lock_guard<std::mutex> g(first_mutex);
                if (!first) first = std::current_exception();
                next = n;      // nothing more to hand out
            }
        });
    for (auto& t : th) t.join();
    if (first) std::rethrow_exception(first);
}

// ---------------------------------------------------------------------------------------------------------------
// Reader processes.  The HDF5 library takes one global lock per call, so inside one process the event tables of 8000
// files come in at 0.36 ms each however many threads ask (2.9 of the 4.1 s of the run recorded in
// profiles/r02_bench_cli.json).  Separate processes do not share that lock.  K children are forked right after the
// input list is known -- before this process creates a thread or touches the GPU, and they never do either --
// child c reads files c, c + K, c + 2K, ... in order and streams each table down its own pipe; the parent takes file i
// from pipe i mod K.  A full pipe blocks its child, which bounds the memory in flight.
// ---------------------------------------------------------------------------------------------------------------
class Reader_Procs {
public:
    ~Reader_Procs() { finish(); }
    size_t size() const { return kids_.size(); }

    void start(const std::vector<std::string>& files, const std::string& ed_group, unsigned k)
    {
        k = (unsigned)std::min<size_t>(k, files.size());
        for (unsigned c = 0; c < k; ++c) {
            int fd[2];
            if (pipe(fd) != 0) break;
#ifdef F_SETPIPE_SZ
            (void)fcntl(fd[1], F_SETPIPE_SZ, 1 << 20);
#endif
            const pid_t pid = fork();
            if (pid < 0) { close(fd[0]); close(fd[1]); break; }
            if (pid == 0) {
                signal(SIGPIPE, SIG_DFL);      // the parent going away ends the child at its next write
                close(fd[0]);
                for (const Kid& o : kids_) close(o.fd);
                // (the stride k is fixed up front: if a later fork fails the whole pool is torn down again)
                for (size_t i = c; i < files.size(); i += k) serve(fd[1], files[i], ed_group);
                _exit(0);                      // no destructors, no atexit handlers of the libraries mapped here
            }
            close(fd[1]);
            kids_.push_back(Kid{pid, fd[0], false});
        }
        if (kids_.size() != k) finish();       // could not fork them all: the caller reads in-process
    }

    // table of file i; calls for one child must come in increasing i (the summary pass walks the files in order)
    enum Got { table_ok, table_failed, child_died_here, child_gone };
    Got next(size_t i, Ed_Table& t)
    {
        Kid& kid = kids_[i % kids_.size()];
        if (kid.dead) return child_gone;
        Header h;
        std::string id;
        t = Ed_Table();
        bool got = read_all(kid.fd, &h, sizeof(h));
        if (got) { id.assign(h.id_len, '\0'); got = !h.id_len || read_all(kid.fd, &id[0], h.id_len); }
        if (got) { t.events.resize(h.n_events); got = !h.n_events || read_all(kid.fd, t.events.data(), h.n_events * sizeof(nchmm_ed_event)); }
        if (!got) {      // the stream ended inside this file's record: the child died reading it (libhdf5 on a corrupt file)
            kid.dead = true;
            t = Ed_Table();
            return child_died_here;
        }
        t.have_sampling_rate = h.have_sr != 0; t.have_events = h.have_ev != 0; t.sampling_rate = h.sampling_rate; t.read_id.swap(id);
        return h.ok ? table_ok : table_failed;
    }

    void finish()
    {
        for (const Kid& k : kids_) close(k.fd);                          // a child still writing gets SIGPIPE
        for (const Kid& k : kids_) { int st = 0; while (waitpid(k.pid, &st, 0) < 0 && errno == EINTR) {} }
        kids_.clear();
    }

private:
    struct Kid { pid_t pid; int fd; bool dead = false; };
    struct Header { uint8_t ok, have_sr, have_ev, pad; uint32_t id_len; double sampling_rate; uint64_t n_events; };
    std::vector<Kid> kids_;

    static bool read_all(int fd, void* p, size_t n)
    {
        char* c = static_cast<char*>(p);
        while (n) {
            const ssize_t r = read(fd, c, n);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) return false;
            c += r; n -= (size_t)r;
        }
        return true;
    }
    static void write_all(int fd, const void* p, size_t n)
    {
        const char* c = static_cast<const char*>(p);
        while (n) {
            const ssize_t r = write(fd, c, n);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) _exit(1);
            c += r; n -= (size_t)r;
        }
    }
    static void serve(int fd, const std::string& fn, const std::string& ed_group)
    {
        Ed_Table t;
        Header h{};
        if (const char* e = std::getenv("NANOCALL_TEST_READER_ABORT"))      // test hook: a file that takes its reader down
            if (*e && fn.find(e) != std::string::npos) abort();
        try { t = read_ed_table(fn, ed_group); h.ok = 1; }
        catch (...) { t = Ed_Table(); h.ok = 0; }      // the parent's summarize() re-reads the file and reports the error
        h.have_sr = t.have_sampling_rate; h.have_ev = t.have_events; h.sampling_rate = t.sampling_rate;
        h.id_len = (uint32_t)t.read_id.size(); h.n_events = t.events.size();
        write_all(fd, &h, sizeof(h));
        write_all(fd, t.read_id.data(), h.id_len);
        write_all(fd, t.events.data(), t.events.size() * sizeof(nchmm_ed_event));
    }
};

// ---------------------------------------------------------------------------------------------------------------
// One worker process per GPU: the link between a worker and the parent.  Frames {type, index, two payload lengths} + payloads
// travel up the worker's data pipe; the parent answers once, on the control pipe, before the counter reduction.
//   'R' index = input index of a read, payloads = its FASTA records, its --dump-params rows   (EVERY read of the share, ascending)
//   'S' index = input index, payload = the read's --stats row                                    (every read, after the last 'R')
//   'U' payload = the RCCL unique id (rank 0 only, when the reduction goes through RCCL; empty: could not make one)
//   'C' payload = 4 host counters + 8 device counters of this worker (uint64); then the worker waits for the parent's verdict:
//       one byte (1: all-reduce with the id that follows, 128 bytes; 0: the parent sums)
//   'G' payload = 1 byte (the all-reduce succeeded) + the 8 reduced counters + this worker's stage clock as text
//   'E' the worker is done; anything else at the end of the stream means it died
// ---------------------------------------------------------------------------------------------------------------
struct Frame_Header { uint8_t type; uint8_t pad[7]; uint64_t index, len_a, len_b; };

static bool fd_read_all(int fd, void* p, size_t n)
{
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t r = read(fd, c, n);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;
        c += r; n -= (size_t)r;
    }
    return true;
}
static bool fd_write_all(int fd, const void* p, size_t n)
{
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t r = write(fd, c, n);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;
        c += r; n -= (size_t)r;
    }
    return true;
}

struct Worker_Link {
    int rank = 0, n_ranks = 1, device = 0;
    int data_fd = -1, ctl_fd = -1;
    bool use_rccl = false;
    std::vector<size_t> global_index;      // of the worker's reads, ascending
    size_t next_emit = 0;                  // reads [0, next_emit) of the share have had their 'R' frame
    std::mutex m;

    void send(uint8_t type, uint64_t index, const std::string& a, const std::string& b = std::string())
    {
        Frame_Header h{};
        h.type = type; h.index = index; h.len_a = a.size(); h.len_b = b.size();
        std::lock_guard<std::mutex> g(m);
        // (the parent gone: nothing to write for any more -- leave at once, without the destructors of a process that holds a GPU)
        if (!fd_write_all(data_fd, &h, sizeof(h)) || !fd_write_all(data_fd, a.data(), a.size()) || !fd_write_all(data_fd, b.data(), b.size())) std::_Exit(EXIT_FAILURE);
    }
    // the record of local read j; reads the pipeline skipped on the way (no events, no candidate) get their empty frame first
    void emit(size_t j, const std::string& fasta, const std::string& dump)
    {
        for (; next_emit < j; ++next_emit) send('R', global_index[next_emit], std::string());
        send('R', global_index[j], fasta, dump);
        next_emit = j + 1;
    }
    void emit_rest() { for (; next_emit < global_index.size(); ++next_emit) send('R', global_index[next_emit], std::string()); }
};

// how far the summary pass has got: reads [0, ready) are summarised (the decode loop runs behind it)
struct Read_Progress {
    std::mutex m;
    std::condition_variable cv;
    size_t ready = 0;
    bool done = false;
    void publish(size_t n, bool finished)
    {
        { std::lock_guard<std::mutex> g(m); ready = n; done = finished; }
        cv.notify_all();
    }
};

static void init_reads(const Pore_Model_Dict_Type& models, const std::list<std::string>& files, std::deque<Fast5_Summary_Type>& reads,
                       Read_Progress& progress, Reader_Procs& readers)
{
    // The reference summarises file after file on one thread (nanocall.cpp:263-273).  Here one thread reads the event
    // tables (HDF5 serialises its calls; from many threads the same reads take 3-4x longer) a block of files ahead, and
    // `-t` threads turn the previous block into summaries (abasic level, strand detection, initial scalings).
    const std::vector<std::string> fv(files.begin(), files.end());   // (`reads` was sized by the caller: its elements do not move)
    const size_t block = 256;
    std::vector<Ed_Table> cur, nxt;
    std::vector<char> cur_ok, nxt_ok;
    auto read_block = [&](size_t b0, std::vector<Ed_Table>& tab, std::vector<char>& ok) {
        const size_t b1 = std::min(fv.size(), b0 + block);
        tab.assign(b1 - b0, Ed_Table());
        ok.assign(b1 - b0, 0);
        auto one = [&](size_t i) {
            try {
                Reader_Procs::Got got = readers.size() ? readers.next(i, tab[i - b0]) : Reader_Procs::child_gone;
                if (got == Reader_Procs::child_gone) {      // no reader processes, or this file's child is no more
                    tab[i - b0] = read_ed_table(fv[i], Fast5_Summary_Type::eventdetection_group());
                    got = Reader_Procs::table_ok;
                }
                if (got == Reader_Procs::child_died_here) {
                    // the file took its reader process down: it is NOT opened again in this process (which holds the GPUs);
                    // the empty table makes summarize() skip the read
                    LOG(warning) << fv[i] << ": the reader process died on this file; read skipped" << std::endl;
                    ok[i - b0] = 1;
                } else {
                    ok[i - b0] = got == Reader_Procs::table_ok ? 1 : 0;
                }
            }
            catch (const Error&) { ok[i - b0] = 0; }   // summarize() re-reads it and reports the error as the reference does
        };
        // File i comes from reader process i % k, each through a pipe of its own, in increasing i: one receiving thread per
        // process (a table is ~300 KB; one thread copying them out of the pipes one after the other was what bounded the whole
        // summary pass: 8000 files in 0.55 s whatever the number of reader processes).
        const size_t k = readers.size();
        static const bool serial_receive = std::getenv("NANOCALL_SERIAL_RECEIVE") != nullptr;      // (A/B switch)
        if (k > 1 && !serial_receive)
            host_parallel(k, (unsigned)k, [&](size_t c) { for (size_t i = b0 + (c + k - b0 % k) % k; i < b1; i += k) one(i); });
        else
            for (size_t i = b0; i < b1; ++i) one(i);
    };
    if (!fv.empty()) read_block(0, cur, cur_ok);
    for (size_t b0 = 0; b0 < fv.size(); b0 += block) {
        const size_t b1 = std::min(fv.size(), b0 + block);
        std::thread reader;
        if (b1 < fv.size()) reader = std::thread([&, b1] { read_block(b1, nxt, nxt_ok); });
        host_parallel(b1 - b0, opts::num_threads, [&](size_t k) {
            reads[b0 + k].summarize(fv[b0 + k], models, opts::double_strand_scaling, cur_ok[k] ? &cur[k] : nullptr);
        });
        if (reader.joinable()) reader.join();
        cur.swap(nxt);
        cur_ok.swap(nxt_ok);
        for (size_t i = b0; i < b1; ++i) LOG(info) << "summary: " << reads[i] << std::endl;
        progress.publish(b1, b1 == fv.size());
    }
    progress.publish(fv.size(), true);
}

// ---------------------------------------------------------------------------------------------------------------
// train_reads + basecall_reads (nanocall.cpp:275-582, 593-869), batched: the reads of a chunk -> SoA events -> job list
// -> nchmm_pool_train_reads -> nchmm_pool_basecall_reads -> FASTA records in input order
// ---------------------------------------------------------------------------------------------------------------
static void write_fasta(std::ostream& os, const std::string& name, const std::string& seq)   // nanocall.cpp:584-591
{
    os << ">" << name << std::endl;
    for (unsigned pos = 0; pos < seq.size(); pos += opts::fasta_line_width) os << seq.substr(pos, opts::fasta_line_width) << std::endl;
}

struct Model_Table {
    std::vector<std::string> names;        // std::map order
    std::vector<int32_t> strand;
    std::vector<float> states;             // n_models x S x 10 (unscaled)
    std::vector<float> mean;               // Pore_Model::mean() per model
};

// One chunk of reads on its way through the three stages of process_reads.
struct Chunk {
    std::vector<size_t> idx;                 // the chunk's reads (indices into `reads`), input order
    std::vector<uint64_t> strand_off;        // 2 nr + 1
    std::vector<uint8_t> together;
    std::vector<float> mean, stdv, start;    // SoA events of both strands of every read
    std::vector<float> r_mean;               // mean level per (read, strand), for the means_apart check
    size_t n_jobs = 0;
    std::vector<int32_t> job_read, job_m0, job_m1, preferred;
    std::vector<float> job_pm, job_st, job_fit;
    std::vector<uint32_t> job_rounds;
    // the decode stage's results
    std::vector<uint16_t> states;
    std::vector<int32_t> best_job;
    std::vector<float> best_logp;
    bool trained = false, decoded = false;
    std::array<std::string, 2> key_of(const Model_Table& M, size_t k) const
    {
        std::array<std::string, 2> key;
        if (job_m0[k] >= 0) key[0] = M.names[(size_t)job_m0[k]];
        if (job_m1[k] >= 0) key[1] = M.names[(size_t)job_m1[k]];
        return key;
    }
};

// hand-over between two stages: one chunk waiting at most (the producer works on the next one meanwhile)
class Chunk_Slot {
public:
    bool put(std::unique_ptr<Chunk> c)       // false: the consumer has gone (an error downstream)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !full_ || closed_; });
        if (closed_) return false;
        c_ = std::move(c); full_ = true;
        cv_.notify_all();
        return true;
    }
    std::unique_ptr<Chunk> take()            // null: the producer has finished
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return full_ || closed_; });
        if (!full_) return nullptr;
        full_ = false;
        cv_.notify_all();
        return std::move(c_);
    }
    void close() { { std::lock_guard<std::mutex> g(m_); closed_ = true; } cv_.notify_all(); }
private:
    std::mutex m_;
    std::condition_variable cv_;
    std::unique_ptr<Chunk> c_;
    bool full_ = false, closed_ = false;
};

// The reference runs its two loops over all reads one after the other, each a pfor over reads (nanocall.cpp:282-579, 611-866),
// and writes the records of a chunk of reads in input order (:858-861).  Here the reads go through in chunks of tens of
// millions of events and the three things a chunk needs run side by side on consecutive chunks:
//   prepare   wait for the summary pass, load the events, pack them, enumerate the (read, model) jobs        -- host threads
//   device    nchmm_pool_train_reads, nchmm_pool_basecall_reads over the pool's GPUs                         -- this thread
//   finish    base sequences, FASTA records in input order, parameter dump, drop the events                  -- host threads
// so that the GPUs decode chunk k while chunk k + 1 is being packed and chunk k - 1 is being written.
static void process_reads(nchmm_pool* pool, const Pore_Model_Dict_Type& models, std::deque<Fast5_Summary_Type>& reads, std::ostream* os_p,
                          uint64_t counters[4], Read_Progress& progress, Worker_Link* link)
{
    // (a worker process hands its records and dump rows to the parent, which owns the output files)
    const bool want_dump = !opts::dump_params_fn.get().empty();
    std::ofstream dump;
    if (want_dump && !link) {
        dump.open(opts::dump_params_fn.get());
        dump << "#read_id\tstrand\tmodel\tscale\tshift\tdrift\tvar\tscale_sd\tvar_sd\tp_stay\tp_skip\tlog_path_prob\trounds\tfit" << std::endl;
        dump << std::hexfloat;
    }
    Model_Table M;
    for (const auto& p : models) {
        M.names.push_back(p.first);
        M.strand.push_back((int32_t)p.second.strand());
        M.states.insert(M.states.end(), p.second.data(), p.second.data() + (size_t)4096 * 10);
        M.mean.push_back(p.second.mean());
    }
    const size_t n_models = M.names.size();
    nchmm_train_opts o;
    check(nchmm_train_opts_default(&o), "nchmm_train_opts_default");
    o.scaling_num_events = opts::scaling_num_events; o.scaling_max_rounds = opts::scaling_max_rounds;
    o.scaling_min_progress = opts::scaling_min_progress; o.scaling_select_threshold = opts::scaling_select_threshold;
    o.min_ed_events = opts::min_ed_events; o.train_scaling = !opts::no_train_scaling; o.train_transitions = !opts::no_train_transitions;
    o.train_drift = opts::train_drift.get() == "1"; o.default_p_stay = opts::pr_stay; o.default_p_skip = opts::pr_skip;

    const uint64_t chunk_cap = std::max<uint64_t>(1, opts::chunk_events.get()) * (uint64_t)nchmm_pool_size(pool);
    // The summary pass (one FAST5 reader thread + the summarising threads) runs AHEAD of this loop: a chunk is decoded as soon
    // as enough summarised reads are waiting (or the pass has finished), so the GPU stages hide behind the file reading.
    const uint64_t chunk_min = std::min<uint64_t>(chunk_cap, (uint64_t)16000000 * (uint64_t)nchmm_pool_size(pool));   // (smaller chunks leave the GPU launches short: 3x the GPU time at 4 M)
    // the host threads are shared by the prepare and the finish stage (they run at the same time)
    const unsigned side_threads = std::max(1u, opts::num_threads / 2);

    // ---------------- prepare: the next chunk of consecutive summarised reads, up to the event budget (at least one) ----------------
    size_t next = 0;
    auto prepare = [&]() -> std::unique_ptr<Chunk> {
        while (next < reads.size()) {
            std::unique_ptr<Chunk> C(new Chunk());
            std::vector<size_t>& idx = C->idx;
            uint64_t budget = 0;
            size_t scan = next;
            for (;;) {
                size_t lim;
                bool all;
                {
                    std::unique_lock<std::mutex> lk(progress.m);
                    progress.cv.wait(lk, [&] { return progress.ready > scan || progress.done; });
                    lim = progress.ready; all = progress.done;
                }
                bool full = false;
                while (scan < lim) {
                    const Fast5_Summary_Type& r = reads[scan];
                    const uint64_t ev = r.num_ed_events ? (r.strand_bounds[1] - r.strand_bounds[0]) + (r.strand_bounds[3] > r.strand_bounds[2] ? r.strand_bounds[3] - r.strand_bounds[2] : 0) : 0;
                    if (!idx.empty() && budget + ev > chunk_cap) { full = true; break; }
                    if (r.num_ed_events) { idx.push_back(scan); budget += ev; }   // (reads without events are skipped, nanocall.cpp:294,623)
                    ++scan;
                }
                if (full || budget >= chunk_min || (all && scan >= reads.size())) break;   // else: wait for more summaries
            }
            next = scan;
            if (idx.empty()) continue;
            const size_t nr = idx.size();
            {
                STAGE("load_events_s");
                host_parallel(nr, side_threads, [&](size_t i) { reads[idx[i]].load_events(); });
            }
            STAGE("soa_and_jobs_s");
            C->strand_off.assign(2 * nr + 1, 0);
            C->together.resize(nr);
            for (size_t i = 0; i < nr; ++i) {
                const Fast5_Summary_Type& r = reads[idx[i]];
                C->strand_off[2 * i + 1] = C->strand_off[2 * i] + r.events(0).size();
                C->strand_off[2 * i + 2] = C->strand_off[2 * i + 1] + r.events(1).size();
                C->together[i] = r.scale_strands_together ? 1 : 0;
            }
            const std::vector<uint64_t>& strand_off = C->strand_off;
            const size_t total = (size_t)strand_off[2 * nr];
            C->mean.resize(total + 1); C->stdv.resize(total + 1); C->start.resize(total + 1);
            host_parallel(nr, side_threads, [&](size_t i) {
                const Fast5_Summary_Type& r = reads[idx[i]];
                for (unsigned st = 0; st < 2; ++st) {
                    size_t k = (size_t)strand_off[2 * i + st];
                    for (const Event_Type& e : r.events(st)) { C->mean[k] = e.mean; C->stdv[k] = e.stdv; C->start[k] = e.start; ++k; }
                }
            });
            // mean of the events' levels per strand, for the means_apart check (nanocall.cpp:628-641,673-683)
            C->r_mean.assign(2 * nr, 0.f);
            for (size_t i = 0; i < nr; ++i)
                for (unsigned st = 0; st < 2; ++st) {
                    const size_t b = (size_t)strand_off[2 * i + st], n = (size_t)(strand_off[2 * i + st + 1] - strand_off[2 * i + st]);
                    if (n < opts::min_ed_events) continue;
                    float sd;
                    check(nchmm_mean_stdv(n, &C->mean[b], &C->r_mean[2 * i + st], &sd), "nchmm_mean_stdv");
                }
            // ---- jobs: one per iteration of the reference's model loops (nanocall.cpp:300-323,356-358,474) ----
            size_t n_jobs = 0;
            check(nchmm_train_enumerate(&o, n_models, M.strand.data(), nr, strand_off.data(), C->together.data(), &n_jobs, nullptr, nullptr, nullptr),
                  "nchmm_train_enumerate");
            C->job_read.resize(n_jobs); C->job_m0.resize(n_jobs); C->job_m1.resize(n_jobs);
            check(nchmm_train_enumerate(&o, n_models, M.strand.data(), nr, strand_off.data(), C->together.data(), &n_jobs, C->job_read.data(), C->job_m0.data(),
                                        C->job_m1.data()), "nchmm_train_enumerate");
            C->n_jobs = n_jobs;
            C->job_pm.resize(6 * n_jobs); C->job_st.resize(4 * n_jobs); C->job_fit.assign(n_jobs, -INFINITY);
            C->job_rounds.assign(n_jobs, 0);
            C->preferred.assign(3 * nr, -1);
            for (size_t k = 0; k < n_jobs; ++k) {
                const Fast5_Summary_Type& r = reads[idx[(size_t)C->job_read[k]]];
                const auto key = C->key_of(M, k);
                const Pore_Model_Parameters_Type& pm = r.pm_params_m.at(key);
                const auto& stp = r.st_params_m.at(key);
                const float p6[6] = {pm.scale, pm.shift, pm.drift, pm.var, pm.scale_sd, pm.var_sd};
                std::copy(p6, p6 + 6, &C->job_pm[6 * k]);
                for (int s = 0; s < 2; ++s) { C->job_st[4 * k + 2 * s] = stp[s].p_stay; C->job_st[4 * k + 2 * s + 1] = stp[s].p_skip; }
            }
            return C;
        }
        return nullptr;
    };

    // ---------------- device: train, then decode the chunk on the pool's GPUs ----------------
    auto on_device = [&](Chunk& C) {
        const std::vector<size_t>& idx = C.idx;
        const size_t nr = idx.size(), n_jobs = C.n_jobs;
        if (opts::train && n_jobs) {
            STAGE("training_total_s");
            const auto t0 = std::chrono::steady_clock::now();
            check(nchmm_pool_train_reads(pool, &o, n_models, M.states.data(), nr, C.strand_off.data(), C.mean.data(), C.stdv.data(), C.start.data(), n_jobs,
                                         C.job_read.data(), C.job_m0.data(), C.job_m1.data(), C.job_pm.data(), C.job_st.data(), C.job_fit.data(),
                                         C.job_rounds.data(), C.preferred.data()), "nchmm_pool_train_reads");
            counters[2] += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            C.trained = true;
        }
        if (opts::basecall && n_jobs) {
            STAGE("basecalling_total_s");
            const size_t total = (size_t)C.strand_off[2 * nr];
            C.states.resize(total + 1);
            C.best_job.assign(2 * nr, -1);
            C.best_logp.assign(2 * nr, NAN);
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = nchmm_pool_basecall_reads(pool, &o, n_models, M.states.data(), nr, C.strand_off.data(), C.mean.data(), C.stdv.data(), C.start.data(),
                                                     n_jobs, C.job_read.data(), C.job_m0.data(), C.job_m1.data(), C.job_pm.data(), C.job_st.data(),
                                                     C.preferred.data(), C.states.data(), C.best_job.data(), C.best_logp.data());
            if (rc != NCHMM_OK && rc != NCHMM_E_NUMERIC) check(rc, "nchmm_pool_basecall_reads");
            counters[3] += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            C.decoded = true;
        }
    };

    // ---------------- finish: base sequences + FASTA records, built per read in parallel, emitted in input order (pfor's output_chunk, :858-861) ----------------
    auto finish = [&](Chunk& C) {
        const std::vector<size_t>& idx = C.idx;
        const size_t nr = idx.size();
        if (C.trained) {
            // what the training of the chunk found goes into the reads' summaries (and the log) here, not on the thread that
            // feeds the GPUs: the decode stage reads the job arrays, not the summaries
            STAGE("apply_training_s");
            const size_t n_jobs = C.n_jobs;
            for (size_t k = 0; k < n_jobs; ++k) {
                Fast5_Summary_Type& r = reads[idx[(size_t)C.job_read[k]]];
                const auto key = C.key_of(M, k);
                Pore_Model_Parameters_Type& pm = r.pm_params_m.at(key);
                pm.scale = C.job_pm[6 * k]; pm.shift = C.job_pm[6 * k + 1]; pm.drift = C.job_pm[6 * k + 2]; pm.var = C.job_pm[6 * k + 3];
                pm.scale_sd = C.job_pm[6 * k + 4]; pm.var_sd = C.job_pm[6 * k + 5];
                auto& stp = r.st_params_m.at(key);
                const bool two_d = C.job_m0[k] >= 0 && C.job_m1[k] >= 0;
                for (int s = 0; s < 2; ++s)
                    if (s == 0 ? C.job_m0[k] >= 0 : C.job_m1[k] >= 0) { stp[s].p_stay = C.job_st[4 * k + 2 * s]; stp[s].p_skip = C.job_st[4 * k + 2 * s + 1]; }
                const int strand_tag = two_d ? 2 : (C.job_m0[k] >= 0 ? 0 : 1);
                const std::string m_name = two_d ? key[0] + "+" + key[1] : key[strand_tag];
                if (logger::info <= logger::threshold()) {   // nanocall.cpp:427-434 / :543-550
                    std::ostringstream stp_s;
                    if (two_d) stp_s << stp[0] << "," << stp[1]; else stp_s << stp[(size_t)strand_tag];
                    LOG(info) << "scaling_result read [" << r.read_id << "] strand [" << strand_tag << "] model [" << m_name << "] pm_params [" << pm
                              << "] st_params [" << stp_s.str() << "] fit [" << C.job_fit[k] << "] rounds [" << C.job_rounds[k] << "]" << std::endl;
                }
            }
            for (size_t i = 0; i < nr; ++i) {   // model selection, nanocall.cpp:437-459,552-570
                Fast5_Summary_Type& r = reads[idx[i]];
                for (int kind = 0; kind < 3; ++kind) {
                    const int32_t k = C.preferred[3 * i + kind];
                    if (k < 0) continue;
                    const auto key = C.key_of(M, (size_t)k);
                    if (kind == 2) { r.preferred_model[2] = key; LOG(info) << "selected_model read [" << r.read_id << "] strand [2] model [" << key[0] << "+" << key[1] << "]" << std::endl; }
                    else { r.preferred_model[(size_t)kind][(size_t)kind] = key[(size_t)kind]; LOG(info) << "selected_model read [" << r.read_id << "] strand [" << kind << "] model [" << key[(size_t)kind] << "]" << std::endl; }
                }
            }
        
        }
        if (C.decoded) {
            STAGE("fasta_s");
            const std::vector<uint64_t>& strand_off = C.strand_off;
            std::vector<std::string> record(nr), dump_rec(nr);
            host_parallel(nr, side_threads, [&](size_t i) {
                Fast5_Summary_Type& r = reads[idx[i]];
                std::ostringstream oss;
                for (unsigned st = 0; st < 2; ++st) {
                    const int32_t k = C.best_job[2 * i + st];
                    if (k < 0) continue;
                    const size_t b = (size_t)strand_off[2 * i + st], n = (size_t)(strand_off[2 * i + st + 1] - strand_off[2 * i + st]);
                    const auto key = C.key_of(M, (size_t)k);
                    const bool two_d = C.job_m0[(size_t)k] >= 0 && C.job_m1[(size_t)k] >= 0;
                    Pore_Model_Parameters_Type best_pm;
                    best_pm.scale = C.job_pm[6 * (size_t)k]; best_pm.shift = C.job_pm[6 * (size_t)k + 1]; best_pm.drift = C.job_pm[6 * (size_t)k + 2];
                    best_pm.var = C.job_pm[6 * (size_t)k + 3]; best_pm.scale_sd = C.job_pm[6 * (size_t)k + 4]; best_pm.var_sd = C.job_pm[6 * (size_t)k + 5];
                    State_Transition_Parameters_Type best_st;
                    best_st.p_stay = C.job_st[4 * (size_t)k + 2 * st]; best_st.p_skip = C.job_st[4 * (size_t)k + 2 * st + 1];
                    // means_apart warning, :673-683: model mean after scaling = mean * scale + shift only approximately; the reference
                    // recomputes the statistics of the scaled model, so do that
                    if (logger::warning <= logger::threshold() && n >= opts::min_ed_events) {
                        Pore_Model_Type pm(models.at(key[st]));
                        pm.scale(best_pm);
                        if (std::abs(C.r_mean[2 * i + st] - pm.mean()) > 5.0) {
                            LOG(warning) << "means_apart read [" << r.read_id << "] strand [" << st << "] model [" << key[st] << "] parameters [" << best_pm
                                         << "] model_mean=[" << pm.mean() << "] events_mean=[" << C.r_mean[2 * i + st] << "]" << std::endl;
                        }
                    }
                    LOG(info) << "best_model read [" << r.read_id << "] strand [" << st << "] model [" << key[st] << "] pm_params [" << best_pm << "] st_params ["
                              << best_st << "] log_path_prob [" << C.best_logp[2 * i + st] << "]" << std::endl;
                    // nanocall.cpp:761-763 (2D) / :836 (1D)
                    r.preferred_model[st][st] = key[st];
                    if (two_d) {
                        r.pm_params_m[r.preferred_model[st]] = best_pm;
                        r.st_params_m[r.preferred_model[st]][st] = best_st;
                    }
                    std::string seq(6 * n + 1, '\0');
                    size_t len = 0;
                    check(nchmm_base_seq(n, &C.states[b], nullptr, &seq[0], &len), "nchmm_base_seq");
                    seq.resize(len);
                    std::ostringstream nm;
                    nm << r.read_id << ":" << r.base_file_name << ":" << st;
                    write_fasta(oss, nm.str(), seq);
                    if (want_dump) {
                        std::ostringstream d;
                        d << std::hexfloat << r.read_id << '\t' << st << '\t' << key[st] << '\t' << best_pm.scale << '\t' << best_pm.shift << '\t' << best_pm.drift
                          << '\t' << best_pm.var << '\t' << best_pm.scale_sd << '\t' << best_pm.var_sd << '\t' << best_st.p_stay << '\t' << best_st.p_skip << '\t'
                          << C.best_logp[2 * i + st] << '\t' << std::dec << C.job_rounds[(size_t)k] << '\t' << std::hexfloat << C.job_fit[(size_t)k] << '\n';
                        dump_rec[i] += d.str();
                    }
                }
                record[i] = oss.str();
            });
            for (size_t i = 0; i < nr; ++i) {
                if (link) link->emit(idx[i], record[i], dump_rec[i]);
                else {
                    *os_p << record[i];
                    if (dump.is_open()) dump << dump_rec[i];
                }
                bool header = false;
                for (char c : record[i]) {   // bases = sequence characters (header lines excluded)
                    if (c == '>') header = true;
                    else if (c == '\n') header = false;
                    else if (!header) ++counters[1];
                }
            }
        }
        counters[0] += nr;
        for (size_t i = 0; i < nr; ++i) reads[idx[i]].drop_events();
    };

    // ---------------- the three stages side by side ----------------
    // (an exception in a side stage is kept, the hand-overs are closed so that nobody waits for a chunk that will not come, and it
    // is rethrown here after the joins; --serial-chunks: one chunk at a time through all three, as rounds 1-4 did)
    if (opts::serial_chunks) {
        while (std::unique_ptr<Chunk> C = prepare()) { on_device(*C); finish(*C); }
        return;
    }
    Chunk_Slot to_device, to_finish;
    std::exception_ptr prepare_error, finish_error;
    std::thread prepare_thread([&] {
        try {
            while (std::unique_ptr<Chunk> C = prepare())
                if (!to_device.put(std::move(C))) break;
        } catch (...) { prepare_error = std::current_exception(); }
        to_device.close();
    });
    std::thread finish_thread([&] {
        try {
            while (std::unique_ptr<Chunk> C = to_finish.take()) finish(*C);
        } catch (...) { finish_error = std::current_exception(); to_finish.close(); }
    });
    std::exception_ptr device_error;
    try {
        while (std::unique_ptr<Chunk> C = to_device.take()) {
            on_device(*C);
            if (!to_finish.put(std::move(C))) break;
        }
    } catch (...) { device_error = std::current_exception(); }
    to_device.close();          // (an error here: the prepare stage stops at its next hand-over)
    to_finish.close();
    prepare_thread.join();
    finish_thread.join();
    for (const std::exception_ptr& e : {device_error, prepare_error, finish_error}) if (e) std::rethrow_exception(e);
}

static double epoch_now()
{
    return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// The whole pipeline over `files` in this process: reader processes, summary pass, devices, training, basecalling, output.
// link == nullptr: the run is this process (output, --stats and --dump-params files written here, every device of the run in the
// pool).  link != nullptr: a worker process of a multi-GPU run -- one device, records / stats rows / counters go to the parent.
static int run_reads(const Pore_Model_Dict_Type& models, const std::list<std::string>& files, Stage_Clock::Scope* whole, Worker_Link* link)
{
    std::deque<Fast5_Summary_Type> reads;
    // reader processes: forked here, while this process is still single-threaded and has not touched the GPU
    Reader_Procs readers;
    {
        int k = opts::reader_procs.get();
        if (k < 0) k = files.size() >= 64 ? (int)std::min<unsigned>(opts::num_threads, 16u) : 0;
        if (k > 1) {
            const std::vector<std::string> fv(files.begin(), files.end());
            readers.start(fv, Fast5_Summary_Type::eventdetection_group(), (unsigned)k);
        }
        unsigned n_threads = 0;      // fork() is only safe here because this process is still single-threaded: say so in the log
        if (DIR* d = opendir("/proc/self/task")) {
            while (const dirent* e = readdir(d)) if (e->d_name[0] != '.') ++n_threads;
            closedir(d);
        }
        LOG(info) << "reader_procs=" << readers.size() << " threads_at_fork=" << n_threads << std::endl;
    }
    std::ofstream ofs;
    std::ostream* os_p = &std::cout;
    if (!link && !opts::output_fn.get().empty()) {
        ofs.open(opts::output_fn.get());
        if (!ofs) { LOG(error) << "cannot open output [" << opts::output_fn.get() << "]" << std::endl; return EXIT_FAILURE; }
        os_p = &ofs;
    }
    uint64_t counters[4] = {0, 0, 0, 0};   // reads, bases, training us, basecalling us
    // is there a device at all?  (before any file is read: a machine without one fails at once)
    // (nothing to train and nothing to decode -- `--no-train --no-basecall --stats f`: segmentation and initial scalings only, as
    // the reference allows -- is host work: no device is asked for)
    const bool need_device = opts::train || opts::basecall;
    int use = 0, n_dev = 0;
    int dc_rc = NCHMM_OK;
    if (need_device) { STAGE("device_count_s"); dc_rc = nchmm_device_count(&n_dev); }
    else { n_dev = 1; LOG(info) << "devices=0 (no training, no basecalling: the summary pass only)" << std::endl; }
    if (dc_rc != NCHMM_OK || n_dev < 1) {
        LOG(error) << "no usable GPU: this build of nanocall decodes on MI355X only (there is no CPU path)" << std::endl;
        return EXIT_FAILURE;
    }
    use = link ? 1 : (opts::gpus.get() > 0 ? opts::gpus.get() : n_dev);
    if (use > n_dev) { LOG(error) << "--gpus " << use << " requested but only " << n_dev << " visible" << std::endl; return EXIT_FAILURE; }
    if (link && link->device >= n_dev) { LOG(error) << "worker " << link->rank << ": no device " << link->device << " (" << n_dev << " visible)" << std::endl; return EXIT_FAILURE; }
    // the summary pass (init_reads) runs on its own threads -- from here on, i.e. while the devices are being initialised below
    // (0.05-0.2 s: runtime start-up, contexts, streams) -- and the decode loop follows it block by block
    reads.resize(files.size());
    Read_Progress progress;
    // (an exception must not leave the thread body -- std::terminate with the GPUs held -- and the decode loop must not wait
    // forever for summaries that will never come: keep it, declare the pass finished with what is ready, rethrow after the join)
    std::exception_ptr summary_error;
    std::thread summary_pass([&] {
        try {
            STAGE("init_reads_s");
            init_reads(models, files, reads, progress, readers);
        } catch (...) {
            summary_error = std::current_exception();
            size_t ready;
            { std::lock_guard<std::mutex> g(progress.m); ready = progress.ready; }
            progress.publish(ready, true);
        }
    });
    // devices: one context + host thread per GPU
    nchmm_pool* pool = nullptr;
    auto open_devices = [&]() -> int {
        // Workspaces: back-pointers 4 KiB and alpha rows 16 KiB per event in flight.  The library would take 60 % / 25 % of the
        // device for them (best for a resident benchmark loop); a command-line run pays for mapping that memory once, ~20 ms per
        // GiB on a cold device, so it caps them where the launches are still long enough (measured: 32 GiB of back-pointers =
        // 3 x 512 reads of 5000 events per launch costs 2 % of the decode rate).
        setenv("NCHMM_WS_BUDGET_MB", "32768", 0);
        setenv("NCHMM_FB_BUDGET_MB", "16384", 0);
        std::vector<int> ids;
        if (link) ids.push_back(link->device);
        else if (const char* e = std::getenv("NANOCALL_DEVICE_IDS"); e && *e) {   // e.g. "0,0": several contexts on one GPU (test hook)
            std::istringstream is(e);
            std::string tok;
            while (std::getline(is, tok, ',')) ids.push_back(std::atoi(tok.c_str()));
            use = (int)ids.size();
        }
        { STAGE("device_init_s"); check(nchmm_pool_create(&pool, use, ids.empty() ? nullptr : ids.data()), "nchmm_pool_create"); }
        LOG(info) << "devices=" << use << " (of " << n_dev << " visible)" << std::endl;
        if (opts::train && files.size() >= 512) {
            // a long run: the alpha-row workspace of the EM rounds now, while the summary pass is still reading files -- on a device
            // whose memory is not mapped yet its first allocation costs ~20 ms per GiB, which would land in the first chunk's rounds
            // (a read brings ~2 model pairs x 4 windows x scaling_num_events / 2 events per round; the library caps it at its budget)
            STAGE("reserve_workspace_s");
            check(nchmm_pool_reserve_fb_workspace(pool, files.size() / (size_t)use * 8 * (size_t)opts::scaling_num_events.get()), "nchmm_pool_reserve_fb_workspace");
        }
        if (opts::basecall && files.size() >= 512) {
            // ... and the back-pointer regions: for strands half as long again as the longest among the first thousand reads the
            // summary pass delivers, at least 6000 events (16 GiB).  They grow when longer
            // strands come -- but growing means freeing and allocating tens of GiB in the middle of the decode stage, which costs
            // 0.5-1.4 s on a device that earlier processes have left memory to be reclaimed on (measured: a run on log-normally long
            // reads behind five runs on 5000-event reads), so the estimate is taken from the data, not from a constant.  Here the
            // allocation happens while the summary pass is still reading files.
            STAGE("reserve_workspace_s");
            size_t seen;
            {
                // (the first thousand summaries are there 0.1 s into the pass; the first chunk needs three times as many, so the
                // wait is not on anybody's path)
                const size_t enough = std::min<size_t>(files.size(), 1024);
                std::unique_lock<std::mutex> lk(progress.m);
                progress.cv.wait(lk, [&] { return progress.ready >= enough || progress.done; });
                seen = progress.ready;
            }
            uint64_t longest_strand = 0;
            for (size_t i = 0; i < seen; ++i) {        // (`reads` was sized up front and [0, ready) is final: safe beside the pass)
                const Fast5_Summary_Type& r = reads[i];
                if (!r.num_ed_events) continue;
                longest_strand = std::max<uint64_t>(longest_strand, r.strand_bounds[1] - r.strand_bounds[0]);
                if (r.strand_bounds[3] > r.strand_bounds[2]) longest_strand = std::max<uint64_t>(longest_strand, r.strand_bounds[3] - r.strand_bounds[2]);
            }
            size_t strand_events = (size_t)std::max<uint64_t>(6000, longest_strand + longest_strand / 2);
            if (const char* e = std::getenv("NANOCALL_RESERVE_EVENTS")) strand_events = (size_t)std::atol(e);      // (measurement hook; 0: no reservation)
            LOG(info) << "reserve_viterbi_workspace strand_events=" << strand_events << " (longest of the first " << seen << " reads: " << longest_strand << ")" << std::endl;
            if (strand_events) check(nchmm_pool_reserve_viterbi_workspace(pool, strand_events), "nchmm_pool_reserve_viterbi_workspace");
        }
        return EXIT_SUCCESS;
    };
    try {
        const int rc = need_device ? open_devices() : EXIT_SUCCESS;
        if (rc != EXIT_SUCCESS) { summary_pass.join(); return rc; }
    } catch (...) {
        summary_pass.join();
        throw;
    }
    try {
        if (opts::train || opts::basecall) { STAGE("process_reads_s"); process_reads(pool, models, reads, os_p, counters, progress, link); }
    } catch (...) {
        summary_pass.join();
        throw;
    }
    summary_pass.join();
    if (summary_error) std::rethrow_exception(summary_error);
    os_p->flush();
    uint64_t dev[8];
    int used_rccl = 0;
    if (link) {
        // ---- a worker: the rest of its records, its --stats rows, its counters; then the reduction with the other workers ----
        link->emit_rest();
        if (!opts::stats_fn.get().empty()) {
            // (the rows are written as ONE stream would write them: the parameter columns leave std::fixed / precision 5 set on
            // the stream, Pore_Model.hpp:72-76, so only the very first row of the file has its abasic level in the default format)
            std::ostringstream row;
            for (size_t j = 0; j < reads.size(); ++j) {
                if (link->global_index[j] != 0) row << std::fixed << std::setprecision(5);
                row.str(std::string());
                reads[j].write_tsv(row);
                link->send('S', link->global_index[j], row.str());
            }
        }
        unsetenv("NCHMM_POOL_FORCE_RCCL");          // (this worker's own figures: a plain read-out; the reduction is across the workers)
        std::fill(dev, dev + 8, (uint64_t)0);
        if (pool) check(nchmm_pool_counters(pool, dev, nullptr), "nchmm_pool_counters");
        if (link->use_rccl && link->rank == 0) {
            uint8_t id[NCHMM_RCCL_ID_BYTES];
            const bool have = need_device && nchmm_rccl_unique_id(id) == NCHMM_OK;
            link->send('U', 0, have ? std::string(reinterpret_cast<const char*>(id), sizeof(id)) : std::string());
        }
        uint64_t mine[12] = {counters[0], counters[1], counters[2], counters[3]};
        std::copy(dev, dev + 8, mine + 4);
        link->send('C', 0, std::string(reinterpret_cast<const char*>(mine), sizeof(mine)));
        uint8_t go = 0, id[NCHMM_RCCL_ID_BYTES];
        if (!fd_read_all(link->ctl_fd, &go, 1) || (go && !fd_read_all(link->ctl_fd, id, sizeof(id)))) std::_Exit(EXIT_FAILURE);      // the parent is gone
        uint64_t red[8];
        std::copy(dev, dev + 8, red);
        if (go) {
            STAGE("counter_allreduce_s");
            used_rccl = nchmm_counters_allreduce(link->device, link->n_ranks, link->rank, id, red) == NCHMM_OK;
        }
        if (pool) { STAGE("device_release_s"); nchmm_pool_destroy(pool); pool = nullptr; }
        LOG(info) << "worker_counters reads=" << counters[0] << " bases=" << counters[1] << " strands_decoded=" << dev[0] << " events_decoded=" << dev[1]
                  << " fb_windows=" << dev[4] << " fb_event_rounds=" << dev[5] << " training_secs=" << counters[2] / 1e6 << " basecalling_secs=" << counters[3] / 1e6 << std::endl;
        delete whole;
        LOG(info) << "worker_stage_wall_secs" << stage_clock.str() << std::endl;
        std::string g(1, used_rccl ? '\1' : '\0');
        g.append(reinterpret_cast<const char*>(red), sizeof(red));
        link->send('G', 0, g, stage_clock.str());
        link->send('E', 0, std::string());
        readers.finish();
        std::cout.flush(); std::cerr.flush(); std::clog.flush();
        std::_Exit(EXIT_SUCCESS);
    }
    std::fill(dev, dev + 8, (uint64_t)0);
    if (pool) check(nchmm_pool_counters(pool, dev, &used_rccl), "nchmm_pool_counters");
    LOG(info) << "counters reads=" << counters[0] << " bases=" << counters[1] << " strands_decoded=" << dev[0] << " events_decoded=" << dev[1]
              << " fb_windows=" << dev[4] << " fb_event_rounds=" << dev[5] << " gathered_by=" << (used_rccl ? "rccl_allreduce" : "host_sum")
              << " training_secs=" << counters[2] / 1e6 << " basecalling_secs=" << counters[3] / 1e6 << std::endl;
    if (pool) { STAGE("device_release_s"); nchmm_pool_destroy(pool); }
    delete whole;
    LOG(info) << "stage_wall_secs" << stage_clock.str() << std::endl;
    if (!opts::stats_fn.get().empty()) {   // nanocall.cpp:893-903
        std::ofstream sfs(opts::stats_fn.get());
        if (!sfs) { LOG(error) << "cannot open stats file [" << opts::stats_fn.get() << "]" << std::endl; return EXIT_FAILURE; }
        Fast5_Summary_Type::write_tsv_header(sfs);
        sfs << std::endl;
        for (const auto& s : reads) {
            s.write_tsv(sfs);
            sfs << std::endl;
        }
        sfs.close();
        if (!sfs) { LOG(error) << "error writing stats file [" << opts::stats_fn.get() << "]" << std::endl; return EXIT_FAILURE; }
    }
    // Everything this run produces is written: close the output here and leave from HERE -- returning would first take down
    // this function's locals (every read's summary and event vectors: a gigabyte in a few hundred thousand allocations for
    // 8000 reads, 0.3 s of free() that buy nothing), then the GPU runtime and the worker threads' statics.
    // (NANOCALL_FULL_EXIT=1: return and run every destructor, for leak checkers.)
    if (ofs.is_open()) {
        ofs.close();
        if (!ofs) { LOG(error) << "error writing output [" << opts::output_fn.get() << "]" << std::endl; return EXIT_FAILURE; }
    }
    {   // (what the kernel has to take down when this process leaves: resident, pinned and peak memory, threads)
        std::ifstream st("/proc/self/status");
        std::string line, keep;
        while (std::getline(st, line))
            for (const char* k : {"VmHWM:", "VmRSS:", "VmPin:", "VmLck:", "Threads:"})
                if (line.compare(0, std::strlen(k), k) == 0) { for (char& ch : line) if (ch == '\t') ch = ' '; keep += " [" + line + "]"; }
        LOG(info) << "memory_at_exit" << keep << std::endl;
    }
    LOG(info) << "epoch_at_exit=" << std::fixed << epoch_now() << std::endl;
    if (!std::getenv("NANOCALL_FULL_EXIT")) {
        readers.finish();            // (the reader processes have served their last file: reaped here, not left to init)
        std::cout.flush(); std::cerr.flush(); std::clog.flush();
        std::_Exit(EXIT_SUCCESS);
    }
    return EXIT_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------
// One worker process per GPU
// ---------------------------------------------------------------------------------------------------------------
// How many devices a process of this environment would see -- asked by a short-lived child, so that THIS process has not touched
// the HIP runtime when it forks its workers (and never does: the parent of a multi-GPU run only moves bytes).
// GPUs this process could open at all: the render nodes under /dev/dri it may open read-write (a container is given the nodes of
// its GPUs only).  An upper bound that costs microseconds and no runtime: with fewer than two there is nothing to fan out over
// and the count itself is left to run_reads; with two or more the exact number (the runtime's, which honours HIP_VISIBLE_DEVICES
// and its relatives) is asked by the child below.
static int openable_render_nodes()
{
    int n = 0;
    if (DIR* d = opendir("/dev/dri")) {
        while (const dirent* e = readdir(d)) {
            if (std::strncmp(e->d_name, "renderD", 7) != 0) continue;
            const std::string path = std::string("/dev/dri/") + e->d_name;
            const int fd = open(path.c_str(), O_RDWR | O_CLOEXEC);
            if (fd >= 0) { ++n; close(fd); }
        }
        closedir(d);
    }
    return n;
}

struct Device_Probe {
    pid_t pid = -1;
    int fd = -1;
    void start()
    {
        int p[2];
        if (pipe(p) != 0) return;
        std::cout.flush(); std::clog.flush();
        pid = fork();
        if (pid < 0) { close(p[0]); close(p[1]); return; }
        if (pid == 0) {
            close(p[0]);
            int n = 0;
            if (nchmm_device_count(&n) != NCHMM_OK) n = 0;
            (void)fd_write_all(p[1], &n, sizeof(n));
            _exit(0);
        }
        close(p[1]);
        fd = p[0];
    }
    int finish()      // the count; -1: could not ask
    {
        if (pid < 0) return -1;
        int n = -1;
        if (!fd_read_all(fd, &n, sizeof(n))) n = -1;
        close(fd);
        int st = 0;
        while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
        pid = -1;
        return n;
    }
};

// frames of one worker, taken off its pipe by a thread of the parent as they come (a worker never waits for the merge)
struct Worker_Stream {
    pid_t pid = -1;
    int data_fd = -1, ctl_fd = -1, device = 0;
    size_t n_reads = 0;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    struct Frame { uint8_t type; uint64_t index; std::string a, b; };
    std::deque<Frame> q;
    bool eof = false;            // the pipe has ended (after 'E': normally; before: the worker died)

    void pump()
    {
        for (;;) {
            Frame_Header h;
            Frame f;
            bool ok = fd_read_all(data_fd, &h, sizeof(h));
            if (ok) { f.type = h.type; f.index = h.index; f.a.resize(h.len_a); f.b.resize(h.len_b); }
            ok = ok && (!h.len_a || fd_read_all(data_fd, &f.a[0], h.len_a)) && (!h.len_b || fd_read_all(data_fd, &f.b[0], h.len_b));
            std::lock_guard<std::mutex> g(m);
            if (!ok) { eof = true; cv.notify_all(); return; }
            const bool last = f.type == 'E';
            q.push_back(std::move(f));
            if (last) eof = true;
            cv.notify_all();
            if (last) return;
        }
    }
    // the next frame; false: the stream ended without one (or nothing came within `timeout_s`, when that is >= 0)
    bool next(Frame& f, double timeout_s = -1.0)
    {
        std::unique_lock<std::mutex> lk(m);
        auto ready = [&] { return !q.empty() || eof; };
        if (timeout_s < 0) cv.wait(lk, ready);
        // (a deadline on the system clock: pthread_cond_timedwait, which GCC 11's ThreadSanitizer follows -- it does not know the
        // pthread_cond_clockwait behind wait_for and would take this thread to hold the mutex while it waits)
        else if (!cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds((long long)(timeout_s * 1e6)), ready)) return false;
        if (q.empty()) return false;
        f = std::move(q.front());
        q.pop_front();
        return true;
    }
};

// `devices[k]` = the device of worker k (as every process of this environment numbers them).  The files are partitioned by size
// (a FAST5 file is its EventDetection table and little else: bytes stand for events until a summary pass has run, and that pass
// is the workers' own), longest-processing-time first -- nchmm_lpt_partition, the rule the in-process pool and bench.py use.
static int fan_out(const Pore_Model_Dict_Type& models, const std::list<std::string>& files, const std::vector<int>& devices, Stage_Clock::Scope* whole)
{
    const std::vector<std::string> fv(files.begin(), files.end());
    const size_t n = fv.size();
    const int W = (int)devices.size();
    std::vector<uint64_t> weight(n, 1);
    for (size_t i = 0; i < n; ++i) {
        struct stat sb;
        if (stat(fv[i].c_str(), &sb) == 0 && sb.st_size > 0) weight[i] = (uint64_t)sb.st_size;
    }
    std::vector<int32_t> owner(n, 0);
    check(nchmm_lpt_partition(n, weight.data(), W, owner.data()), "nchmm_lpt_partition");
    const bool distinct = std::set<int>(devices.begin(), devices.end()).size() == devices.size();
    const char* force = std::getenv("NCHMM_POOL_FORCE_RCCL");
    const bool use_rccl = distinct && (W > 1 || (force && force[0] == '1'));
    const unsigned threads_each = std::max(1u, (opts::num_threads.get() + (unsigned)W - 1) / (unsigned)W);
    // (RCCL across processes exchanges memory handles; a host driver that only exports dmabuf handles needs this set, as bench.py's
    // launcher does -- a value the user gave stands; if the communicator cannot be formed the counters are summed on the host)
    if (use_rccl) setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);

    std::vector<std::unique_ptr<Worker_Stream>> ws;
    for (int k = 0; k < W; ++k) {
        int up[2], down[2];
        if (pipe(up) != 0) throw std::runtime_error("pipe() failed");
        if (pipe(down) != 0) { close(up[0]); close(up[1]); throw std::runtime_error("pipe() failed"); }
#ifdef F_SETPIPE_SZ
        (void)fcntl(up[1], F_SETPIPE_SZ, 1 << 20);
#endif
        std::cout.flush(); std::clog.flush();
        const pid_t pid = fork();
        if (pid < 0) { close(up[0]); close(up[1]); close(down[0]); close(down[1]); throw std::runtime_error("fork() failed"); }
        if (pid == 0) {
            // ---- worker k: nothing in this process has touched the HIP runtime yet; this one will, for device devices[k] only ----
            signal(SIGPIPE, SIG_IGN);              // (a parent that has gone shows as a failed write: Worker_Link::send leaves)
            close(up[0]); close(down[1]);
            for (const auto& o : ws) { close(o->data_fd); close(o->ctl_fd); }
            // (a worker's records travel up its pipe: nothing it or a library it loads -- RCCL's banner, a runtime's notice -- prints
            // on standard output may reach the FASTA stream the parent is writing there)
            (void)dup2(STDERR_FILENO, STDOUT_FILENO);
            logger::tag() = "[w" + std::to_string(k) + "] ";
            Worker_Link link;
            link.rank = k; link.n_ranks = W; link.device = devices[(size_t)k]; link.data_fd = up[1]; link.ctl_fd = down[0]; link.use_rccl = use_rccl;
            std::list<std::string> mine;
            for (size_t i = 0; i < n; ++i) if (owner[i] == k) { mine.push_back(fv[i]); link.global_index.push_back(i); }
            opts::num_threads.get() = threads_each;
            if (const char* e = std::getenv("NANOCALL_TEST_WORKER_ABORT"))      // test hook: the worker of that rank dies before its first record
                if (std::atoi(e) == k) abort();
            int rc = EXIT_FAILURE;
            try {
                if (mine.empty()) {
                    // (more workers than files: nothing to decode, but the reduction still counts this rank in)
                    uint64_t zero[12] = {0};
                    if (use_rccl && k == 0) {
                        uint8_t id[NCHMM_RCCL_ID_BYTES];
                        const bool have = nchmm_rccl_unique_id(id) == NCHMM_OK;
                        link.send('U', 0, have ? std::string(reinterpret_cast<const char*>(id), sizeof(id)) : std::string());
                    }
                    link.send('C', 0, std::string(reinterpret_cast<const char*>(zero), sizeof(zero)));
                    uint8_t go = 0, id[NCHMM_RCCL_ID_BYTES];
                    if (!fd_read_all(link.ctl_fd, &go, 1) || (go && !fd_read_all(link.ctl_fd, id, sizeof(id)))) std::_Exit(EXIT_FAILURE);
                    uint64_t red[8] = {0};
                    const bool ok = go && nchmm_counters_allreduce(link.device, W, k, id, red) == NCHMM_OK;
                    std::string g(1, ok ? '\1' : '\0');
                    g.append(reinterpret_cast<const char*>(red), sizeof(red));
                    link.send('G', 0, g);
                    link.send('E', 0, std::string());
                    std::_Exit(EXIT_SUCCESS);
                }
                rc = run_reads(models, mine, new Stage_Clock::Scope(stage_clock, "worker_total_s"), &link);      // (leaves from inside when all went well)
            } catch (const std::exception& e) {
                LOG(error) << e.what() << std::endl;
            }
            std::cout.flush(); std::cerr.flush(); std::clog.flush();
            std::_Exit(rc == EXIT_SUCCESS ? EXIT_FAILURE : rc);      // run_reads returning at all is an early failure
        }
        close(up[1]); close(down[0]);
        ws.emplace_back(new Worker_Stream());
        ws.back()->pid = pid; ws.back()->data_fd = up[0]; ws.back()->ctl_fd = down[1]; ws.back()->device = devices[(size_t)k];
        for (size_t i = 0; i < n; ++i) ws.back()->n_reads += owner[i] == k;
    }
    signal(SIGPIPE, SIG_IGN);                      // (a worker that died shows as a failed write on its control pipe)
    LOG(info) << "workers=" << W << " devices=[" << [&] { std::ostringstream o; for (int k = 0; k < W; ++k) o << (k ? "," : "") << devices[(size_t)k]; return o.str(); }()
              << "] threads_per_worker=" << threads_each << " counters_through=" << (use_rccl ? "rccl_allreduce" : "host_sum") << " files_per_worker=["
              << [&] { std::ostringstream o; for (int k = 0; k < W; ++k) o << (k ? "," : "") << ws[(size_t)k]->n_reads; return o.str(); }() << "]" << std::endl;
    for (auto& w : ws) { Worker_Stream* p = w.get(); p->th = std::thread([p] { p->pump(); }); }

    // ---- the output, in input order (nanocall.cpp:859-861): read i comes from worker owner[i], whose frames are ascending ----
    std::ofstream ofs, dump;
    std::ostream* os_p = &std::cout;
    if (!opts::output_fn.get().empty()) {
        ofs.open(opts::output_fn.get());
        if (!ofs) { LOG(error) << "cannot open output [" << opts::output_fn.get() << "]" << std::endl; for (auto& w : ws) kill(w->pid, SIGKILL); for (auto& w : ws) { w->th.join(); int st; waitpid(w->pid, &st, 0); } return EXIT_FAILURE; }
        os_p = &ofs;
    }
    if (!opts::dump_params_fn.get().empty()) {
        dump.open(opts::dump_params_fn.get());
        dump << "#read_id\tstrand\tmodel\tscale\tshift\tdrift\tvar\tscale_sd\tvar_sd\tp_stay\tp_skip\tlog_path_prob\trounds\tfit" << std::endl;
    }
    std::vector<char> dead((size_t)W, 0);
    std::vector<size_t> lost((size_t)W, 0);
    {
        STAGE("merge_output_s");
        for (size_t i = 0; i < n; ++i) {
            const size_t k = (size_t)owner[i];
            if (dead[k]) { ++lost[k]; continue; }
            Worker_Stream::Frame f;
            if (!ws[k]->next(f) || f.type != 'R' || f.index != i) { dead[k] = 1; ++lost[k]; continue; }
            *os_p << f.a;
            if (dump.is_open()) dump << f.b;
        }
        os_p->flush();
    }
    // ---- --stats rows, then the counters ----
    std::vector<std::string> stats_row(n);
    std::vector<std::array<uint64_t, 12>> mine((size_t)W);
    std::string unique_id;
    bool have_id = false;
    for (size_t k = 0; k < (size_t)W; ++k) {
        mine[k].fill(0);
        while (!dead[k]) {
            Worker_Stream::Frame f;
            if (!ws[k]->next(f)) { dead[k] = 1; break; }
            if (f.type == 'S' && f.index < n) stats_row[(size_t)f.index].swap(f.a);
            else if (f.type == 'U') { unique_id.swap(f.a); have_id = unique_id.size() == NCHMM_RCCL_ID_BYTES; }
            else if (f.type == 'C' && f.a.size() == sizeof(uint64_t) * 12) { std::memcpy(mine[k].data(), f.a.data(), f.a.size()); break; }
            else { dead[k] = 1; break; }
        }
    }
    const bool any_dead = std::find(dead.begin(), dead.end(), (char)1) != dead.end();
    // the verdict: every worker is alive and waiting for it -- all-reduce among them, or the sum is taken here
    const bool go = use_rccl && have_id && !any_dead;
    for (size_t k = 0; k < (size_t)W; ++k) {
        if (dead[k]) continue;
        const uint8_t b = go ? 1 : 0;
        if (!fd_write_all(ws[k]->ctl_fd, &b, 1) || (go && !fd_write_all(ws[k]->ctl_fd, unique_id.data(), unique_id.size()))) dead[k] = 1;
    }
    uint64_t red[8] = {0};
    bool reduced = go;
    std::vector<std::string> worker_stages((size_t)W);
    for (size_t k = 0; k < (size_t)W; ++k) {
        if (dead[k]) { reduced = false; continue; }
        // (a reduction that does not come back within a minute: a rank is stuck in the rendezvous -- the figures are summed here)
        Worker_Stream::Frame f;
        if (!ws[k]->next(f, go ? 60.0 : -1.0) || f.type != 'G' || f.a.size() != 1 + sizeof(red)) {
            LOG(warning) << "worker " << k << ": no answer to the counter reduction; its counters are summed on the host" << std::endl;
            reduced = false;
            kill(ws[k]->pid, SIGKILL);
            continue;
        }
        worker_stages[k] = f.b;
        if (!f.a[0]) reduced = false;
        else if (reduced) std::memcpy(red, f.a.data() + 1, sizeof(red));          // (every rank holds the same sums)
        Worker_Stream::Frame e;
        (void)ws[k]->next(e, 10.0);                                               // 'E'
    }
    uint64_t host[4] = {0, 0, 0, 0}, dev[8] = {0};
    for (size_t k = 0; k < (size_t)W; ++k) {
        host[0] += mine[k][0]; host[1] += mine[k][1];
        host[2] = std::max(host[2], mine[k][2]); host[3] = std::max(host[3], mine[k][3]);      // the stages run side by side: wall = the slowest worker's
        for (int q = 0; q < 8; ++q) dev[q] += mine[k][4 + (size_t)q];
    }
    if (reduced) std::copy(red, red + 8, dev);
    int rc = EXIT_SUCCESS;
    for (size_t k = 0; k < (size_t)W; ++k) {
        close(ws[k]->ctl_fd);
        int st = 0;
        while (waitpid(ws[k]->pid, &st, 0) < 0 && errno == EINTR) {}
        ws[k]->th.join();
        close(ws[k]->data_fd);
        const bool failed = dead[k] || !WIFEXITED(st) || WEXITSTATUS(st) != EXIT_SUCCESS;
        if (failed && (dead[k] || lost[k])) {
            // a worker that failed is reported, with what it leaves undone; it is never started again in place (its device may be
            // in any state), and the records of the other workers are all in the output
            LOG(error) << "worker " << k << " (device " << ws[k]->device << ") failed"
                       << (WIFSIGNALED(st) ? " with signal " + std::to_string(WTERMSIG(st)) : WIFEXITED(st) ? " with exit code " + std::to_string(WEXITSTATUS(st)) : std::string())
                       << ": " << lost[k] << " of its " << ws[k]->n_reads << " reads are not in the output" << std::endl;
            rc = EXIT_FAILURE;
        }
    }
    for (size_t k = 0; k < (size_t)W; ++k)
        if (!worker_stages[k].empty()) { LOG(info) << "worker " << k << " device " << ws[k]->device << " reads " << ws[k]->n_reads << " stage_wall_secs" << worker_stages[k] << std::endl; }
    LOG(info) << "counters reads=" << host[0] << " bases=" << host[1] << " strands_decoded=" << dev[0] << " events_decoded=" << dev[1]
              << " fb_windows=" << dev[4] << " fb_event_rounds=" << dev[5] << " gathered_by=" << (reduced ? "rccl_allreduce" : "host_sum")
              << " training_secs=" << host[2] / 1e6 << " basecalling_secs=" << host[3] / 1e6 << " workers=" << W << std::endl;
    delete whole;
    LOG(info) << "stage_wall_secs" << stage_clock.str() << std::endl;
    if (!opts::stats_fn.get().empty()) {   // nanocall.cpp:893-903
        std::ofstream sfs(opts::stats_fn.get());
        if (!sfs) { LOG(error) << "cannot open stats file [" << opts::stats_fn.get() << "]" << std::endl; return EXIT_FAILURE; }
        Fast5_Summary_Type::write_tsv_header(sfs);
        sfs << std::endl;
        for (size_t i = 0; i < n; ++i)
            if (!dead[(size_t)owner[i]] || !stats_row[i].empty()) sfs << stats_row[i] << std::endl;
        sfs.close();
        if (!sfs) { LOG(error) << "error writing stats file [" << opts::stats_fn.get() << "]" << std::endl; return EXIT_FAILURE; }
    }
    if (ofs.is_open()) {
        ofs.close();
        if (!ofs) { LOG(error) << "error writing output [" << opts::output_fn.get() << "]" << std::endl; return EXIT_FAILURE; }
    }
    if (dump.is_open()) dump.close();
    LOG(info) << "epoch_at_exit=" << std::fixed << epoch_now() << std::endl;
    return rc;
}

static int real_main()
{
    // (wall-clock marks for whoever times the process from outside: tools/bench_cli.py splits its wall into before / inside / after)
    LOG(info) << "epoch_at_main=" << std::fixed << epoch_now() << std::endl;
    Stage_Clock::Scope* whole = new Stage_Clock::Scope(stage_clock, "main_total_s");
    Pore_Model_Dict_Type models;
    State_Transitions_Type default_transitions;
    std::list<std::string> files;
    { STAGE("init_models_s"); init_models(models); }
    init_transitions(default_transitions);
    // Which devices, and how many processes.  One GPU (or --single-process): this process does everything, as before.  More: one
    // worker process per GPU, forked below while this process is still single-threaded and has not touched the HIP runtime --
    // so even the device count is asked by a child (started here, answered while the input list is being made).
    std::vector<int> worker_devices;
    if (const char* e = std::getenv("NANOCALL_WORKER_DEVICES"); e && *e) {      // e.g. "0,0,0,0": four workers on GPU 0 (test hook; one entry: one worker)
        std::istringstream is(e);
        std::string tok;
        while (std::getline(is, tok, ',')) worker_devices.push_back(std::atoi(tok.c_str()));
    }
    int probed = -2;
    const char* in_process_ids = std::getenv("NANOCALL_DEVICE_IDS");
    const bool may_fan_out = !opts::single_process && !(in_process_ids && *in_process_ids);
    Device_Probe probe;
    if (may_fan_out && worker_devices.empty() && opts::gpus.get() != 1 && openable_render_nodes() >= 2) probe.start();
    { STAGE("init_files_s"); init_files(files); }
    if (probe.pid >= 0) { STAGE("device_count_s"); probed = probe.finish(); }
    if (may_fan_out && worker_devices.empty() && probed >= 2) {
        const int use = opts::gpus.get() > 0 ? opts::gpus.get() : probed;
        if (use > probed) { LOG(error) << "--gpus " << use << " requested but only " << probed << " visible" << std::endl; return EXIT_FAILURE; }
        // (no more workers than input files: a run over three files on an eight-GPU node is three workers, over one file this process)
        const int n_workers = (int)std::min<size_t>((size_t)use, files.size());
        if (n_workers >= 2) for (int k = 0; k < n_workers; ++k) worker_devices.push_back(k);
        else if (use >= 2) opts::gpus.get() = 1;
    }
    if (may_fan_out && !worker_devices.empty()) return fan_out(models, files, worker_devices, whole);
    return run_reads(models, files, whole, nullptr);
}

int main(int argc, char* argv[])
{
    int rc = 0;
    if (!opts::parse(argc, argv, &rc)) return rc;
    logger::threshold() = logger::info;
    for (const auto& l : opts::log_level.get()) {   // "level" or "facility:level" (facilities are not separated here)
        const auto p = l.find(':');
        logger::threshold() = std::max(logger::threshold(), logger::parse_level(p == std::string::npos ? l : l.substr(p + 1)));
        if (p == std::string::npos) logger::threshold() = logger::parse_level(l);
    }
    Fast5_Summary_Type::verbose() = logger::threshold() >= logger::info;
    LOG(info) << "program: " << opts::program_name << std::endl;
    LOG(info) << "version: " << NANOCALL_AMD_VERSION << std::endl;
    LOG(info) << "args: " << opts::orig_argv << std::endl;
    LOG(info) << "num_threads=" << opts::num_threads.get() << std::endl;
    State_Transition_Parameters_Type::default_p_stay() = opts::pr_stay;
    State_Transition_Parameters_Type::default_p_skip() = opts::pr_skip;
    Fast5_Summary_Type::min_ed_events() = opts::min_ed_events;
    Fast5_Summary_Type::max_ed_events() = opts::max_ed_events;
    Fast5_Summary_Type::eventdetection_group() = opts::ed_group;
    Fast5_Summary_Type::template_only() = opts::template_only;
    Fast5_Summary_Type::trim_margins() = {{opts::trim_ed_sq_start, opts::trim_ed_sq_end, opts::trim_ed_hp_start, opts::trim_ed_hp_end}};
    Fast5_Summary_Type::ed_cache_budget() = (size_t)opts::ed_cache_mb.get() << 20;
    LOG(info) << "eventdetection_group=" << (Fast5_Summary_Type::eventdetection_group().empty() ? std::string("smallest") : Fast5_Summary_Type::eventdetection_group()) << std::endl;
    // pore-related options, nanocall.cpp:936-970
    if (!opts::train_drift.get().empty() && opts::train_drift.get() != "0" && opts::train_drift.get() != "1") {
        LOG(error) << "train-drift not understdood: " << opts::train_drift.get() << std::endl;
        return EXIT_FAILURE;
    }
    if (opts::pore.get() == "r9") {
        Fast5_Summary_Type::abasic_level_top_percent() = 1.0;
        Fast5_Summary_Type::abasic_level_top_offset() = 0.0;
        Fast5_Summary_Type::hairpin_island_window_size() = 10;
        Fast5_Summary_Type::hairpin_island_window_load() = 5;
        if (opts::train_drift.get().empty()) opts::train_drift.get() = "0";
    } else if (opts::pore.get() == "r73") {
        Fast5_Summary_Type::abasic_level_top_percent() = 1.0;
        Fast5_Summary_Type::abasic_level_top_offset() = 5.0;
        Fast5_Summary_Type::hairpin_island_window_size() = 5;
        Fast5_Summary_Type::hairpin_island_window_load() = 5;
        if (opts::train_drift.get().empty()) opts::train_drift.get() = "1";
    } else {
        LOG(error) << "unknown pore type: " << opts::pore.get() << std::endl;
        return EXIT_FAILURE;
    }
    Parameter_Trainer<float, 6>::pm_train_drift() = opts::train_drift.get() == "1";
    LOG(info) << "ed_event_trimming: " << " sq_start=" << Fast5_Summary_Type::trim_margins()[0] << " sq_end=" << Fast5_Summary_Type::trim_margins()[1]
              << " hp_start=" << Fast5_Summary_Type::trim_margins()[2] << " hp_end=" << Fast5_Summary_Type::trim_margins()[3] << std::endl;
    if (!opts::template_only.get())
        LOG(info) << "hairpin_detection:" << " abasic_level_top_percent=" << Fast5_Summary_Type::abasic_level_top_percent()
                  << " abasic_level_top_offset=" << Fast5_Summary_Type::abasic_level_top_offset()
                  << " hairpin_island_window_size=" << Fast5_Summary_Type::hairpin_island_window_size()
                  << " hairpin_island_window_load=" << Fast5_Summary_Type::hairpin_island_window_load() << std::endl;
    else
        LOG(info) << "hairpin_detection: disabled" << std::endl;
    // training / basecalling switches, nanocall.cpp:995-1038
    if (opts::train && opts::no_train) { LOG(error) << "either --train or --no-train may be used, but not both" << std::endl; return EXIT_FAILURE; }
    else if (!opts::train && !opts::no_train) opts::train.set(true);
    if (opts::basecall && opts::no_basecall) { LOG(error) << "either --basecall or --no-basecall may be used, but not both" << std::endl; return EXIT_FAILURE; }
    else if (!opts::basecall && !opts::no_basecall) opts::basecall.set(true);
    if (opts::train && !opts::no_train_scaling) {
        if (opts::single_strand_scaling && opts::double_strand_scaling) {
            LOG(error) << "either --single-strand-scaling or --double-strand-scaling may be used, but not both" << std::endl;
            return EXIT_FAILURE;
        } else if (!opts::single_strand_scaling && !opts::double_strand_scaling) {
            opts::double_strand_scaling.set(true);
        }
    }
    if (opts::scaling_select_threshold.get() < 0.0) { LOG(error) << "invalid scaling_select_threshold: " << opts::scaling_select_threshold.get() << std::endl; return EXIT_FAILURE; }
    if (opts::scaling_min_progress.get() < 0.0) { LOG(error) << "invalid scaling_min_progress: " << opts::scaling_min_progress.get() << std::endl; return EXIT_FAILURE; }
    if (!opts::output_fn.get().empty() && opts::write_fast5) {
        LOG(error) << "output may be written to fast5 files or to a single output file, but not both" << std::endl;
        return EXIT_FAILURE;
    }
    if (opts::write_fast5) { LOG(error) << "--write-fast5 is not supported by this build (FASTA output only)" << std::endl; return EXIT_FAILURE; }
    LOG(info) << "train=" << opts::train.get() << std::endl;
    if (opts::train) {
        LOG(info) << "train_scaling=" << !opts::no_train_scaling.get() << std::endl;
        LOG(info) << "train_transitions=" << !opts::no_train_transitions.get() << std::endl;
        if (!opts::no_train_scaling) {
            LOG(info) << "double_strands_scaling=" << opts::double_strand_scaling.get() << std::endl;
            LOG(info) << "scaling_num_events=" << opts::scaling_num_events.get() << std::endl;
            LOG(info) << "scaling_max_rounds=" << opts::scaling_max_rounds.get() << std::endl;
            LOG(info) << "scaling_min_progress=" << opts::scaling_min_progress.get() << std::endl;
            LOG(info) << "scaling_select_threshold=" << opts::scaling_select_threshold.get() << std::endl;
            LOG(info) << "train_drift=" << opts::train_drift.get() << std::endl;
        }
    }
    LOG(info) << "basecall=" << opts::basecall.get() << std::endl;
    try {
        rc = real_main();
    } catch (const std::exception& e) {
        LOG(error) << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    // (a successful run has left from real_main already; what comes back here failed early or wants the full exit)
    std::cout.flush(); std::cerr.flush(); std::clog.flush();
    if (!std::getenv("NANOCALL_FULL_EXIT")) std::_Exit(rc);
    return rc;
}
