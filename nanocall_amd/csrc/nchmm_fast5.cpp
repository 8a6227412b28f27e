// nchmm_fast5.cpp -- FAST5 ingest through the HDF5 C API, loaded at run time (see include/nanocall_fast5.h).
#include "nanocall_fast5.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#ifdef NCHMM_HAVE_HDF5_HEADERS
#include <hdf5.h>

namespace {

thread_local std::string g_err;
// H5F_ACC_RDONLY without the header's H5check()/H5open() prefix calls (those would be link-time references)
constexpr unsigned kAccRdOnly = 0x0000u;

// the entry points used, resolved from libhdf5 at first use
#define H5_FUNCS(X)                                                                                                      \
    X(H5open) X(H5Eset_auto2) X(H5Fis_hdf5) X(H5Fopen) X(H5Fclose) X(H5Lexists) X(H5Gopen2) X(H5Gclose) X(H5Literate)   \
    X(H5Aexists) X(H5Aopen) X(H5Aread) X(H5Aget_type) X(H5Aclose) X(H5Tget_class) X(H5Tis_variable_str) X(H5Tget_size)  \
    X(H5Tcopy) X(H5Tset_size) X(H5Tclose) X(H5Dopen2) X(H5Dget_space) X(H5Sget_simple_extent_npoints) X(H5Dget_type)    \
    X(H5Tget_member_index) X(H5Tcreate) X(H5Tinsert) X(H5Dread) X(H5Dclose) X(H5Sclose) X(H5free_memory)                \
    X(H5Tset_cset) X(H5Tset_strpad) X(H5get_libversion)

// The struct layouts, the width of hid_t and the H5Literate callback signature are those of the HEADERS this file was
// compiled against; the library is whatever dlopen finds at run time, and H5check() -- the guard the headers normally
// plant -- is bypassed on purpose (no link-time reference).  With 1.12+ headers `H5Literate` is a macro for H5Literate2
// (another callback type): the X-macro above would then declare a member of the new type and look the OLD symbol up by
// its spelled-out name.  So: refuse such headers at compile time, and refuse a library of another major.minor at run time.
#if H5_VERS_MAJOR != 1 || H5_VERS_MINOR > 10
#error "nchmm_fast5.cpp is written against the HDF5 1.8 / 1.10 API (H5Literate with H5L_info_t); build with those headers or extend H5_FUNCS to the versioned symbols"
#endif

struct Hdf5 {
    void* handle = nullptr;
#define X(f) decltype(&::f) f = nullptr;
    H5_FUNCS(X)
#undef X
    hid_t t_double = -1, t_int64 = -1, t_c_s1 = -1;
    bool ok = false;
    std::string why;

    Hdf5()
    {
        std::vector<std::string> names;
        if (const char* e = std::getenv("NCHMM_HDF5_LIB")) names.push_back(e);
        for (const char* n : {"libhdf5.so", "libhdf5.so.103", "libhdf5_serial.so", "libhdf5_serial.so.103", "/opt/conda/lib/libhdf5.so",
                              "/opt/conda/lib/libhdf5.so.103", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so"})
            names.push_back(n);
        for (const auto& n : names) {
            handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { why = "libhdf5 not found (set NCHMM_HDF5_LIB)"; return; }
#define X(f)                                                              \
    f = reinterpret_cast<decltype(f)>(dlsym(handle, #f));                 \
    if (!f) { why = std::string("libhdf5 lacks ") + #f; return; }
        H5_FUNCS(X)
#undef X
        unsigned maj = 0, min = 0, rel = 0;
        if (H5get_libversion(&maj, &min, &rel) < 0 || maj != (unsigned)H5_VERS_MAJOR || min != (unsigned)H5_VERS_MINOR) {
            why = "libhdf5 " + std::to_string(maj) + "." + std::to_string(min) + "." + std::to_string(rel) + " found at run time, but this library was compiled against the " +
                  std::to_string(H5_VERS_MAJOR) + "." + std::to_string(H5_VERS_MINOR) + " headers (set NCHMM_HDF5_LIB to a matching libhdf5)";
            return;
        }
        if (H5open() < 0) { why = "H5open failed"; return; }
        auto glob = [&](const char* name) -> hid_t {
            void* p = dlsym(handle, name);
            return p ? *reinterpret_cast<hid_t*>(p) : (hid_t)-1;
        };
        t_double = glob("H5T_NATIVE_DOUBLE_g"); t_int64 = glob("H5T_NATIVE_INT64_g"); t_c_s1 = glob("H5T_C_S1_g");
        if (t_double < 0 || t_int64 < 0 || t_c_s1 < 0) { why = "libhdf5 lacks the native type ids"; return; }
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);   // errors come back as return codes; no stack dump on stderr
        ok = true;
    }
};

Hdf5& h5()
{
    static Hdf5 lib;
    return lib;
}

int fail(const std::string& msg) { g_err = msg; return NCHMM_E_IO; }

// One file at a time.  The thread-safe HDF5 build serialises every API call on its own global lock anyway; with a few
// dozen host threads contending for it call by call, 2000 files took 1.6 s instead of 0.43 s from one thread
// (profiles/r02_bench_cli.json).  Taking one mutex per FILE keeps the callers' other work (segmentation, scaling) parallel
// and the HDF5 part at its single-thread speed.
std::mutex& file_mutex()
{
    static std::mutex m;
    return m;
}

struct Names { std::vector<std::string> v; };
herr_t collect(hid_t, const char* name, const H5L_info_t*, void* data)
{
    static_cast<Names*>(data)->v.emplace_back(name);
    return 0;
}

std::vector<std::string> children(hid_t loc, const char* path)
{
    Names nm;
    Hdf5& L = h5();
    if (L.H5Lexists(loc, path, H5P_DEFAULT) <= 0) return nm.v;
    const hid_t g = L.H5Gopen2(loc, path, H5P_DEFAULT);
    if (g < 0) return nm.v;
    hsize_t idx = 0;
    L.H5Literate(g, H5_INDEX_NAME, H5_ITER_INC, &idx, collect, &nm);
    L.H5Gclose(g);
    std::sort(nm.v.begin(), nm.v.end());
    return nm.v;
}

bool path_exists(hid_t f, const std::string& path)
{
    // H5Lexists wants every intermediate link to exist
    Hdf5& L = h5();
    size_t pos = 1;
    while (true) {
        pos = path.find('/', pos);
        const std::string sub = path.substr(0, pos);
        if (L.H5Lexists(f, sub.c_str(), H5P_DEFAULT) <= 0) return false;
        if (pos == std::string::npos) return true;
        ++pos;
    }
}

bool read_number_attr(hid_t obj, const char* name, double* out)
{
    Hdf5& L = h5();
    if (L.H5Aexists(obj, name) <= 0) return false;
    const hid_t a = L.H5Aopen(obj, name, H5P_DEFAULT);
    if (a < 0) return false;
    const herr_t rc = L.H5Aread(a, L.t_double, out);   // HDF5 converts integer / float storage types
    L.H5Aclose(a);
    return rc >= 0;
}

bool read_string_attr(hid_t obj, const char* name, std::string* out)
{
    Hdf5& L = h5();
    if (L.H5Aexists(obj, name) <= 0) return false;
    const hid_t a = L.H5Aopen(obj, name, H5P_DEFAULT);
    if (a < 0) return false;
    bool ok = false;
    const hid_t ft = L.H5Aget_type(a);
    if (ft >= 0 && L.H5Tget_class(ft) == H5T_STRING) {
        if (L.H5Tis_variable_str(ft) > 0) {
            char* p = nullptr;
            const hid_t mt = L.H5Tcopy(L.t_c_s1);
            L.H5Tset_size(mt, H5T_VARIABLE);
            if (L.H5Aread(a, mt, &p) >= 0 && p) { *out = p; L.H5free_memory(p); ok = true; }
            L.H5Tclose(mt);
        } else {
            const size_t sz = L.H5Tget_size(ft);
            std::vector<char> buf(sz + 1, 0);
            const hid_t mt = L.H5Tcopy(L.t_c_s1);
            L.H5Tset_size(mt, sz + 1);
            if (L.H5Aread(a, mt, buf.data()) >= 0) { *out = buf.data(); ok = true; }
            L.H5Tclose(mt);
        }
    }
    if (ft >= 0) L.H5Tclose(ft);
    L.H5Aclose(a);
    return ok;
}

}  // namespace

extern "C" {

int nchmm_fast5_available(void) { return h5().ok ? 1 : 0; }

const char* nchmm_fast5_last_error(void) { return g_err.c_str(); }

int nchmm_fast5_is_valid_file(const char* path)
{
    if (!path || !h5().ok) return 0;
    Hdf5& L = h5();
    std::lock_guard<std::mutex> one_file(file_mutex());
    if (L.H5Fis_hdf5(path) <= 0) return 0;
    const hid_t f = L.H5Fopen(path, kAccRdOnly, H5P_DEFAULT);
    if (f < 0) return 0;
    L.H5Fclose(f);
    return 1;
}

void nchmm_fast5_release(nchmm_fast5_read* r)
{
    if (!r) return;
    std::free(r->events);
    r->events = nullptr;
    r->n_events = 0;
}

int nchmm_fast5_load(const char* path, const char* ed_group, nchmm_fast5_read* out)
{
    if (!path || !out) return NCHMM_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (!h5().ok) return fail(h5().why);
    Hdf5& L = h5();
    std::lock_guard<std::mutex> one_file(file_mutex());
    const hid_t f = L.H5Fopen(path, kAccRdOnly, H5P_DEFAULT);
    if (f < 0) return fail(std::string(path) + ": cannot open as HDF5");
    int rc = NCHMM_OK;
    do {
        // sampling rate (Fast5_Summary.hpp:162-167)
        if (path_exists(f, "/UniqueGlobalKey/channel_id")) {
            const hid_t g = L.H5Gopen2(f, "/UniqueGlobalKey/channel_id", H5P_DEFAULT);
            if (g >= 0) {
                double v = 0;
                if (read_number_attr(g, "sampling_rate", &v)) { out->have_sampling_rate = 1; out->sampling_rate = v; }
                L.H5Gclose(g);
            }
        }
        // EventDetection group: the requested one, or the smallest name present (:86-90, nanocall.cpp:56,927)
        std::string grp = ed_group ? ed_group : "";
        if (grp.empty()) {
            for (const auto& n : children(f, "/Analyses"))
                if (n.compare(0, 15, "EventDetection_") == 0) { grp = n.substr(15); break; }
            if (grp.empty()) break;   // no events: have_events stays 0
        }
        std::snprintf(out->ed_group, sizeof(out->ed_group), "%s", grp.c_str());
        const std::string reads_path = "/Analyses/EventDetection_" + grp + "/Reads";
        if (!path_exists(f, reads_path)) break;
        const std::vector<std::string> reads = children(f, reads_path.c_str());
        if (reads.empty()) break;
        const std::string read_path = reads_path + "/" + reads.front();
        if (!path_exists(f, read_path + "/Events")) break;
        std::snprintf(out->read_name, sizeof(out->read_name), "%s", reads.front().c_str());
        {
            const hid_t g = L.H5Gopen2(f, read_path.c_str(), H5P_DEFAULT);
            if (g >= 0) {
                std::string id;
                if (read_string_attr(g, "read_id", &id)) std::snprintf(out->read_id, sizeof(out->read_id), "%s", id.c_str());
                L.H5Gclose(g);
            }
        }
        const hid_t d = L.H5Dopen2(f, (read_path + "/Events").c_str(), H5P_DEFAULT);
        if (d < 0) { rc = fail(std::string(path) + ": cannot open " + read_path + "/Events"); break; }
        const hid_t sp = L.H5Dget_space(d), ft = L.H5Dget_type(d);
        const hssize_t n = sp >= 0 ? L.H5Sget_simple_extent_npoints(sp) : -1;
        const bool compound = ft >= 0 && L.H5Tget_class(ft) == H5T_COMPOUND;
        const bool has_stdv = compound && L.H5Tget_member_index(ft, "stdv") >= 0;
        const bool has_var = compound && L.H5Tget_member_index(ft, "variance") >= 0;
        if (n < 0 || !compound || L.H5Tget_member_index(ft, "mean") < 0 || L.H5Tget_member_index(ft, "start") < 0
            || L.H5Tget_member_index(ft, "length") < 0 || (!has_stdv && !has_var)) {
            rc = fail(std::string(path) + ": " + read_path + "/Events is not an event table (mean, stdv|variance, start, length)");
        } else {
            // memory type = nchmm_ed_event; HDF5 converts each member from whatever the file stores
            const hid_t mt = L.H5Tcreate(H5T_COMPOUND, sizeof(nchmm_ed_event));
            L.H5Tinsert(mt, "mean", offsetof(nchmm_ed_event, mean), L.t_double);
            L.H5Tinsert(mt, has_stdv ? "stdv" : "variance", offsetof(nchmm_ed_event, stdv), L.t_double);
            L.H5Tinsert(mt, "start", offsetof(nchmm_ed_event, start), L.t_int64);
            L.H5Tinsert(mt, "length", offsetof(nchmm_ed_event, length), L.t_int64);
            out->events = static_cast<nchmm_ed_event*>(std::malloc(sizeof(nchmm_ed_event) * (size_t)std::max<hssize_t>(n, 1)));
            if (!out->events) rc = NCHMM_E_NOMEM;
            else if (n > 0 && L.H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out->events) < 0)
                rc = fail(std::string(path) + ": reading " + read_path + "/Events failed");
            else {
                out->n_events = (size_t)n;
                if (!has_stdv)
                    for (size_t i = 0; i < out->n_events; ++i) out->events[i].stdv = std::sqrt(out->events[i].stdv);
                out->have_events = 1;
            }
            L.H5Tclose(mt);
        }
        if (ft >= 0) L.H5Tclose(ft);
        if (sp >= 0) L.H5Sclose(sp);
        L.H5Dclose(d);
    } while (false);
    L.H5Fclose(f);
    if (rc != NCHMM_OK) nchmm_fast5_release(out);
    return rc;
}

}  // extern "C"

#else   // built without the HDF5 headers: FAST5 ingest is unavailable, everything else works

extern "C" {
int nchmm_fast5_available(void) { return 0; }
const char* nchmm_fast5_last_error(void) { return "built without HDF5 headers"; }
int nchmm_fast5_is_valid_file(const char*) { return 0; }
void nchmm_fast5_release(nchmm_fast5_read* r) { if (r) { r->events = nullptr; r->n_events = 0; } }
int nchmm_fast5_load(const char*, const char*, nchmm_fast5_read* out)
{
    if (out) std::memset(out, 0, sizeof(*out));
    return NCHMM_E_IO;
}
}
#endif
