// make_fast5.cpp -- TEST TOOL: write a minimal ONT-layout FAST5 (HDF5) from an EventDetection table in the
// `#nanocall-events` text form, so that the FAST5 ingest (nanocall_fast5.h) and the CLI can be tested without
// lab data (the reference ships no FAST5; there is no h5py in the image).  Links libhdf5 directly.
//
//   make_fast5 [--variance] [--f32] [--no-rate] [--no-read-id] [--ed-group 000] [--read-number 7] in.events out.fast5
//
// Layout written (what fast5::File reads for nanocall):
//   /                                  attr file_version (double)
//   /UniqueGlobalKey/channel_id        attr sampling_rate (double)   [unless --no-rate]
//   /Analyses/EventDetection_<grp>/Reads/Read_<N>
//        attr read_id (fixed-length string) [unless --no-read-id], read_number (uint32), start_time (int64)
//        dataset Events: compound {start: int64, length: int64, mean: f64|f32, stdv|variance: f64|f32}
#include <hdf5.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

struct Row { long long start, length; double mean, sd; };
struct RowF { long long start, length; float mean, sd; };

int main(int argc, char** argv)
{
    bool variance = false, f32 = false, no_rate = false, no_id = false;
    std::string grp = "000", in, out;
    unsigned read_number = 7;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--variance") variance = true;
        else if (a == "--f32") f32 = true;
        else if (a == "--no-rate") no_rate = true;
        else if (a == "--no-read-id") no_id = true;
        else if (a == "--ed-group" && i + 1 < argc) grp = argv[++i];
        else if (a == "--read-number" && i + 1 < argc) read_number = (unsigned)std::atoi(argv[++i]);
        else if (in.empty()) in = a;
        else out = a;
    }
    if (in.empty() || out.empty()) { std::cerr << "usage: make_fast5 [options] in.events out.fast5\n"; return 2; }
    std::ifstream is(in);
    if (!is) { std::cerr << "cannot open " << in << "\n"; return 1; }
    double rate = 4000.0;
    std::string read_id, line;
    std::vector<Row> rows;
    while (std::getline(is, line)) {
        if (line.empty()) continue;
        if (line[0] == '#') {
            std::istringstream ls(line.substr(1));
            std::string key; ls >> key;
            if (key == "sampling_rate") ls >> rate;
            else if (key == "read_id") ls >> read_id;
            continue;
        }
        std::istringstream ls(line);
        Row r;
        if (ls >> r.mean >> r.sd >> r.start >> r.length) { if (variance) r.sd = r.sd * r.sd; rows.push_back(r); }
    }
    const hid_t f = H5Fcreate(out.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) { std::cerr << "cannot create " << out << "\n"; return 1; }
    const hid_t scalar = H5Screate(H5S_SCALAR);
    auto attr = [&](hid_t obj, const char* name, hid_t type, const void* v) {
        const hid_t a = H5Acreate2(obj, name, type, scalar, H5P_DEFAULT, H5P_DEFAULT);
        H5Awrite(a, type, v);
        H5Aclose(a);
    };
    const double version = 0.6;
    attr(f, "file_version", H5T_NATIVE_DOUBLE, &version);
    const hid_t lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(lcpl, 1);
    const hid_t ch = H5Gcreate2(f, "/UniqueGlobalKey/channel_id", lcpl, H5P_DEFAULT, H5P_DEFAULT);
    if (!no_rate) attr(ch, "sampling_rate", H5T_NATIVE_DOUBLE, &rate);
    H5Gclose(ch);
    const std::string rp = "/Analyses/EventDetection_" + grp + "/Reads/Read_" + std::to_string(read_number);
    const hid_t rg = H5Gcreate2(f, rp.c_str(), lcpl, H5P_DEFAULT, H5P_DEFAULT);
    if (!no_id && !read_id.empty()) {
        const hid_t st = H5Tcopy(H5T_C_S1);
        H5Tset_size(st, read_id.size() + 1);
        attr(rg, "read_id", st, read_id.c_str());
        H5Tclose(st);
    }
    attr(rg, "read_number", H5T_NATIVE_UINT, &read_number);
    const long long start_time = rows.empty() ? 0 : rows.front().start;
    attr(rg, "start_time", H5T_NATIVE_LLONG, &start_time);
    const hsize_t dims[1] = {rows.size()};
    const hid_t sp = H5Screate_simple(1, dims, nullptr);
    const char* sd_name = variance ? "variance" : "stdv";
    if (f32) {
        std::vector<RowF> rf(rows.size());
        for (size_t i = 0; i < rows.size(); ++i) rf[i] = RowF{rows[i].start, rows[i].length, (float)rows[i].mean, (float)rows[i].sd};
        const hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(RowF));
        H5Tinsert(t, "start", offsetof(RowF, start), H5T_NATIVE_LLONG); H5Tinsert(t, "length", offsetof(RowF, length), H5T_NATIVE_LLONG);
        H5Tinsert(t, "mean", offsetof(RowF, mean), H5T_NATIVE_FLOAT); H5Tinsert(t, sd_name, offsetof(RowF, sd), H5T_NATIVE_FLOAT);
        const hid_t d = H5Dcreate2(rg, "Events", t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (!rf.empty()) H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, rf.data());
        H5Dclose(d); H5Tclose(t);
    } else {
        const hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(Row));
        H5Tinsert(t, "start", offsetof(Row, start), H5T_NATIVE_LLONG); H5Tinsert(t, "length", offsetof(Row, length), H5T_NATIVE_LLONG);
        H5Tinsert(t, "mean", offsetof(Row, mean), H5T_NATIVE_DOUBLE); H5Tinsert(t, sd_name, offsetof(Row, sd), H5T_NATIVE_DOUBLE);
        const hid_t d = H5Dcreate2(rg, "Events", t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (!rows.empty()) H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, rows.data());
        H5Dclose(d); H5Tclose(t);
    }
    H5Sclose(sp); H5Gclose(rg); H5Pclose(lcpl); H5Sclose(scalar); H5Fclose(f);
    return 0;
}
