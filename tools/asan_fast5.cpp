// asan_fast5.cpp -- the FAST5 reader (nchmm_fast5.cpp: HDF5 C API, dlopen'ed) under AddressSanitizer + UBSan.
//   asan_fast5 <file>...   every path goes through is_valid_file / load / release; damaged files must come back as an
// error code or as "no events", never as a crash or a leak in OUR code (libhdf5 itself is the system's build).
//   make -C tools asan-fast5   (tests/test_fast5_ingest.py::test_fast5_reader_under_sanitizers feeds it the fixtures,
// truncations of them and byte-flipped copies)
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include "nanocall_fast5.h"
#include "nanocall_hip.h"
int main(int argc, char** argv)
{
    if (!nchmm_fast5_available()) { puts("hdf5 unavailable"); return 0; }
    size_t loaded = 0, refused = 0, events = 0;
    for (int i = 1; i < argc; ++i) {
        const int valid = nchmm_fast5_is_valid_file(argv[i]);
        for (const char* grp : {(const char*)nullptr, "", "000", "999", "a-group-name-that-is-far-too-long-for-the-field"}) {
            nchmm_fast5_read r;
            const int rc = nchmm_fast5_load(argv[i], grp, &r);
            if (rc == NCHMM_OK) {
                ++loaded;
                if (r.have_events) { events += r.n_events; volatile double touch = 0; for (size_t k = 0; k < r.n_events; ++k) touch += r.events[k].mean + (double)r.events[k].length; }
                if (std::strlen(r.read_id) >= sizeof(r.read_id) || std::strlen(r.ed_group) >= sizeof(r.ed_group)) return 4;
            } else {
                ++refused;
                if (!nchmm_fast5_last_error()) return 5;
            }
            nchmm_fast5_release(&r);
            nchmm_fast5_release(&r);      // idempotent
        }
        (void)valid;
    }
    std::printf("fast5 reader under ASan/UBSan: ok (%zu loads, %zu refusals, %zu events)\n", loaded, refused, events);
    return 0;
}
