/*
 * nanocall_fast5.h -- FAST5 (HDF5) ingest for the MI355X-native nanocall core: the few calls of the fast5::File
 * class that Fast5_Summary makes (src/nanocall/Fast5_Summary.hpp:154-184,505-525; fast5.hpp itself is an
 * un-vendored submodule of the reference), as a plain C ABI in libnanocall_hip.so.
 *
 *   fast5::File::is_valid_file(fn)                        -> nchmm_fast5_is_valid_file        (nanocall.cpp:212,225,247)
 *   f.open / have_sampling_rate / get_sampling_rate       -> nchmm_fast5_load: sampling_rate  (Fast5_Summary.hpp:160-172)
 *   have_eventdetection_events(group)                     -> ... have_events                  (:174-178)
 *   get_eventdetection_event_params(group).read_id        -> ... read_id                      (:179-183)
 *   get_eventdetection_events(group)                      -> ... events, n_events             (:505-509)
 *
 * File layout read (the ONT FAST5 layout the fast5 library wraps):
 *   /UniqueGlobalKey/channel_id            attribute sampling_rate (float or integer)
 *   /Analyses/EventDetection_<grp>/Reads/<Read_N>
 *        attribute read_id (string, optional)
 *        dataset   Events: compound with members mean, start, length and stdv (or variance, whose square root is
 *                  taken); any float / integer member types (converted by HDF5 to double / int64)
 * <grp> = the requested --ed-group, or the smallest group name present; <Read_N> = the first read of the group.
 *
 * libhdf5 is loaded at run time (dlopen; NCHMM_HDF5_LIB overrides the search list), so the library has no link
 * dependency on it: without HDF5 every call here fails with NCHMM_E_IO and nchmm_fast5_available() is 0.
 * --write-fast5 (Fast5_Summary::add_basecall_*, :379-437) is not provided.
 */
#ifndef NANOCALL_FAST5_H
#define NANOCALL_FAST5_H

#include "nanocall_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NCHMM_E_IO (-7)   /* HDF5 missing / file unreadable: what the reference reports as hdf5_tools::Exception (:311-315) */

typedef struct nchmm_fast5_read {
    int32_t have_sampling_rate;     /* fast5::File::have_sampling_rate() */
    int32_t have_events;            /* have_eventdetection_events(group) */
    double sampling_rate;
    char ed_group[32];              /* the EventDetection group used ("000", ...) */
    char read_name[64];             /* "Read_<N>" */
    char read_id[256];              /* "" when the attribute is absent */
    size_t n_events;
    nchmm_ed_event* events;         /* owned by the struct: nchmm_fast5_release() */
} nchmm_fast5_read;

int nchmm_fast5_available(void);
int nchmm_fast5_is_valid_file(const char* path);            /* 1 / 0 */
int nchmm_fast5_load(const char* path, const char* ed_group /* NULL or "": smallest available */, nchmm_fast5_read* out);
void nchmm_fast5_release(nchmm_fast5_read* r);
const char* nchmm_fast5_last_error(void);                   /* of the calling thread */

#ifdef __cplusplus
}
#endif
#endif
