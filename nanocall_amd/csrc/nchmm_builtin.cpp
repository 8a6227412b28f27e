// nchmm_builtin.cpp -- the six builtin pore-model tables inside the library, so that a host program (the nanocall CLI)
// needs no data file at run time.  Counterpart of Builtin_Model (src/nanocall/Builtin_Model.{hpp,cpp},
// src/builtin_models/builtin_model_{names,strands,init_lists}.inl); the tables themselves are the data file
// nanocall_amd/data/builtin_models.f32 (tools/extract_builtin_models.py), pulled in with .incbin below.
#include "nanocall_hip.h"

extern "C" {
extern const unsigned char nchmm_builtin_blob[];
extern const unsigned char nchmm_builtin_blob_end[];
}

#define NCHMM_STR2(x) #x
#define NCHMM_STR(x) NCHMM_STR2(x)
__asm__(".section .rodata\n"
        ".balign 64\n"
        ".global nchmm_builtin_blob\n"
        "nchmm_builtin_blob:\n"
        ".incbin \"" NCHMM_STR(NCHMM_BUILTIN_F32) "\"\n"
        ".global nchmm_builtin_blob_end\n"
        "nchmm_builtin_blob_end:\n"
        ".previous\n");

namespace {
// src/builtin_models/builtin_model_names.inl:1-13 / builtin_model_strands.inl:1-13 (also in nanocall_amd/data/builtin_models.json;
// tests/test_host_prep.py compares the two)
const char* const kNames[] = {"r73.t.006.ont.model",  "r73.c.p1.006.ont.model", "r73.c.p2.006.ont.model",
                              "r9.t.007.ont.model",   "r9.c.p1.007.ont.model",  "r9.c.p2.007.ont.model"};
const int kStrands[] = {0, 1, 1, 0, 1, 1};
constexpr int kNum = 6;
constexpr size_t kTableBytes = (size_t)NCHMM_N_STATES * 4 * sizeof(float);
}  // namespace

extern "C" {

int nchmm_builtin_count(void)
{
    return (size_t)(nchmm_builtin_blob_end - nchmm_builtin_blob) == kNum * kTableBytes ? kNum : 0;
}
const char* nchmm_builtin_name(int i) { return (i >= 0 && i < kNum) ? kNames[i] : nullptr; }
int nchmm_builtin_strand(int i) { return (i >= 0 && i < kNum) ? kStrands[i] : -1; }
const float* nchmm_builtin_table(int i)
{
    if (i < 0 || i >= nchmm_builtin_count()) return nullptr;
    return reinterpret_cast<const float*>(nchmm_builtin_blob + (size_t)i * kTableBytes);
}

}  // extern "C"
