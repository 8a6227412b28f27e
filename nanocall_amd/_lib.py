"""ctypes loader for libnanocall_hip.so (the C-ABI in include/nanocall_hip.h).

The library is the product; this module is only a binding.  There is no fallback: if the shared
object is missing, importing any device entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnanocall_hip.so")

_lib = None

c_f32p = C.POINTER(C.c_float)
c_u64p = C.POINTER(C.c_uint64)
c_u32p = C.POINTER(C.c_uint32)
c_u16p = C.POINTER(C.c_uint16)
c_i32p = C.POINTER(C.c_int32)
vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/nanocall_hip.h one to one
SIGNATURES = {
    "nchmm_strerror": (C.c_char_p, [C.c_int]),
    "nchmm_abi_version": (C.c_int, []),
    "nchmm_builtin_count": (C.c_int, []),
    "nchmm_builtin_name": (C.c_char_p, [C.c_int]),
    "nchmm_builtin_strand": (C.c_int, [C.c_int]),
    "nchmm_builtin_table": (C.POINTER(C.c_float), [C.c_int]),
    "nchmm_model_load": (C.c_int, [vp, vp]),
    "nchmm_model_scale": (C.c_int, [vp, vp]),
    "nchmm_model_pack6": (C.c_int, [vp, vp]),
    "nchmm_transitions_fast": (C.c_int, [C.c_float, C.c_float, vp, vp, vp, vp]),
    "nchmm_events_prepare": (C.c_int, [C.c_size_t, vp, vp, vp, C.c_float, vp, vp]),
    "nchmm_base_seq": (C.c_int, [C.c_size_t, vp, vp, vp, vp]),
    "nchmm_write_fasta": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint, vp, C.c_size_t, vp]),
    "nchmm_st_train_kmers": (C.c_int, [vp, vp]),
    "nchmm_train_pm_finish": (C.c_int, [C.c_size_t, vp, vp, vp, vp, C.c_int, vp, vp, vp]),
    "nchmm_train_st_finish": (C.c_int, [C.c_size_t, vp, vp, vp]),
    "nchmm_train_pm_solve": (C.c_int, [C.c_size_t, vp, C.c_int, vp, vp, vp]),
    "nchmm_segment_opts_default": (C.c_int, [vp, C.c_char_p]),
    "nchmm_mean_stdv": (C.c_int, [C.c_size_t, vp, vp, vp]),
    "nchmm_read_summarize": (C.c_int, [vp, C.c_size_t, vp, C.c_float, C.c_int, vp]),
    "nchmm_read_load_events": (C.c_int, [vp, vp, C.c_float, C.c_int, vp, vp, vp, vp, vp]),
    "nchmm_initial_scaling": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp]),
    "nchmm_fast5_available": (C.c_int, []),
    "nchmm_fast5_is_valid_file": (C.c_int, [C.c_char_p]),
    "nchmm_fast5_load": (C.c_int, [C.c_char_p, C.c_char_p, vp]),
    "nchmm_fast5_release": (None, [vp]),
    "nchmm_fast5_last_error": (C.c_char_p, []),
    "nchmm_train_opts_default": (C.c_int, [vp]),
    "nchmm_train_enumerate": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, vp, vp, vp]),
    "nchmm_train_reads": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t] + [vp] * 8),
    "nchmm_basecall_reads": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t] + [vp] * 9),
    "nchmm_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "nchmm_destroy": (C.c_int, [vp]),
    "nchmm_last_hip_error": (C.c_int, [vp]),
    "nchmm_set_stream": (C.c_int, [vp, vp]),
    "nchmm_use_own_stream": (C.c_int, [vp]),
    "nchmm_synchronize": (C.c_int, [vp]),
    "nchmm_put_model": (C.c_int, [vp, C.c_int, vp]),
    "nchmm_put_transitions": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "nchmm_reserve_slots": (C.c_int, [vp, C.c_int]),
    "nchmm_put_models_scaled": (C.c_int, [vp, C.c_int, C.c_size_t, vp, vp, vp]),
    "nchmm_put_transitions_fast": (C.c_int, [vp, C.c_int, C.c_size_t, vp, vp]),
    "nchmm_viterbi": (C.c_int, [vp, C.c_size_t] + [vp] * 9),
    "nchmm_viterbi_dev": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_size_t] + [vp] * 10),
    "nchmm_viterbi_dev_enqueue": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_size_t] + [vp] * 10),
    "nchmm_viterbi_dev_join": (C.c_int, [vp]),
    "nchmm_viterbi_strand": (C.c_int, [vp, vp, C.c_float, C.c_float, C.c_size_t, vp, vp, vp, vp, vp]),
    "nchmm_viterbi_strand_scaled": (C.c_int, [vp, vp, vp, C.c_float, C.c_float, C.c_size_t, vp, vp, vp, vp, vp]),
    "nchmm_fwbw_windows": (C.c_int, [vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t] + [vp] * 9),
    "nchmm_model_image": (C.c_int, [vp, vp, vp]),
    "nchmm_put_model_images": (C.c_int, [vp, C.c_int, C.c_size_t, vp, vp]),
    "nchmm_viterbi_raw": (C.c_int, [vp, C.c_size_t, vp, vp, vp, C.c_size_t] + [vp] * 8),
    "nchmm_viterbi_begin": (C.c_int, [vp, C.c_size_t] + [vp] * 9),
    "nchmm_viterbi_raw_begin": (C.c_int, [vp, C.c_size_t, vp, vp, vp, C.c_size_t] + [vp] * 8),
    "nchmm_viterbi_end": (C.c_int, [vp]),
    "nchmm_viterbi_in_flight": (C.c_int, [vp]),
    "nchmm_logf": (C.c_int, [vp, C.c_size_t, vp, vp]),
    "nchmm_fwbw": (C.c_int, [vp, C.c_size_t] + [vp] * 13),
    "nchmm_fwbw_dev": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_size_t] + [vp] * 13),
    "nchmm_em_load_events": (C.c_int, [vp, C.c_size_t, vp, vp, vp, vp]),
    "nchmm_em_round": (C.c_int, [vp, C.c_size_t] + [vp] * 7 + [C.c_size_t, vp, C.c_int, vp, vp, vp]),
    "nchmm_counters": (C.c_int, [vp, vp]),
    "nchmm_last_kernel_ms": (C.c_int, [vp, vp]),
    "nchmm_shader_clock_mhz": (C.c_int, [vp, vp]),
    "nchmm_profile_ticks": (C.c_int, [vp, vp, C.c_int]),
    "nchmm_profile_blocks": (C.c_int, [vp, vp]),
    "nchmm_grid_slots": (C.c_int, [vp, vp]),
    "nchmm_set_sweep": (C.c_int, [vp, C.c_int]),
    "nchmm_sweep_stats": (C.c_int, [vp, vp]),
    "nchmm_ahead_stats": (C.c_int, [vp, vp]),
    "nchmm_mem_stats": (C.c_int, [vp, vp]),
    "nchmm_device_count": (C.c_int, [vp]),
    "nchmm_device_mem_info": (C.c_int, [C.c_int, vp, vp]),
    "nchmm_pool_create": (C.c_int, [C.POINTER(vp), C.c_int, vp]),
    "nchmm_pool_destroy": (C.c_int, [vp]),
    "nchmm_pool_size": (C.c_int, [vp]),
    "nchmm_reserve_fb_workspace": (C.c_int, [vp, C.c_size_t]),
    "nchmm_pool_reserve_fb_workspace": (C.c_int, [vp, C.c_size_t]),
    "nchmm_reserve_viterbi_workspace": (C.c_int, [vp, C.c_size_t]),
    "nchmm_pool_reserve_viterbi_workspace": (C.c_int, [vp, C.c_size_t]),
    "nchmm_pool_ctx": (vp, [vp, C.c_int]),
    "nchmm_lpt_partition": (C.c_int, [C.c_size_t, vp, C.c_int, vp]),
    "nchmm_pool_train_reads": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t] + [vp] * 8),
    "nchmm_pool_basecall_reads": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t] + [vp] * 9),
    "nchmm_pool_counters": (C.c_int, [vp, vp, vp]),
    "nchmm_rccl_unique_id": (C.c_int, [vp]),
    "nchmm_counters_allreduce": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp]),
}


class TrainOpts(C.Structure):
    """nchmm_train_opts"""
    _fields_ = [("scaling_num_events", C.c_uint32), ("scaling_max_rounds", C.c_uint32), ("scaling_min_progress", C.c_float),
                ("scaling_select_threshold", C.c_float), ("min_ed_events", C.c_uint32), ("train_scaling", C.c_int32),
                ("train_transitions", C.c_int32), ("train_drift", C.c_int32), ("default_p_stay", C.c_float),
                ("default_p_skip", C.c_float)]


class SegmentOpts(C.Structure):
    """nchmm_segment_opts"""
    _fields_ = [("min_ed_events", C.c_uint32), ("max_ed_events", C.c_uint32), ("abasic_level_top_percent", C.c_double),
                ("abasic_level_top_offset", C.c_double), ("template_only", C.c_uint32), ("trim_margins", C.c_uint32 * 4)]


class ReadSummary(C.Structure):
    """nchmm_read_summary"""
    _fields_ = [("num_ed_events", C.c_uint32), ("abasic_level", C.c_float), ("strand_bounds", C.c_uint32 * 4),
                ("scale_strands_together", C.c_int32), ("time_length", C.c_float * 2)]


class Fast5Read(C.Structure):
    """nchmm_fast5_read"""
    _fields_ = [("have_sampling_rate", C.c_int32), ("have_events", C.c_int32), ("sampling_rate", C.c_double),
                ("ed_group", C.c_char * 32), ("read_name", C.c_char * 64), ("read_id", C.c_char * 256),
                ("n_events", C.c_size_t), ("events", C.c_void_p)]


class NchmmError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().nchmm_strerror(code).decode()
        if code == -7:
            msg += ": " + lib().nchmm_fast5_last_error().decode()
        super().__init__(f"{where}: {msg} (code {code})")


def _share_torch_hip_runtime():
    """One HIP runtime per process.  libnanocall_hip.so needs `libamdhip64.so.7` (found in /opt/rocm/lib); torch's
    libtorch_hip.so needs `libamdhip64.so` (no version suffix) and finds its OWN bundled copy through its $ORIGIN
    rpath.  The dynamic loader tells libraries apart by the name they were asked for and by file identity, so if the
    product is loaded first the process ends up with two runtimes and the second one finds no GPU
    ("RuntimeError: No HIP GPUs are available" from torch.cuda, round-2 GPUTEST).  Torch's copy carries the SONAME
    libamdhip64.so.7, so mapping it first, globally, satisfies both: the product's NEEDED entry matches the loaded
    SONAME, and torch later re-opens the same file.  Without torch installed (the C++ command line, a C host) nothing
    happens and the product uses /opt/rocm's runtime.  NANOCALL_HIP_RUNTIME=system skips this."""
    if os.environ.get("NANOCALL_HIP_RUNTIME", "") == "system":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")     # locates the package without importing it
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


_hip_runtime = None


def lib():
    """Load the shared library (once).  Raises if it has not been built -- no fallback."""
    global _lib, _hip_runtime
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C nanocall_amd/csrc). nanocall_amd has no CPU/PyTorch fallback.")
        _hip_runtime = _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the ABI and this table ever diverge
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(code, where):
    if code != 0:
        raise NchmmError(code, where)
