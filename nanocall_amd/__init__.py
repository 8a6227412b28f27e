"""nanocall_amd -- MI355X-native HMM basecalling core for nanocall (Viterbi + forward-backward EM
over the 4096-state 6-mer pore HMM) behind a C ABI (include/nanocall_hip.h).

Python here is only the test/bench binding over that ABI:
  nanocall_amd.api     numpy-level wrappers of the C entry points (host prep + device calls)
  nanocall_amd.models  the six builtin pore-model tables (data)
  nanocall_amd.synth   the seeded synthetic event generator of SURVEY.md section 8d
"""
from . import _lib  # noqa: F401
from ._lib import NchmmError  # noqa: F401
from .api import (Context, model_load, model_scale, model_pack6, transitions_fast, events_prepare,  # noqa: F401
                  base_seq, write_fasta, st_train_kmers, scaled_model_table, train_pm_finish, train_st_finish,
                  device_count, device_mem_info)
from .models import builtin_model, builtin_names, builtin_strands  # noqa: F401

__version__ = "0.1.0"
