"""Builtin pore-model tables (data): six 4096 x 4 float32 tables
(level_mean, level_stdv, sd_mean, sd_stdv) in k-mer order -- the same numbers the reference carries
as generated initializer lists (src/nanocall/Builtin_Model.cpp:15-17,
src/builtin_models/builtin_model_names.inl:1-13).  Produced by tools/extract_builtin_models.py.
"""
import json
import os

import numpy as np

_D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_meta = None
_tables = None


def _load():
    global _meta, _tables
    if _tables is None:
        with open(os.path.join(_D, "builtin_models.json")) as f:
            _meta = json.load(f)
        _tables = np.fromfile(os.path.join(_D, "builtin_models.f32"), dtype="<f4").reshape(_meta["shape"])
    return _meta, _tables


def builtin_names():
    return list(_load()[0]["names"])


def builtin_strands():
    return list(_load()[0]["strands"])


def builtin_model(name_or_index):
    """Return the 4096 x 4 float32 table of a builtin model (by index or by name prefix such as
    'r73.t' / 'r9.t.007.ont.model')."""
    meta, t = _load()
    if isinstance(name_or_index, int):
        return t[name_or_index].copy()
    hits = [i for i, n in enumerate(meta["names"]) if n == name_or_index or n.startswith(name_or_index + ".")]
    if len(hits) != 1:
        raise KeyError(f"builtin model {name_or_index!r}: {len(hits)} matches in {meta['names']}")
    return t[hits[0]].copy()
