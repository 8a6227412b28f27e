#!/usr/bin/env python3
"""Extract the six builtin pore-model tables from the reference into a binary data file.

Reads  (THIS container only; the reference never travels to the GPU box):
  $REFERENCE_DIR/src/builtin_models/builtin_model_{names,strands,init_lists}.inl
Writes:
  nanocall_amd/data/builtin_models.f32   6 x 4096 x 4 little-endian float32
                                         (level_mean, level_stdv, sd_mean, sd_stdv) in k-mer order
  nanocall_amd/data/builtin_models.json  names, strands, shape, sha256 of the .f32

The tables are DATA (ONT pore-model parameters), the same role weights play for a network;
the reference carries them as generated initializer lists (src/nanocall/Builtin_Model.cpp:15-17).
Decimal text -> double -> float32 is the same two-step rounding the C++ compiler applies to a
`std::vector<float>{62.784241, ...}` initializer, so the bits are identical; tests/test_ref_pins.py
re-checks that against oracle/_ref (the reference's own Builtin_Model.cpp compiled as-is).
"""
import hashlib, json, os, re, sys
import numpy as np

ref = os.environ.get("REFERENCE_DIR", "/root/reference")
d = os.path.join(ref, "src", "builtin_models")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "nanocall_amd", "data")

names = re.findall(r'"([^"]+)"', open(os.path.join(d, "builtin_model_names.inl")).read())
strands = [int(x) for x in re.findall(r"\d+", open(os.path.join(d, "builtin_model_strands.inl")).read())]
txt = open(os.path.join(d, "builtin_model_init_lists.inl")).read()
vals = re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", txt)
arr = np.array([float(v) for v in vals], dtype=np.float64).astype(np.float32)
assert len(names) == len(strands) == 6, (names, strands)
assert arr.size == 6 * 4096 * 4, arr.size
arr = arr.reshape(6, 4096, 4)
raw = arr.astype("<f4").tobytes()
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, "builtin_models.f32"), "wb") as f:
    f.write(raw)
meta = {"names": names, "strands": strands, "shape": [6, 4096, 4],
        "columns": ["level_mean", "level_stdv", "sd_mean", "sd_stdv"],
        "sha256": hashlib.sha256(raw).hexdigest()}
with open(os.path.join(out_dir, "builtin_models.json"), "w") as f:
    json.dump(meta, f, indent=1)
print(meta)
