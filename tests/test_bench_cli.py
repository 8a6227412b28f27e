"""bench.py's launcher contract, on CPU: --gpus N must never report an N-GPU number from fewer devices."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_gpus_n_without_n_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has the GPUs")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert "only" in p.stderr and "GPU" in p.stderr
    assert '"n_gpus"' not in p.stdout          # no JSON line was printed


def test_world_size_must_match_gpus():
    p = _run(["--gpus", "4", "--steps", "1"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr and '"n_gpus"' not in p.stdout


def test_shard_generation_matches_the_plain_generator():
    """bench.generate_shard (chunked, threaded) yields exactly synth.generate's reads for the rank's LPT shard."""
    sys.path.insert(0, ROOT)
    import bench
    import nanocall_amd as na
    from nanocall_amd import shard, synth
    table = na.builtin_model("r73.t")
    mine = shard.lpt_partition(np.full(12, 40), 3)[1]          # reads 4..7
    off, mean, stdv, start = bench.generate_shard(table, mine, 40, threads=2)
    ref = synth.generate(table, 4, 40, first_read=4)
    assert off.tolist() == [0, 40, 80, 120, 160]
    assert np.array_equal(mean, ref["mean"].reshape(-1)) and np.array_equal(stdv, ref["stdv"].reshape(-1))
    assert np.array_equal(start, ref["start"].reshape(-1))
    # a ragged shard (non-consecutive ids) is generated read by read
    off, mean, _, _ = bench.generate_shard(table, [1, 5, 6], 40, threads=2)
    assert np.array_equal(mean[:40], synth.generate(table, 1, 40, first_read=1)["mean"][0])
    assert np.array_equal(mean[40:], synth.generate(table, 2, 40, first_read=5)["mean"].reshape(-1))
