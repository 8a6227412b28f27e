"""bench.py's N > 1 path on a 1-GPU box: `--gpus 2` launches its own two ranks (torch.distributed.run), both on GPU 0 with
gloo for the counter all-reduce (NCHMM_BENCH_SHARE_GPU0=1, a test hook: RCCL cannot put two ranks on one device).  What
runs is everything else the 8-GPU line depends on: the self-launcher, one LPT shard per rank, per-rank generation of its
own reads, the barrier-bracketed timing with the max over ranks, the summed counters, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_gpu0_and_report_whole_job_throughput():
    env = dict(os.environ, NCHMM_BENCH_SHARE_GPU0="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "600",
                        "--events", "800"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["reads_per_gpu"] == 600 and "read-sharded x2" in d["config"]["parallelism"]
    # counters are summed over both ranks: (warmup + steps) launches of 600 reads x 800 events each
    assert d["counters"]["reads"] == 2 * 600 * 3 and d["counters"]["events"] == 2 * 600 * 800 * 3
    # whole-job value = all ranks' events / max-over-ranks time
    assert abs(d["value"] - 2 * 600 * 800 * 2 / (d["ms_per_step"] * 2 * 1e-3) / 1e6) <= 0.01 * d["value"]
    assert "cpu_baseline" not in d and "fwbw" not in d   # N = 1 only
    assert 400 <= d["device"]["shader_clock_mhz_under_load"] <= 2600
