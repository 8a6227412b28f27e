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


def _bench(args, env_extra=None):
    env = dict(os.environ, NCHMM_BENCH_SHARE_GPU0="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                   # rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("ranks", [2, 4])
def test_ranks_share_gpu0_and_report_whole_job_throughput(ranks):
    d = _bench(["--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--reads", "300", "--events", "800"])
    assert d["n_gpus"] == ranks and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["reads_per_gpu"] == 300 and d["config"]["reads_total"] == 300 * ranks
    assert f"read-sharded x{ranks}" in d["config"]["parallelism"]
    assert f"communicator of {ranks} ranks" in d["config"]["collective"]
    # counters are summed over all ranks: (warmup + steps) launches of 300 reads x 800 events each
    assert d["counters"]["reads"] == ranks * 300 * 3 and d["counters"]["events"] == ranks * 300 * 800 * 3
    # whole-job value = all ranks' events / max-over-ranks time
    assert abs(d["value"] - ranks * 300 * 800 * 2 / (d["ms_per_step"] * 2 * 1e-3) / 1e6) <= 0.01 * d["value"]
    assert "cpu_baseline" not in d and "fwbw" not in d and "end_to_end" not in d   # N = 1 only
    assert 400 <= d["device"]["shader_clock_mhz_under_load"] <= 2600


@pytest.mark.parametrize("ranks", [1, 2, 4])
def test_strong_scaling_splits_one_global_read_set(ranks):
    """--scaling strong: the SAME global read set at every N (BASELINE config 4 as written is 100 000 reads; 1 001 here,
    deliberately not divisible), split by lpt_partition -- the counters prove every read was decoded exactly once per step."""
    d = _bench(["--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--reads", "1001", "--events", "600", "--scaling", "strong",
                "--no-cpu-baseline", "--no-fwbw", "--no-end-to-end"])
    assert d["n_gpus"] == ranks and d["scaling"] == "strong"
    assert d["config"]["reads_total"] == 1001 and "1001 reads x 600 events in total" in d["config"]["workload"]
    assert d["counters"]["reads"] == 1001 * 3 and d["counters"]["events"] == 1001 * 600 * 3
    assert abs(d["value"] - 1001 * 600 * 2 / (d["ms_per_step"] * 2 * 1e-3) / 1e6) <= 0.01 * d["value"]


def test_default_line_carries_every_object_of_the_contract():
    """N = 1 defaults (what the driver runs), on a reduced read count so that the test stays short: roofline, cpu_baseline
    (threads = physical cores, the sample string says what ran), end_to_end (host pointers, PCIe inclusive), fwbw."""
    d = _bench(["--steps", "2", "--warmup", "1", "--reads", "256", "--cpu-threads", "8"], env_extra={"NCHMM_BENCH_SHARE_GPU0": "0"})
    assert d["n_gpus"] == 1 and d["config"]["collective"].startswith("none")
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["kernel_ms"] > 0
    c = d["cpu_baseline"]
    assert c["cores"] == 8 and "8 read-parallel threads" in c["sample"] and c["parity_checked_reads"] == 256 and c["kind"] == "port"
    e = d["end_to_end"]
    # streamed batches keep the kernels back to back, which the device-resident loop (one hipEvent read-back per step) does not
    # quite: on a quarter-size batch the PCIe-inclusive figure may come out a little above the headline, never far
    assert e["identical_to_device_resident_run"] and 0 < e["value"] < d["value"] * 1.25
    assert 0 < e["one_call"]["value"] < e["value"] * 1.05 and e["batches"] >= 5
    assert d["device"]["peak_mem_bytes"] > 0 and d["device"]["library_peak_bytes"] > 0 and len(d["output_sha256_16"]) == 16
    assert d["fwbw"]["roofline"]["frac"] > 0
