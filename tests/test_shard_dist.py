"""Multi-GPU path on CPU: read sharding + the one counter all-reduce, world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nanocall_amd import shard


def test_lpt_partition_balances_and_covers():
    rng = np.random.default_rng(0)
    lens = rng.integers(10, 50000, size=1000)
    for ws in (1, 2, 4, 8):
        parts = shard.lpt_partition(lens, ws)
        allidx = np.concatenate(parts)
        assert sorted(allidx.tolist()) == list(range(1000))
        loads = np.array([lens[p].sum() for p in parts])
        assert loads.max() - loads.min() <= lens.max()
    # equal lengths (BASELINE config 4): contiguous slices
    parts = shard.lpt_partition(np.full(100000, 5000), 8)
    assert [len(p) for p in parts] == [12500] * 8 and all((np.diff(p) == 1).all() for p in parts)
    assert shard.lpt_partition([], 2)[0].size == 0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = np.arange(1, 41) * 100
    mine = shard.lpt_partition(lens, world)[rank]
    local = np.array([len(mine), int(lens[mine].sum()), 0, 1, 0, 0, 0, 0, 1], np.uint64)      # last slot: "a rank was here" (bench.py)
    tot = shard.gather_counters(local)
    mx = shard.max_over_ranks(1.0 + rank)
    per = shard.gather_per_rank([10.0 + rank, 2000.0 + 7 * rank, 15.0, float(len(mine))])    # bench.py: step time, clock, kernel ms, reads
    q.put((rank, tot.tolist(), mx, mine.tolist(), per.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_counter_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    lens = np.arange(1, 41) * 100
    seen = []
    for rank, tot, mx, mine, per in res:
        assert tot[0] == 40 and tot[1] == int(lens.sum()) and tot[3] == world and tot[8] == world
        assert mx == 2.0
        # every rank sees every rank's row, in rank order
        assert per == [[10.0, 2000.0, 15.0, float(len(shard.lpt_partition(lens, world)[0]))], [11.0, 2007.0, 15.0, float(len(shard.lpt_partition(lens, world)[1]))]]
        seen += mine
    assert sorted(seen) == list(range(40))


def test_gather_per_rank_without_a_process_group_is_one_row():
    assert shard.gather_per_rank([1.5, 2.5]).tolist() == [[1.5, 2.5]]


def test_c_abi_lpt_partition_equals_the_python_one():
    """nchmm_lpt_partition (what the C++ host / CLI shards with) against nanocall_amd.shard.lpt_partition."""
    from nanocall_amd import api
    rng = np.random.default_rng(5)
    cases = [rng.integers(10, 50000, size=1000), np.full(100000, 5000), np.full(7, 3), np.array([], np.int64),
             np.array([5, 5, 9, 9, 1]), rng.integers(1, 4, size=64)]
    for lens in cases:
        for ws in (1, 2, 3, 8):
            of = api.lpt_partition(lens, ws)
            parts = shard.lpt_partition(lens, ws)
            for k in range(ws):
                assert np.array_equal(np.nonzero(of == k)[0], parts[k]), (len(lens), ws, k)
