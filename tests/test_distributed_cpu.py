"""CPU, 2 processes over gloo: replica sharding is deterministic (a replica's trajectory does not depend on how many
ranks there are) and the counter reduction adds up.  The compute stands in with the oracle (no GPU here); the sharding /
Philox world_base / reduction logic is the product's (reinlife_amd/distributed.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, TICKS, SEED = 6, 12, 99


def _run(first, count):
    from oracle import oracle as orc
    ow = orc.OracleWorlds(n_worlds=count, seed=SEED, world_base=first, n_brains=2)
    ow.reset_synthetic(100)
    acted = 0
    for t in range(TICKS):
        rng = np.random.RandomState(1000 * t)  # same action table for every replica layout
        table = rng.randint(0, 8, size=(TOTAL, ow.cap)).astype(np.int8)
        ow.step(table[first:first + count])
        ow.update()
        acted += int(ow.n_acted.sum())
    return ow, acted


def _worker(rank, world_size, port, q):
    sys.path.insert(0, ROOT)
    from reinlife_amd import distributed as rd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    first, count = rd.shard(TOTAL, rank, world_size)
    ow, acted = _run(first, count)
    c, tmax, table = rd.reduce_counters(torch.tensor([float(acted), float(count)], dtype=torch.float64), 0.5 + rank, dist)
    q.put((rank, first, count, ow.s["cell_type"].copy(), ow.s["a_health"].copy(), ow.s["n_agents"].copy(), c.tolist(), tmax, table.numpy(), acted))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sharding_is_layout_independent_and_counters_reduce():
    sys.path.insert(0, ROOT)
    from reinlife_amd import distributed as rd
    assert [rd.shard(7, r, 3) for r in range(3)] == [(0, 3), (3, 2), (5, 2)]
    ref, ref_acted = _run(0, TOTAL)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, first, count, cells, health, n, c, tmax, table, acted in res:
        assert np.array_equal(cells, ref.s["cell_type"][first:first + count])
        assert np.array_equal(n, ref.s["n_agents"][first:first + count])
        assert np.array_equal(health, ref.s["a_health"][first:first + count])
        assert c == [float(ref_acted), float(TOTAL)] and tmax == 1.5
        # the ONE collective also hands every rank the per-rank rows (bench.py's straggler report): [acted, count, elapsed] of rank r
        assert table.shape == (2, 3) and table[rank].tolist() == [float(acted), float(count), 0.5 + rank]
    assert sum(r[-1] for r in res) == ref_acted
    # without a process group the same call degenerates to the caller's own row
    c, tmax, table = rd.reduce_counters(torch.tensor([3.0, 4.0], dtype=torch.float64), 0.25, None)
    assert c.tolist() == [3.0, 4.0] and tmax == 0.25 and table.shape == (1, 3)
