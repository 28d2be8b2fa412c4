"""The sharding helpers and the host-staged gather of include/biogpu.h (csrc/comm.hip) without a GPU: three processes meet
in POSIX shared memory, bring ragged numbers of records and must all receive every record in rank order, twice (the
barrier and the data segments are reused); bg_shard_range / bg_shard_balanced against shard.py's definitions."""
import multiprocessing as mp
import os

import numpy as np

from rust_bio_amd import comm, shard


def test_shard_range_and_balanced_match_the_python_definitions():
    for n in (0, 1, 7, 1000, 10_000_019):
        for world in (1, 2, 3, 8):
            got = [comm.shard_range(n, r, world) for r in range(world)]
            assert got == [shard.partition(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    rng = np.random.default_rng(3)
    for trial in range(20):
        costs = rng.integers(1, 1000, size=int(rng.integers(1, 400))).astype(np.uint64) ** 2
        world = int(rng.integers(1, 9))
        b = comm.shard_balanced(costs, world)
        assert b[0] == 0 and b[-1] == len(costs) and all(x <= y for x, y in zip(b, b[1:]))
        total = float(costs.sum())
        for r in range(1, world):  # boundary r: the running cost up to it does not pass r / world of the total, one more unit does
            run = float(costs[:b[r]].sum())
            assert run <= total * r / world + 1e-6
            assert b[r] == len(costs) or run + float(costs[b[r]]) > total * r / world - 1e-6


def _worker(rank, world, name, q):
    try:
        c = comm.Comm.host(None, rank, world, name)
        for round_ in range(2):
            n_local = [5, 0, 12][rank] + round_
            local = (np.arange(n_local * 3, dtype=np.int32).reshape(n_local, 3) + 1000 * rank + 7 * round_)
            total = sum([5, 0, 12]) + world * round_
            allr, counts = c.gather_host(local, total)
            want = np.concatenate([(np.arange((k + round_) * 3, dtype=np.int32).reshape(k + round_, 3) + 1000 * r + 7 * round_)
                                   for r, k in enumerate([5, 0, 12])])
            assert counts.tolist() == [k + round_ for k in [5, 0, 12]]
            assert (allr == want).all()
        c.free()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_host_staged_gather_three_ranks_ragged():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "t%d" % os.getpid()
    procs = [ctx.Process(target=_worker, args=(r, 3, name, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "ok"), (1, "ok"), (2, "ok")], res
