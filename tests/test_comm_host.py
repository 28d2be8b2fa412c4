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


def _cap_worker(rank, world, name, q):
    """`all` too small for the ranks' records -> BG_ERR_OPS_CAP on EVERY rank, nothing written, and the
    communicator stays usable (the next gather, with room, succeeds)"""
    try:
        from rust_bio_amd import _lib
        c = comm.Comm.host(None, rank, world, name)
        n_local = [4, 3][rank]
        local = np.arange(n_local * 2, dtype=np.int64).reshape(n_local, 2) + 100 * rank
        guard = np.full((3 + 8, 2), -1, dtype=np.int64)  # room for 3 records, 8 guard records behind
        counts = np.zeros(world, dtype=np.uint64)
        # rank 1 says 3 (too small), rank 0 says 100: the smallest capacity decides for everybody
        rc = _lib.lib().bg_gather_records_host(c.h, local.ctypes.data, n_local, 16, guard.ctypes.data, [100, 3][rank], counts.ctypes.data)
        assert rc == -9, rc  # BG_ERR_OPS_CAP
        assert (guard == -1).all()  # not one byte of `all` touched
        allr, counts = c.gather_host(local, 7)
        assert counts.tolist() == [4, 3] and (allr[:4] == np.arange(8).reshape(4, 2)).all() and (allr[4:] == np.arange(6).reshape(3, 2) + 100).all()
        c.free()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_host_staged_gather_refuses_a_too_small_all_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "c%d" % os.getpid()
    procs = [ctx.Process(target=_cap_worker, args=(r, 2, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "ok"), (1, "ok")], res


def _stale_worker(rank, world, name, delay, q):
    try:
        import time
        time.sleep(delay)
        c = comm.Comm.host(None, rank, world, name)
        local = np.full((rank + 1, 1), rank, dtype=np.int32)
        allr, counts = c.gather_host(local, 3)
        assert counts.tolist() == [1, 2] and allr[:, 0].tolist() == [0, 1, 1]
        c.free()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_init_survives_a_stale_control_segment_of_the_same_name():
    """a control segment left by a crashed run — all ranks 'arrived', barrier state mid-phase — must not
    be mistaken for this run's: rank 1 starts FIRST and maps it; rank 0 arrives a second later and replaces it"""
    name = "s%d" % os.getpid()
    path = "/dev/shm/bg_%s_ctl" % name
    with open(path, "wb") as f:  # plausible garbage: arrived = 7, count = 1, sense = 1, then zeros
        f.write(np.array([7, 1, 1, 0], dtype=np.uint32).tobytes() + bytes(8192))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stale_worker, args=(r, 2, name, [1.0, 0.0][r], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "ok"), (1, "ok")], res
    assert not os.path.exists(path)
