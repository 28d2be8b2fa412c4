"""world_size-2 `gloo` test of the multi-GPU path (rust-bio_amd/shard.py): contiguous
partition, the single all-gather of fixed-size result records, ragged shards, max-over-ranks."""
import os
import socket
import subprocess
import sys
import textwrap

import torch

from rust_bio_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_everything_once():
    for n in (0, 1, 7, 100, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [shard.partition(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_partition_balanced_by_cost():
    costs = [1] * 50 + [100] * 10
    b = shard.partition_balanced(costs, 4)
    assert b[0] == 0 and b[-1] == 60 and b == sorted(b)
    sums = [sum(costs[b[i]:b[i + 1]]) for i in range(4)]
    assert max(sums) <= 2 * (sum(costs) / 4) + 100


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from rust_bio_amd import shard
    rank, local_rank, world = shard.init_process_group("gloo")
    assert world == 2
    n = 1001
    lo, hi = shard.partition(n, rank, world)
    # fixed-size records of this rank's shard: (query id, id*3, id*7)
    ids = torch.arange(lo, hi, dtype=torch.int64)
    local = torch.stack((ids, ids * 3, ids * 7), dim=1)
    full = shard.gather_records(local)           # ragged shards (500 / 501): padded + trimmed
    assert full.shape == (n, 3), full.shape
    assert (full[:, 0] == torch.arange(n)).all() and (full[:, 2] == torch.arange(n) * 7).all()
    even = shard.gather_records(local[:500], counts=[500, 500])   # equal shards: one collective
    assert even.shape == (1000, 3)
    assert (even[:500, 0] == torch.arange(0, 500)).all() and (even[500:, 0] == torch.arange(500, 1000)).all()
    t = shard.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert t == 2.0
    shard.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_gather_records_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()
        assert f"rank {r} ok" in out.decode()
