"""The N > 1 path through the real kernels: two ranks share the one available GPU, `gloo` carries the single
all-gather of fixed-size result records (CPU side), and the gathered records of the sharded run must equal the
unsharded run — for the FM index (BASELINE configs[2], "1->8 GPU shard") and for the aligner.  RCCL itself only
runs in bench.py on a multi-GPU node; everything else of that path (partition, replicas of the index, record
layout, gather, trimming of ragged shards) is what runs here."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from rust_bio_amd import _lib, shard, synth
    from rust_bio_amd.bwt import Occ, bwt, less
    from rust_bio_amd.fmindex import FMIndex
    from rust_bio_amd.pairwise import Aligner, Scoring
    from rust_bio_amd.suffix_array import suffix_array
    rank, local_rank, world = shard.init_process_group("gloo")
    assert world == 2
    dev = torch.device("cuda", 0)            # both ranks on the one GPU
    torch.cuda.set_device(dev)
    ctx = _lib.Context(0)
    st = torch.cuda.current_stream().cuda_stream

    # ---- FM index: replica per rank, queries sharded, one gather of 24-byte records
    g = synth.genome(200_000, 3)
    sa = suffix_array(g); b = bwt(g, sa); ls = less(b, b"ACGTNacgtn")
    fm = FMIndex(b, ls, Occ(b, 128, b"ACGTNacgtn"), ctx=ctx)
    n_q, P = 50_001, 60                      # odd: ragged shards
    pat, off = synth.fm_patterns(g, n_q, P, seed=4)
    d_pat, d_off = torch.from_numpy(pat).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev)

    def search(lo, hi):
        k = hi - lo
        tag = torch.empty(k, dtype=torch.uint8, device=dev); ml = torch.empty(k, dtype=torch.int32, device=dev)
        l = torch.empty(k, dtype=torch.int64, device=dev); u = torch.empty(k, dtype=torch.int64, device=dev)
        so = (d_off[lo:hi + 1] - d_off[lo]).contiguous()
        sp = d_pat[int(off[lo]):int(off[hi])].contiguous()
        fm.backward_search_dev(k, sp.data_ptr(), so.data_ptr(), tag.data_ptr(), l.data_ptr(), u.data_ptr(), ml.data_ptr(), st)
        torch.cuda.synchronize()
        return torch.stack((l, u, ml.to(torch.int64) | (tag.to(torch.int64) << 32)), dim=1).cpu()

    lo, hi = shard.partition(n_q, rank, world)
    full = shard.gather_records(search(lo, hi))
    assert full.shape == (n_q, 3)
    if rank == 0:
        assert (full == search(0, n_q)).all(), "sharded FM records differ from the unsharded run"

    # ---- aligner: pairs sharded, 20-byte score/coordinate records gathered
    n_pairs, L = 3001, 150
    x, xo, y, yo = synth.sw_pairs(n_pairs, L, seed=2)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)
    def align(lo, hi):
        k = hi - lo
        dx, dy = torch.from_numpy(x[lo * L:hi * L].copy()).to(dev), torch.from_numpy(y[lo * L:hi * L].copy()).to(dev)
        o = torch.arange(k + 1, dtype=torch.int64, device=dev) * L
        out = torch.empty(k * 64, dtype=torch.uint8, device=dev); ops = torch.empty(k * (2 * L + 4), dtype=torch.uint8, device=dev)
        al.align_dev(3, k, dx.data_ptr(), o.data_ptr(), dy.data_ptr(), o.data_ptr(), L, L, out.data_ptr(), ops.data_ptr(), 2 * L + 4, st)
        torch.cuda.synchronize()
        return out.view(torch.int32).view(k, 16)[:, :5].contiguous().cpu()
    lo, hi = shard.partition(n_pairs, rank, world)
    full = shard.gather_records(align(lo, hi))
    assert full.shape == (n_pairs, 5)
    if rank == 0:
        assert (full == align(0, n_pairs)).all(), "sharded alignment records differ from the unsharded run"
    # ---- ragged pairs, shards balanced by the sum of DP cells (SURVEY.md section 8(e)): mixed read lengths, the boundary
    # of partition_balanced, ragged gather (different record counts per rank), against the unsharded run
    rng = np.random.default_rng(9)
    lens_x = rng.choice([40, 75, 150, 300, 600], size=1501)
    lens_y = (lens_x * rng.uniform(0.8, 1.3, size=1501)).astype(np.int64) + 1
    rxo = np.concatenate([[0], np.cumsum(lens_x)]).astype(np.int64)
    ryo = np.concatenate([[0], np.cumsum(lens_y)]).astype(np.int64)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rx, ry = acgt[rng.integers(0, 4, size=int(rxo[-1]))], acgt[rng.integers(0, 4, size=int(ryo[-1]))]
    bounds = shard.partition_balanced(lens_x * lens_y, world)
    cells = [int((lens_x[bounds[r]:bounds[r + 1]] * lens_y[bounds[r]:bounds[r + 1]]).sum()) for r in range(world)]
    assert bounds[0] == 0 and bounds[-1] == 1501 and bounds[1] != 1501 // 2            # not the equal-count split
    assert abs(cells[0] - cells[1]) <= int((lens_x * lens_y).max())                     # balanced to within one pair
    def align_ragged(lo, hi):
        k = hi - lo
        dx = torch.from_numpy(rx[rxo[lo]:rxo[hi]].copy()).to(dev); dy = torch.from_numpy(ry[ryo[lo]:ryo[hi]].copy()).to(dev)
        ox = torch.from_numpy(rxo[lo:hi + 1] - rxo[lo]).to(dev); oy = torch.from_numpy(ryo[lo:hi + 1] - ryo[lo]).to(dev)
        mx, my = int(lens_x.max()), int(lens_y.max())
        out = torch.empty(k * 64, dtype=torch.uint8, device=dev); ops = torch.empty(k * (mx + my + 4), dtype=torch.uint8, device=dev)
        al.align_dev(3, k, dx.data_ptr(), ox.data_ptr(), dy.data_ptr(), oy.data_ptr(), mx, my, out.data_ptr(), ops.data_ptr(), mx + my + 4, st)
        torch.cuda.synchronize()
        return out.view(torch.int32).view(k, 16)[:, :5].contiguous().cpu()
    full = shard.gather_records(align_ragged(bounds[rank], bounds[rank + 1]))
    assert full.shape == (1501, 5)
    if rank == 0:
        assert (full == align_ragged(0, 1501)).all(), "cost-balanced ragged shards differ from the unsharded run"
    t = shard.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert t == 2.0
    shard.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_two_ranks_on_one_gpu_sharded_equals_unsharded(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()
        assert f"rank {r} ok" in out.decode()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (both on GPU 0 here through
    BENCH_SINGLE_GPU: the gloo/RCCL choice follows the device count) and reports n_gpus: 2."""
    import json
    env = dict(os.environ, BENCH_SINGLE_GPU="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--pairs", "20000",
           "--genome", "300000", "--fm-big-genome", "0", "--queries", "40000", "--banded-pairs", "97", "--pipeline-reads", "3000",
           "--pipeline-reads-total", "5001", "--skip-ingest", "--skip-k1"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["fm"]["strong"]["queries_total"] == 40000
    assert d["fm"]["strong"]["sharded_equals_unsharded"] is True
    assert d["fm"]["strong"]["gathered_records"] == 40000
    # configs[3] and configs[4] as strong legs: the same pairs / reads in total, split over the two ranks (ragged shards)
    bs, ss = d["banded"]["strong"], d["seed_extend"]["strong"]
    assert bs["pairs_total"] == 97 and bs["pairs_per_gpu"] == 48 and bs["sharded_equals_unsharded"] is True
    assert ss["reads_total"] == 5001 and ss["gathered_records"] == 5001 and ss["sharded_equals_unsharded"] is True
    # the collective of every strong leg is the PRODUCT's (bg_gather_records behind the C ABI; host-staged here: two ranks
    # share the one GPU), cross-checked against torch.distributed's gather of the same records — equal and ragged shards
    for leg in (d["fm"]["strong"], bs, ss):
        assert leg["collective"] == "bg_gather_records (host-staged)", leg["collective"]
        assert leg["capi_gather_equals_torch_gather"] is True
