"""bg_gather_records on the device (csrc/comm.hip): (1) the RCCL flavour with a world of one — ncclGetUniqueId,
ncclCommInitRank and ncclAllGather really run on this GPU, equal and ragged (grouped-broadcast) paths; (2) the
host-staged flavour with TWO ranks sharing the one GPU: each aligns its bg_shard_range slice through the device entry
point and gathers the 20-byte score/coordinate records — equal to the unsharded run."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world_of_one_runs_the_collective():
    import torch
    from rust_bio_amd import _lib, comm
    ctx = _lib.Context(0)
    c = comm.Comm.rccl(ctx, 0, 1, comm.unique_id())
    local = torch.arange(5 * 7, dtype=torch.int32, device="cuda:0").reshape(5, 7).contiguous()
    out = torch.zeros_like(local)
    counts = c.gather_ptr(local.data_ptr(), 5, 28, out.data_ptr())
    torch.cuda.synchronize()
    assert counts.tolist() == [5] and torch.equal(out, local)
    # an `all` that is too small is refused BEFORE the collective writes (device and host flavours)
    small = torch.full((2 + 6, 7), -1, dtype=torch.int32, device="cuda:0")
    cnt = np.zeros(1, dtype=np.uint64)
    rc = _lib.lib().bg_gather_records_cap(c.h, local.data_ptr(), 5, 28, small.data_ptr(), 2, cnt.ctypes.data, 0)
    torch.cuda.synchronize()
    assert rc == -9 and bool((small == -1).all())  # BG_ERR_OPS_CAP, not a byte written
    hloc = local.cpu().numpy()
    hsmall = np.full((1 + 6, 7), -1, dtype=np.int32)
    rc = _lib.lib().bg_gather_records_host(c.h, hloc.ctypes.data, 5, 28, hsmall.ctypes.data, 1, cnt.ctypes.data)
    assert rc == -9 and (hsmall == -1).all()
    hall, cnt2 = c.gather_host(hloc, 5)  # ... and with room: staged through device scratch sized from the counts
    assert cnt2.tolist() == [5] and (hall == hloc).all()
    c.free()


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    torch.cuda.init()
    from rust_bio_amd import _lib, comm, synth
    from rust_bio_amd.pairwise import Aligner, Scoring
    rank, world, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    c = comm.Comm.host(ctx, rank, world, name)
    st = torch.cuda.current_stream().cuda_stream
    n_pairs, L = 2001, 150
    x, xo, y, yo = synth.sw_pairs(n_pairs, L, seed=2)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)
    def align(lo, hi):
        k = hi - lo
        dx, dy = torch.from_numpy(x[lo * L:hi * L].copy()).to(dev), torch.from_numpy(y[lo * L:hi * L].copy()).to(dev)
        o = torch.arange(k + 1, dtype=torch.int64, device=dev) * L
        out = torch.empty(k * 64, dtype=torch.uint8, device=dev); ops = torch.empty(k * (2 * L + 4), dtype=torch.uint8, device=dev)
        al.align_dev(3, k, dx.data_ptr(), o.data_ptr(), dy.data_ptr(), o.data_ptr(), L, L, out.data_ptr(), ops.data_ptr(), 2 * L + 4, st)
        return out.view(torch.int32).view(k, 16)[:, :5].contiguous()
    lo, hi = comm.shard_range(n_pairs, rank, world)
    mine = align(lo, hi)
    full = torch.zeros((n_pairs, 5), dtype=torch.int32, device=dev)
    counts = c.gather_ptr(mine.data_ptr(), hi - lo, 20, full.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(counts.sum()) == n_pairs and counts[rank] == hi - lo
    assert torch.equal(full, align(0, n_pairs)), "gathered records differ from the unsharded run"
    c.free()
    print("rank", rank, "ok")
""") % ROOT


def test_two_ranks_on_one_gpu_gather_through_host_memory(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    name = "g%d" % os.getpid()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", name], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0 and b"ok" in out, (r, err.decode()[-2000:])
