#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: GCUPS (Smith-Waterman) + FM-index queries/s.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a launcher: re-executes itself under `python -m torch.distributed.run --nnodes=1
        --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU (RCCL over xGMI).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        (what the driver does): RANK / LOCAL_RANK / WORLD_SIZE come from the environment.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  * headline (`value`): BASELINE configs[1] — 1 M x 150 bp synthetic read pairs per GPU through `Aligner::local`
    (affine gaps, Scoring::from_scores(-5,-1,1,-1)): fill + traceback, score + coordinates + full operation list;
  * `fm`: BASELINE configs[2] — FMIndex over a 100 Mbp synthetic genome, 10 M x 100 bp backward_search per GPU
    (weak), and `fm.strong`: the same 10 M queries in total split over the ranks (strong), both with the single
    all-gather of the result records inside the timed step;
  * `value_int32` / `int32`: the headline workload, same pairs, through the int32 kernel K1 (the reference's width);
  * `fm_big`: the same searches on the 3 Gbp index of configs[4] (1 GB of rank blocks: four times the Infinity Cache);
  * `seed_extend` (configs[4]): 1.25 M reads per GPU vs that 3 Gbp genome through bg_seed_extend_batch_dev (weak), and
    `seed_extend.strong`: 10 M reads in total split over the ranks, record all-gather inside the step;
  * `banded` (configs[3]): 100 000 x 10 kb pairs per GPU, banded::Aligner::semiglobal (weak), `banded.strong`: the 100 000
    pairs in total split over the ranks; oracle parity on >= 1 % of the pairs (records + every operation);
  * `k1_int32`: the general int32 kernel at 1 M pairs (BLOSUM62 protein pairs; DNA with scores beyond 12 bits);
  * `ingest` (FASTQ text -> records).
Units (pairs / queries / reads) shard across ranks; every rank builds its own replica of the index on its GPU
(suffix array, BWT and SA samples on the device).  Rank 0 prints ONE JSON line.  The CPU legs (oracle parity over the whole workload +
`cpu_baseline`, median of 3) run on rank 0 of the single-GPU run only.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_LANE_OPS = 78.6e12  # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (full-rate VALU ops)
N_ALPHABET = b"ACGTNacgtn"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1_000_000, help="read pairs per GPU (configs[1]: 1M)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=100_000_000, help="FM leg: genome length (configs[2]: 100 Mbp)")
    ap.add_argument("--queries", type=int, default=10_000_000, help="FM leg: patterns per GPU (configs[2]: 10M)")
    ap.add_argument("--pattern-len", type=int, default=100)
    ap.add_argument("--skip-fm", action="store_true")
    ap.add_argument("--fm-big-genome", type=int, default=3_000_000_000,
                    help="genome of the second FM leg and of the seed-and-extend leg (BASELINE configs[4]: 3 Gbp; 1 GB of rank "
                         "blocks, four times the 256 MiB Infinity Cache); 0: skip, seed-and-extend then runs on --genome")
    ap.add_argument("--host-sa", action="store_true", help="build suffix arrays with the host SA-IS instead of the device builder")
    ap.add_argument("--skip-k1", action="store_true")
    ap.add_argument("--skip-packed", action="store_true", help="no 2-bit packed A/B legs (packed2)")
    ap.add_argument("--k1-pairs", type=int, default=1_000_000, help="int32-kernel legs: pairs per GPU (configs[1]'s 1M)")
    ap.add_argument("--skip-semiglobal", action="store_true", help="no Aligner::semiglobal leg on the headline pairs")
    ap.add_argument("--skip-banded", action="store_true")
    ap.add_argument("--banded-pairs", type=int, default=100_000,
                    help="banded leg: 10 kb pairs per GPU, and in total for banded.strong (configs[3]: 100k, split over the GPUs)")
    ap.add_argument("--banded-parity-pairs", type=int, default=5120,
                    help="banded leg: pairs compared with the oracle (5 % of configs[3]; ~13 s of the 16-thread oracle)")
    ap.add_argument("--banded-chunk", type=int, default=0, help="banded leg: pairs per sub-batch (0 = the library's default)")
    ap.add_argument("--skip-pipeline", action="store_true")
    ap.add_argument("--pipeline-reads", type=int, default=1_250_000,
                    help="seed-and-extend leg: reads per GPU (configs[4]: 10 M reads over 8 GPUs = 1.25 M each)")
    ap.add_argument("--pipeline-reads-total", type=int, default=10_000_000,
                    help="seed_extend.strong: reads in total, split over the GPUs (configs[4]: 10 M); 0: skip")
    ap.add_argument("--skip-ingest", action="store_true")
    ap.add_argument("--ingest-reads", type=int, default=1_000_000, help="FASTQ ingest leg: four-line records per GPU")
    ap.add_argument("--skip-fmd", action="store_true", help="no FMD-index SMEM leg")
    ap.add_argument("--fmd-genome", type=int, default=100_000_000, help="SMEM leg: genome length T (the index is over T$R$: 2 T + 2 symbols)")
    ap.add_argument("--fmd-reads", type=int, default=500_000, help="SMEM leg: 150 bp reads per GPU")
    ap.add_argument("--quick", action="store_true",
                    help="A/B work: no CPU legs (parity, cpu_baseline), no 3 Gbp index, no packed / strong / host-API side legs, one "
                         "sub-batch of the banded workload — every kernel still runs on its full-size shape; NOT the measured configuration")
    ap.add_argument("--skip-cpu", action="store_true", help="no oracle parity / cpu_baseline legs")
    ap.add_argument("--parity-frac", type=float, default=1.0, help="fraction of every leg's output compared with the oracle")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start N ranks of this very command."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


ARGS = parse_args()
if ARGS.quick:
    ARGS.skip_cpu = ARGS.skip_packed = True
    ARGS.fm_big_genome = 0
    ARGS.pipeline_reads_total = 0
    ARGS.banded_pairs = min(ARGS.banded_pairs, 16384)
    ARGS.banded_parity_pairs = 0
relaunch_under_torchrun(ARGS)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from rust_bio_amd import _lib, shard, synth, synth_gpu  # noqa: E402
from rust_bio_amd.bwt import Occ, bwt, less  # noqa: E402
from rust_bio_amd.fmindex import FMIndex  # noqa: E402
from rust_bio_amd.pairwise import Aligner, Scoring  # noqa: E402
from rust_bio_amd.suffix_array import suffix_array  # noqa: E402


class Gatherer:
    """The one collective of a sharded step — the all-gather of fixed-size result records — through the PRODUCT's entry
    point (include/biogpu.h: bg_gather_records_cap, csrc/comm.hip): an RCCL communicator per rank when every rank owns a
    device (the unique id travels from rank 0 over the torch.distributed store), the host-staged flavour (POSIX shared memory)
    when several ranks share one GPU (BENCH_SINGLE_GPU=1, the one-GPU test box).  torch.distributed (shard.py) only carries
    the rendezvous, the barrier and the max-over-ranks of the timing — and stays as the cross-check: `check()` gathers one
    tensor both ways and compares.  If the communicator cannot be made (no librccl.so), the legs fall back to shard.py and
    say so in `collective`."""

    def __init__(self, ctx, rank, world, stream):
        from rust_bio_amd import comm
        self.rank, self.world, self.stream, self.comm, self.kind = rank, world, stream, None, "none (1 GPU)"
        self.why = None
        if world == 1:
            return
        import torch.distributed as dist
        try:
            if shard.single_gpu_mode():
                name = "bench%s" % os.environ.get("MASTER_PORT", "0")
                self.comm = comm.Comm.host(ctx, rank, world, name)
                self.kind = "bg_gather_records (host-staged)"
            else:
                box = [comm.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                self.comm = comm.Comm.rccl(ctx, rank, world, box[0])
                self.kind = "bg_gather_records (RCCL: one ncclAllGather per step; grouped ncclBroadcast when shards are ragged)"
        except Exception as e:  # noqa: BLE001 — reported, not hidden: the line says which collective ran
            self.comm, self.why = None, repr(e)
            self.kind = "torch.distributed all_gather (fallback: %s)" % self.why
        # every rank must have taken the same route
        ok = torch.tensor([1 if self.comm is not None else 0], dtype=torch.int64)
        flags = shard.gather_records(ok.view(1, 1).to(torch.device("cuda", torch.cuda.current_device()))).cpu().view(-1).tolist()
        if self.comm is not None and not all(flags):
            self.comm.free()
            self.comm, self.kind = None, "torch.distributed all_gather (fallback: a rank has no communicator)"

    def gather(self, local, counts):
        """local: contiguous CUDA tensor [n_local, ...]; counts: records of every rank.  All records in rank order."""
        if self.world == 1:
            return local
        if self.comm is None:
            return shard.gather_records(local, counts=counts)
        total = int(sum(counts))
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        rec_bytes = local.element_size() * int(np.prod(local.shape[1:], dtype=np.int64))
        got = self.comm.gather_ptr(local.data_ptr(), int(local.shape[0]), rec_bytes, out.data_ptr(), self.stream, all_cap=total)
        assert [int(c) for c in got] == [int(c) for c in counts], (got, counts)
        return out

    def proof(self, local=None, counts=None):
        """What ran, read back from the library (bg_comm_world -> ncclCommCount / ncclCommUserRank, the path of the last gather)
        and, given a leg's records, the gather alone timed with HIP events on the stream it is queued on: the line answers
        "did RCCL see N ranks" and "what did the collective cost" by itself."""
        if self.world == 1 or self.comm is None:
            return {"rccl_ranks": 0, "why": self.why or "1 GPU"}
        out = self.comm.world_info()
        out["ragged_path_used"] = out["last_path"] == "grouped_broadcast"
        if local is not None:
            st = torch.cuda.ExternalStream(self.stream) if self.stream else torch.cuda.current_stream()
            self.gather(local, counts)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(st)
            for _ in range(5):
                self.gather(local, counts)
            e1.record(st)
            torch.cuda.synchronize()
            out["gather_ms"] = round(e0.elapsed_time(e1) / 5, 4)
            out["gather_bytes_per_rank_out"] = int(sum(counts)) * local.element_size() * int(np.prod(local.shape[1:], dtype=np.int64))
        return out

    def check(self, local, counts):
        """the same records through torch.distributed: must be identical"""
        if self.world == 1 or self.comm is None:
            return None
        a = self.gather(local, counts)
        torch.cuda.synchronize()
        return bool(torch.equal(a, shard.gather_records(local, counts=counts)))


GATHER = None


def host_cores():
    """cores this process may actually run on (cgroup / affinity aware)"""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota: threads beyond it are only throttled
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


_CSRC_SHA = None


def _newest_profile(pattern):
    """newest committed counter summary — only if it was collected with the kernel sources that run now (the files carry
    tools/csrc_hash.py's hash of rust-bio_amd/csrc): counters of edited kernels are reported as null, not as stale numbers"""
    import glob
    global _CSRC_SHA
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    d = json.load(open(files[-1]))
    if _CSRC_SHA is None:
        from csrc_hash import csrc_sha
        _CSRC_SHA = csrc_sha(ROOT)
    return d if d.get("csrc_sha") == _CSRC_SHA else None


def pmc_traffic(kernel, shape_key, shape_val):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass (tools/collect_profiles.sh:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  None when no pass has
    been recorded for this launch shape — the counters cannot be read from inside the process."""
    d = _newest_profile("r*_pmc_traffic.json")
    if not d or (d.get("launch_shape") or {}).get(shape_key) != shape_val:
        return None
    hits = [int(sum(v["mean_bytes"] for v in c.values())) for name, c in d["kernels"].items() if kernel in name]
    return max(hits) if hits else None  # several instantiations of one kernel: the one that did the work


def ingest_traffic(text_bytes):
    """HBM bytes of all the ingest kernels of one bg_fastq_parse_dev call (newest committed PMC pass)"""
    d = _newest_profile("r*_pmc_traffic.json")
    if not d or (d.get("launch_shape") or {}).get("ingest_bytes") != text_bytes or not d.get("ingest_bytes_per_call"):
        return None
    return int(d["ingest_bytes_per_call"])


def fm_big_traffic(n_q, index_bytes):
    """FETCH + WRITE bytes per launch of the FM search on the big index (its own PMC passes, tools/collect_profiles.sh)"""
    d = (_newest_profile("r*_pmc_traffic.json") or {}).get("fm_big")
    if not d or d.get("queries_per_launch") != n_q or d.get("index_bytes") != index_bytes:
        return None
    return int(sum(d[c]["mean_bytes"] for c in ("FETCH_SIZE", "WRITE_SIZE") if c in d))


def seed_extend_traffic(reads, genome):
    """FETCH + WRITE bytes of ALL kernels of one bg_seed_extend_batch_dev call (K5 on the seed windows, K6, the se_* stages,
    K1p semiglobal, K2) from the newest committed PMC passes of this leg (tools/collect_profiles.sh: the `big` passes)"""
    d = (_newest_profile("r*_pmc_traffic.json") or {}).get("seed_extend")
    if not d or d.get("reads_per_call") != reads or d.get("genome") != genome:
        return None
    return int(d["bytes_per_call"])


def valu_frac(kernel, launch_ms, shape_key, shape_val):
    """VALU issue utilisation of `kernel`: SQ_INSTS_VALU (wave instructions per launch, from the newest committed
    tools/sq_counters.sh pass of this command) x 64 lanes / launch time / 78.6 T full-rate lane-ops/s."""
    d = _newest_profile("r*_sq_counters.json")
    if not d or not launch_ms or (d.get("launch_shape") or {}).get(shape_key) != shape_val:
        return None
    hits = [c.get("SQ_INSTS_VALU") for name, c in d["kernels"].items() if kernel in name and c.get("SQ_INSTS_VALU")]
    if not hits:
        return None
    return round(max(hits) * 64.0 / (launch_ms * 1e-3) / VALU_PEAK_LANE_OPS, 4)


_GATHER = {}


def gather_ceiling(footprint_bytes, line_bytes=64):
    """The chip's measured rate for K5's access shape — quads reading two dependent random lines per step at full
    occupancy — over a table of `footprint_bytes`: tools/microbench/ub_gather64 (64-byte lines: single LF steps) or
    ub_gather128 (128-byte lines, four lanes per query: the 2-step rank blocks), built by __graft_entry__.build(), run here,
    on this GPU, next to the leg it is compared with.  Returns G lines/s or None when the binary is missing."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "microbench", "ub_gather128" if line_bytes == 128 else "ub_gather64")
    mb = round(footprint_bytes / 1e6, 1)
    if (mb, line_bytes) in _GATHER:
        return _GATHER[(mb, line_bytes)]
    val = None
    if os.path.exists(exe):
        try:
            env = dict(os.environ, HIP_VISIBLE_DEVICES=str(torch.cuda.current_device()))
            out = subprocess.run([exe, "1000", str(mb)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120, env=env)
            for ln in out.stdout.decode().splitlines():
                if ln.startswith("{"):
                    d = json.loads(ln)
                    if line_bytes == 128 and d.get("variant") == "w128x4":
                        val = d["glines_per_s"]
                    if line_bytes == 64 and d.get("dep") == 1 and d.get("lines_per_step") == 2:
                        val = d["glines_per_s"]
        except (OSError, subprocess.SubprocessError, ValueError):
            val = None
    _GATHER[(mb, line_bytes)] = val
    return val


def fm_gather_fields(fm, n_q, pat, off, bufs, stream, launch_ms, block_bytes, pattern_bytes, with_ceiling=True):
    """The FM legs' roofline.  Numerator = the bytes this kernel MUST move per launch with the block layout it runs on:
    the rank-block lines it has to fetch — counted exactly by the counting instantiation of K5 (outside the timed region):
    one line per (l - 1, r) block access, one when both ranks fall into the same block; 128-byte lines answering two LF steps
    with the 2-step blocks, 64-byte lines otherwise — x the line size, + the pattern bytes + 24 result bytes per query.
    (SURVEY.md §8(d)'s figure — two 64-byte lines per LF step — described the round-1 kernel; it is kept as the informational
    `alg_bytes_survey_per_query`, no fraction is formed from it: the 2-step kernel fetches a quarter of those lines.)
    `frac_of_gather_ceiling`: the same line rate against the chip's measured rate for this access shape
    (tools/microbench/ub_gather64 / ub_gather128) at the footprint of the blocks the search really reads."""
    d_tag, d_lo, d_hi, d_ml = bufs
    lines = fm.backward_search_count_lines_dev(n_q, pat.data_ptr(), off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(),
                                               d_ml.data_ptr(), stream)
    s2 = fm.step2_bytes()
    line_bytes = 128 if s2 else 64
    foot = s2 if s2 else block_bytes
    rate = lines / (launch_ms * 1e-3) / 1e9
    alg = float(lines) * line_bytes + float(pattern_bytes) + 24.0 * n_q
    ach = alg / (launch_ms * 1e-3) / 1e9
    out = {"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5), "alg_bytes_per_query": round(alg / n_q, 1),
           "alg_bytes_definition": "requested rank-block lines (counted) x line_bytes + pattern bytes + 24 B of results per query",
           "requested_lines_per_launch": lines, "line_bytes": line_bytes, "lf_steps_per_block_access": 2 if s2 else 1,
           "glines_per_s": round(rate, 2), "requested_gb_per_s": round(rate * line_bytes, 1)}
    if with_ceiling:
        ceil_g = gather_ceiling(foot, line_bytes)
        out.update({"gather_ceiling_glines_per_s": ceil_g,
                    "frac_of_gather_ceiling": round(rate / ceil_g, 4) if ceil_g else None,
                    "gather_ceiling_source": f"tools/microbench/{'ub_gather128 (w128x4' if s2 else 'ub_gather64 ('}two dependent random "
                                             f"{line_bytes}-B lines per quad-step, 8 waves/SIMD) on a {round(foot / 1e6)} MB table, run beside this leg"})
    return out


def timed_steps(fn, steps, warmup, device):
    for _ in range(warmup):
        fn()
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    shard.barrier()
    dt = time.perf_counter() - t0
    return shard.max_over_ranks(dt, device)


def median_time(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def kernel_timing(ctx, fn, reps=2):
    """per-kernel launch durations of `fn` from HIP events on the launch stream (outside the timed region)"""
    ctx.enable_timing(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    tm = ctx.timing()
    ctx.enable_timing(False)
    return tm


def sw_ops_equal(hrec, hops, stride, oout, oops, ostride):
    """records + every operation of every pair (device ops are right-aligned in their slot)"""
    ns = len(hrec)
    ok = all((hrec[f].astype(np.int64) == oout[f].astype(np.int64)).all()
             for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"))
    if not ok:
        return False
    kq = hrec["n_ops"].astype(np.int64)
    base = int(hrec["ops_off"][0]) + int(kq[0]) - stride  # slot 0 of this chunk
    if not (hrec["ops_off"].astype(np.int64) == base + (np.arange(ns) + 1) * stride - kq).all():
        return False
    dev_mask = np.arange(stride)[None, :] >= (stride - kq)[:, None]
    or_mask = np.arange(ostride)[None, :] < kq[:, None]
    kind = (oops.reshape(ns, ostride) & np.uint64(0xFF)).astype(np.uint8)
    return bool((hops.reshape(ns, stride)[dev_mask] == kind[or_mask]).all())


def sw_parity(orc, osc, mode, x, y, L, d_out, d_ops, stride, n_check, threads, chunk=100_000):
    """oracle pass over the first n_check pairs, chunked (the oracle's ops are 8 bytes each); returns
    (bit_exact, seconds of oracle time)"""
    ok, t_cpu = True, 0.0
    for c0 in range(0, n_check, chunk):
        k = min(chunk, n_check - c0)
        hx = x[c0 * L:(c0 + k) * L].cpu().numpy()
        hy = y[c0 * L:(c0 + k) * L].cpu().numpy()
        ho = np.arange(k + 1, dtype=np.uint64) * np.uint64(L)
        t0 = time.perf_counter()
        oout, oops, ostride = orc.align_batch(osc, mode, hx, ho, hy, ho, threads=threads)
        t_cpu += time.perf_counter() - t0
        hrec = d_out[c0 * 64:(c0 + k) * 64].cpu().numpy().view(_lib.ALN_DTYPE)
        hops = d_ops[c0 * stride:(c0 + k) * stride].cpu().numpy()
        ok = ok and sw_ops_equal(hrec, hops, stride, oout, oops, ostride)
    return bool(ok), t_cpu


def sw_roofline(kernel, fill_ms, tb_ms, pairs_per_launch, L, n_ops_mean, note, shape_key="sw_pairs_per_launch"):
    # algorithmic bytes per pair (traceback spilled to HBM):
    #   m + n + 24 + n_ops + 2 B x (m+1)(n+1) reference traceback cells
    alg = L + L + 24 + n_ops_mean + 2.0 * (L + 1) * (L + 1)
    ach = alg * pairs_per_launch / (fill_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(kernel, shape_key, int(pairs_per_launch)),
            "valu_frac": valu_frac(kernel, fill_ms, shape_key, int(pairs_per_launch)),
            "launch_ms": round(fill_ms, 4), "traceback_launch_ms": round(tb_ms, 4),
            "alg_bytes_per_pair": round(alg, 1), "pairs_per_launch": int(pairs_per_launch), "note": note}


def main():
    args = ARGS
    rank, local_rank, world = shard.init_process_group()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = _lib.Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    global GATHER
    GATHER = Gatherer(ctx, rank, world, stream)
    L = args.read_len
    n_pairs = args.pairs
    do_cpu = rank == 0 and world == 1 and not args.skip_cpu
    threads = args.cpu_threads or host_cores()
    parity = {} if do_cpu else None
    orc = None
    if do_cpu:
        import oracle_py as orc

    # ------------------------------------------------------------------ SW leg (headline)
    x, xo, y, yo = synth_gpu.sw_pairs_big(n_pairs, L, seed=2 + 100003 * rank, device=dev)
    stride = 2 * L + 4
    d_out = torch.empty(n_pairs * 64, dtype=torch.uint8, device=dev)
    d_ops = torch.empty(n_pairs * stride, dtype=torch.uint8, device=dev)
    aligner = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)

    def sw_step():
        aligner.align_dev(3, n_pairs, x.data_ptr(), xo.data_ptr(), y.data_ptr(), yo.data_ptr(),
                          L, L, d_out.data_ptr(), d_ops.data_ptr(), stride, stream)
        if world > 1:  # the single collective: scores + coordinates of every pair
            rec = d_out.view(torch.int32).view(n_pairs, 16)[:, :5].contiguous()
            GATHER.gather(rec, [n_pairs] * world)

    sw_t = timed_steps(sw_step, args.steps, args.warmup, dev)
    gcups = world * float(n_pairs) * L * L * args.steps / sw_t / 1e9
    tm = kernel_timing(ctx, sw_step)
    fill_ms = tm["fill_ms"] / max(1, tm["fill_launches"])
    tb_ms = tm["traceback_ms"] / max(1, tm["traceback_launches"])
    pairs_per_launch = n_pairs / (tm["fill_launches"] / 2)
    rec = d_out.view(torch.int32).view(n_pairs, 16)
    n_ops_mean = float(rec[:, 7].to(torch.int64).sum().item()) / n_pairs
    fill_kernel = "sw_fill_pk16_kernel" if L <= 192 else "sw_fill_kernel"  # K1p: two pairs per lane (short reads)
    roofline = sw_roofline(fill_kernel, fill_ms, tb_ms, pairs_per_launch, L, n_ops_mean,
                           "VALU-bound integer DP: `frac` is the required HBM figure on the reference's 2 B/cell "
                           "traceback, `valu_frac` the issue-rate utilisation that actually bounds it (DESIGN.md §4)")

    cpu_baseline = None
    host_api = None
    if do_cpu:
        osc = orc.make_scoring(-5, -1, 1, -1)
        n_chk = max(1, int(n_pairs * args.parity_frac))
        ok, t_par = sw_parity(orc, osc, "local", x, y, L, d_out, d_ops, stride, n_chk, threads)
        parity.update({"sw_pairs_checked": n_chk, "sw_pairs_total": n_pairs, "sw_bit_exact": ok})
        # CPU baseline: median of 3 on a bounded sample (all host threads; one thread)
        ns = min(n_pairs, 12_000 * threads)
        hx, hy = x[:ns * L].cpu().numpy(), y[:ns * L].cpu().numpy()
        ho = np.arange(ns + 1, dtype=np.uint64) * np.uint64(L)
        t_all = median_time(lambda: orc.align_batch(osc, "local", hx, ho, hy, ho, threads=threads))
        n1 = min(ns, 5000)
        t_one = median_time(lambda: orc.align_batch(osc, "local", hx[:n1 * L], ho[:n1 + 1], hy[:n1 * L], ho[:n1 + 1], threads=1))
        cpu_baseline = {"value": round(ns * L * L / t_all / 1e9, 4), "unit": "GCUPS", "cores": threads, "kind": "port",
                        "sample": f"{ns} of the {n_pairs} pairs, median of 3 runs, C++ restatement of rust-bio 4.0.1 "
                                  "Aligner::local (oracle/), one Aligner per thread; rust-bio itself cannot be built here",
                        "single_thread_value": round(n1 * L * L / t_one / 1e9, 4),
                        "full_parity_pass_value": round(n_chk * L * L / t_par / 1e9, 4)}
        # PCIe-inclusive rate of the host-buffer entry point (bg_align_batch): never the headline value
        hxa, hya = np.array(x.cpu().numpy()), np.array(y.cpu().numpy())  # the caller's own (numpy-allocated) buffers
        hoa = np.arange(n_pairs + 1, dtype=np.uint64) * np.uint64(L)
        hout, hopsb = aligner.align_arrays(3, hxa, hoa, hya, hoa)  # warm-up: sizes the staging sets, touches the result pages
        t_h = median_time(lambda: aligner.align_arrays(3, hxa, hoa, hya, hoa, out=hout, ops=hopsb))
        host_api = {"value": round(n_pairs * L * L / t_h / 1e9, 2), "unit": "GCUPS", "pairs": n_pairs, "host_threads": threads,
                    "bytes_over_pcie_GB_per_s": round((2 * n_pairs * L + n_pairs * 64 + float(n_ops_mean) * n_pairs) / t_h / 1e9, 1),
                    "note": "bg_align_batch with pageable host buffers in and out (PCIe-inclusive), median of 3: stages of "
                            "122880 pairs through three pinned staging sets; uploads by copy commands, records and device-compacted "
                            "operations brought back by a small kernel on a high-priority stream, all overlapped with the next stages' kernels"}
        del hout, hopsb, hxa, hya

    result = {"metric": "GCUPS (SW) + FM-index queries/sec", "value": round(gcups, 3), "unit": "GCUPS",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(sw_t / args.steps * 1e3, 3), "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "int16" if L <= 192 else "int32", "data": "synthetic",
              "config": {"workload": f"{n_pairs} x {L} bp synthetic read pairs per GPU, Aligner::local "
                                     "affine-gap (-5,-1,+1,-1), score+coords+traceback ops (BASELINE configs[1])",
                         "pairs_per_gpu": n_pairs, "read_len": L, "parallelism": f"shard{world}"},
              "roofline": roofline}
    if world > 1:  # the headline step's collective, proven from the library's side (bg_comm_world) and timed alone
        rec5 = d_out.view(torch.int32).view(n_pairs, 16)[:, :5].contiguous()
        result["collective"] = GATHER.kind
        result["collective_proof"] = GATHER.proof(rec5, [n_pairs] * world)
        del rec5
    if host_api:
        result["host_api"] = host_api
    # A/B: the same pairs as 2-bit streams (bg_pack2_dev + bg_align_batch_packed_dev: K1p loads codes instead of bytes)
    if not args.skip_packed:
        from rust_bio_amd import pack2
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xpk, bad_x = pack2.pack_dev(x, ctx=ctx, stream=stream)
        ypk, bad_y = pack2.pack_dev(y, ctx=ctx, stream=stream)
        torch.cuda.synchronize()
        pack_ms = (time.perf_counter() - t0) * 1e3
        d_outp, d_opsp = torch.empty_like(d_out), torch.empty_like(d_ops)

        def swp_step():
            aligner.align_packed_dev(3, n_pairs, xpk.data_ptr(), xo.data_ptr(), ypk.data_ptr(), yo.data_ptr(), L, L,
                                     d_outp.data_ptr(), d_opsp.data_ptr(), stride, stream=stream)

        tp = timed_steps(swp_step, args.steps, args.warmup, dev)
        tmp_ = kernel_timing(ctx, swp_step)
        same = bool(torch.equal(d_out, d_outp)) and bad_x == 0 and bad_y == 0
        kq = rec[:, 7].to(torch.int64)
        for c0 in range(0, n_pairs, 1 << 18):
            k = min(1 << 18, n_pairs - c0)
            m = torch.arange(stride, device=dev)[None, :] >= (stride - kq[c0:c0 + k])[:, None]
            same = same and bool((d_ops.view(n_pairs, stride)[c0:c0 + k][m] == d_opsp.view(n_pairs, stride)[c0:c0 + k][m]).all())
        result["packed2"] = {"value": round(world * float(n_pairs) * L * L * args.steps / tp / 1e9, 3), "unit": "GCUPS",
                             "ms_per_step": round(tp / args.steps * 1e3, 3),
                             "fill_launch_ms": round(tmp_["fill_ms"] / max(1, tmp_["fill_launches"]), 4),
                             "byte_flavour_ms_per_step": round(sw_t / args.steps * 1e3, 3),
                             "pack_ms_both_streams": round(pack_ms, 3),
                             "records_and_ops_equal_byte_run": same,
                             "config": {"workload": "the headline pairs as 2-bit streams (16 symbols per dword) through "
                                                    "bg_align_batch_packed_dev: A/B against the byte flavour above"}}
        if parity is not None:
            parity["sw_packed2_equals_bytes_all_pairs"] = same
        del xpk, ypk, d_outp, d_opsp
    # the same workload at the reference's arithmetic width: K1 (int32 recurrence) on the very same pairs
    if not args.skip_k1:
        ctx.set_option("no_pk16", 1)
        d_out32, d_ops32 = torch.empty_like(d_out), torch.empty_like(d_ops)

        def sw32_step():
            aligner.align_dev(3, n_pairs, x.data_ptr(), xo.data_ptr(), y.data_ptr(), yo.data_ptr(),
                              L, L, d_out32.data_ptr(), d_ops32.data_ptr(), stride, stream)

        t32 = timed_steps(sw32_step, args.steps, args.warmup, dev)
        tm32 = kernel_timing(ctx, sw32_step)
        ctx.set_option("no_pk16", 0)
        # every record byte and every operation byte equal to the int16 run (which the oracle pass above covers pair by pair)
        same = bool(torch.equal(d_out, d_out32))
        kq = rec[:, 7].to(torch.int64)
        for c0 in range(0, n_pairs, 1 << 18):
            k = min(1 << 18, n_pairs - c0)
            m = torch.arange(stride, device=dev)[None, :] >= (stride - kq[c0:c0 + k])[:, None]
            same = same and bool((d_ops.view(n_pairs, stride)[c0:c0 + k][m] == d_ops32.view(n_pairs, stride)[c0:c0 + k][m]).all())
        f32 = tm32["fill_ms"] / max(1, tm32["fill_launches"])
        result["value_int32"] = round(world * float(n_pairs) * L * L * args.steps / t32 / 1e9, 3)
        result["int32"] = {"value": result["value_int32"], "unit": "GCUPS", "dtype": "int32",
                           "ms_per_step": round(t32 / args.steps * 1e3, 3),
                           "config": {"workload": "the headline workload (same pairs, same scoring) through K1, the int32 kernel: "
                                                  "ctx option no_pk16 = 1"},
                           "records_and_ops_equal_int16_run": same,
                           "roofline": sw_roofline("sw_fill_kernel<10, 16, 0, true, true, true>", f32,  # (R, LP, SM, LOCAL, NARROW, LF)
                                                   tm32["traceback_ms"] / max(1, tm32["traceback_launches"]),
                                                   n_pairs / (tm32["fill_launches"] / 2), L, n_ops_mean,
                                                   "K1 (int32): VALU-bound", shape_key="k1_pairs_per_launch")}
        if parity is not None:
            parity["sw_int32_equals_int16_all_pairs"] = same
        del d_out32, d_ops32
    # north_star names Aligner::semiglobal next to ::local (mod.rs:954-984): the same pairs through the semiglobal flavour of
    # K1p (the general packed kernel: no LF shortcut), its own roofline, every pair against the oracle
    if not args.skip_semiglobal:
        d_outs, d_opss = torch.empty_like(d_out), torch.empty_like(d_ops)

        def sg_step():
            aligner.align_dev(2, n_pairs, x.data_ptr(), xo.data_ptr(), y.data_ptr(), yo.data_ptr(),
                              L, L, d_outs.data_ptr(), d_opss.data_ptr(), stride, stream)

        tsg = timed_steps(sg_step, args.steps, args.warmup, dev)
        tmsg = kernel_timing(ctx, sg_step)
        fsg = tmsg["fill_ms"] / max(1, tmsg["fill_launches"])
        n_ops_sg = float(d_outs.view(torch.int32).view(n_pairs, 16)[:, 7].to(torch.int64).sum().item()) / n_pairs
        sg = {"value": round(world * float(n_pairs) * L * L * args.steps / tsg / 1e9, 3), "unit": "GCUPS",
              "dtype": "int16" if L <= 192 else "int32", "ms_per_step": round(tsg / args.steps * 1e3, 3),
              "config": {"workload": f"{n_pairs} x {L} bp read pairs per GPU (the headline pairs), Aligner::semiglobal "
                                     "affine-gap (-5,-1,+1,-1), score+coords+traceback ops"},
              "roofline": sw_roofline("sw_fill_pk16_kernel<10, 16, 2, 2, 0, 0, false>" if L <= 192 else "sw_fill_kernel", fsg,
                                      tmsg["traceback_ms"] / max(1, tmsg["traceback_launches"]),
                                      n_pairs / (tmsg["fill_launches"] / 2), L, n_ops_sg,
                                      "K1p, semiglobal flavour (x-suffix-clip fold and clip candidates kept): VALU-bound",
                                      shape_key="sg_pairs_per_launch")}
        if do_cpu:
            n_chk = max(1, int(n_pairs * args.parity_frac))
            oksg, t_parsg = sw_parity(orc, osc, "semiglobal", x, y, L, d_outs, d_opss, stride, n_chk, threads)
            parity.update({"semiglobal_pairs_checked": n_chk, "semiglobal_bit_exact": oksg})
            sg["cpu_baseline"] = {"value": round(n_chk * L * L / t_parsg / 1e9, 4), "unit": "GCUPS", "cores": threads, "kind": "port",
                                  "sample": f"{n_chk} of the {n_pairs} pairs (the parity pass), oracle Aligner::semiglobal"}
        result["semiglobal"] = sg
        del d_outs, d_opss
    del x, y, d_ops, d_out
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ int32 kernel legs (K1)
    if not args.skip_k1:
        result["k1_int32"] = k1_legs(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity)

    # ------------------------------------------------------------------ FM leg
    if not args.skip_fm:
        fm_legs(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity, result)

    if args.fm_big_genome:
        fm_big_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity, result)

    # ------------------------------------------------------------------ FMD-index SMEM leg
    if not args.skip_fmd:
        result["fmd_smems"] = fmd_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, parity)

    # ------------------------------------------------------------------ banded leg (configs[3] shape)
    if not args.skip_banded:
        result["banded"] = banded_leg(args, ctx, dev, rank, world, do_cpu, orc, threads, parity)

    # ------------------------------------------------------------------ ingest leg
    if not args.skip_ingest:
        result["ingest"] = ingest_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, parity)

    if rank == 0:
        if cpu_baseline is not None:
            result["cpu_baseline"] = cpu_baseline
        if parity is not None:
            result["parity"] = parity
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def fmd_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, parity):
    """FMDIndex::all_smems (fmindex.rs:479-501; smems 363-434, backward_ext / forward_ext 527-564) for a batch of reads on an
    FMD index over T$R$ (fmindex.rs:311-340), K7 (csrc/fmd_smems.hip) through bg_fmd_smems_batch64_dev: text, suffix array,
    BWT and index built on the device; a sample of the reads against the oracle's restatement, which is also the CPU baseline
    (one core: the reference's FMDIndex is a single-threaded API)."""
    from rust_bio_amd.suffix_array import bwt_dev, suffix_array_dev
    G, NR, L, MINLEN, CAP = args.fmd_genome, args.fmd_reads, 150, 20, 24
    N = 2 * G + 2
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a_, b_ in zip(b"ACGTN", b"TGCAN"):
        comp[a_] = b_
    t0 = time.perf_counter()
    g = torch.empty(N, dtype=torch.uint8, device=dev)
    g[:G] = synth_gpu.genome(G, seed=51, device=dev)[:G]
    g[G] = ord("$")
    g[G + 1:N - 1] = comp[g[:G].flip(0).to(torch.int64)]
    g[N - 1] = ord("$")
    d_sa = suffix_array_dev(g, ctx=ctx)
    d_b = bwt_dev(g, d_sa, ctx=ctx)
    del d_sa
    fm = FMIndex.from_device(d_b, 128, N_ALPHABET, ctx=ctx)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    z = synth_gpu.splitmix64(4321 + 100003 * rank, NR, dev)
    start = torch.remainder(z & ((1 << 62) - 1), G - L)
    ar = torch.arange(L, dtype=torch.int64, device=dev)
    reads = g[start.view(-1, 1) + ar.view(1, -1)].clone()
    u = synth_gpu.splitmix64(99, NR * L, dev).view(NR, L)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    reads = torch.where((u & 0xFFFF) < int(0.03 * 65536), acgt[(u >> 20) & 3], reads)
    rc = torch.arange(NR, device=dev) % 2 == 1
    reads[rc] = comp[reads[rc].flip(1).to(torch.int64)]
    reads = reads.contiguous()
    r_off = torch.arange(NR + 1, dtype=torch.int64, device=dev) * L
    d_cnt = torch.zeros(NR, dtype=torch.int32, device=dev)
    d_out = torch.zeros((NR, CAP, 6), dtype=torch.int64, device=dev)

    def step():
        _lib.check(_lib.lib().bg_fmd_smems_batch64_dev(fm.h, 1, NR, reads.data_ptr(), r_off.data_ptr(), None, MINLEN, L, CAP,
                                                       d_cnt.data_ptr(), d_out.data_ptr(), stream), "bg_fmd_smems_batch64_dev")
        if world > 1:  # the single collective: the SMEM count of every read
            GATHER.gather(d_cnt.view(-1, 1), [NR] * world)

    t = timed_steps(step, args.steps, args.warmup, dev)
    cnt = d_cnt.cpu().numpy().astype(np.int64)
    n_smem = int(cnt.sum())
    # every extension (backward_ext, fmindex.rs:527-558) ranks two rows = two 64-byte lines; a read takes at least L - 1
    # forward extensions and as many backward ones as its SMEM lists hold: 2 (L - 1) is a LOWER bound of the extensions per read
    alg = NR * (L + 2.0 * (L - 1) * 128 + 4) + n_smem * 48.0
    leg = {"value": round(world * NR * args.steps / t, 1), "unit": "reads/s", "ms_per_step": round(t / args.steps * 1e3, 3), "scaling": "weak",
           "config": {"workload": f"FMDIndex over T$R$ of a {G} bp synthetic genome ({N} symbols), all_smems(min length {MINLEN}) of {NR} x {L} bp "
                                  "reads per GPU (3 % substitutions, half reverse-complemented)", "index_bytes": fm.device_bytes(),
                      "text_to_index_s": round(build_s, 2)},
           "smems": n_smem, "reads_over_cap": int((cnt > CAP).sum()),
           "roofline": {"bound": "hbm", "kernel": "fmd_smems_kernel<false, true>", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "achieved": round(alg / (t / args.steps) / 1e9, 1), "traffic": None, "achieved_is_a_lower_bound": True,
                        "note": "two random 64-byte rank lines per bi-interval extension, at least 2 (L - 1) extensions per read; the walk is "
                                "latency-bound (one dependent block access per extension, a quad of lanes per read)"}}
    leg["roofline"]["frac"] = round(leg["roofline"]["achieved"] / HBM_PEAK_GBS, 4)
    if do_cpu:
        ns = min(NR, 3000)
        h_b = d_b.cpu().numpy()
        ls = np.asarray(fm._less, dtype=np.uint64)
        ofmd = orc.FMDIndex(h_b, ls, orc.Occ(h_b, 128, N_ALPHABET))
        h_reads = reads[:ns].cpu().numpy()
        out = d_out[:ns].cpu().numpy().astype(np.uint64)
        ofmd.ext_calls(reset=True)
        t0 = time.perf_counter()
        want = [ofmd.all_smems(h_reads[q].tobytes(), MINLEN) for q in range(ns)]
        ct = time.perf_counter() - t0
        # the extensions the ALGORITHM makes (counted in the oracle's restatement on the sample): the roofline's numerator
        ext_per_read = ofmd.ext_calls() / ns
        alg = NR * (L + ext_per_read * 128 + 4) + n_smem * 48.0
        leg["roofline"].update({"achieved": round(alg / (t / args.steps) / 1e9, 1), "achieved_is_a_lower_bound": False,
                                "extensions_per_read": round(ext_per_read, 1),
                                "note": "two random 64-byte rank lines per bi-interval extension; extensions per read counted in the oracle's "
                                        "restatement on the parity sample; the walk is latency-bound (dependent block accesses, a quad of lanes per read)"})
        leg["roofline"]["frac"] = round(leg["roofline"]["achieved"] / HBM_PEAK_GBS, 4)
        ok = all([((int(r[0]), int(r[1]), int(r[2]), int(r[3])), int(r[4]), int(r[5])) for r in out[q, :cnt[q]]] == want[q] for q in range(ns))
        parity["fmd_reads_checked"] = ns
        parity["fmd_bit_exact"] = bool(ok)
        leg["cpu_baseline"] = {"value": round(ns / ct, 1), "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": f"{ns} of the reads, oracle FMDIndex::all_smems (fmindex.rs:363-564 restated), one call per read"}
    del g, d_b, fm, reads, d_out
    torch.cuda.empty_cache()
    return leg


def k1_legs(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity):
    """The general kernel K1 (int32 DP values, LDS score table): what runs when K1p's conditions do not hold —
    tabulated match functions (BLOSUM62 protein pairs, pairwise/mod.rs:1309-1337) and scores beyond 12 bits."""
    L, n = 150, args.k1_pairs
    stride = 2 * L + 4
    mat = np.zeros((256, 256), dtype=np.int32)
    for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "blosum62.json")))["pairs"].items():
        mat[ord(k[0]), ord(k[1])] = v
    legs = {}
    cases = [("blosum62_protein", synth_gpu.protein_pairs(n, L, seed=12 + 100003 * rank, device=dev),
              Scoring.new(-5, -1, mat), dict(matrix=mat),
              f"{n} x {L} aa protein pairs per GPU, Aligner::local, blosum62 tabulated (27 byte classes in LDS), gaps -5/-1"),
             ("wide_scores_dna", synth_gpu.sw_pairs_big(n, L, seed=13 + 100003 * rank, device=dev),
              Scoring.from_scores(-500, -100, 100, -100), dict(match=100, mismatch=-100),
              f"{n} x {L} bp DNA pairs per GPU, Aligner::local, from_scores(-500,-100,100,-100): scores beyond K1p's 12 bits")]
    # template arguments <R, LP, SM, LOCAL, NARROW> of the instantiation each case runs (profile lookups go by name)
    knames = {"blosum62_protein": "sw_fill_kernel<10, 16, 1, true, true, false>",
              "wide_scores_dna": "sw_fill_kernel<10, 16, 0, true, true, true>"}  # R = 10 rows x 16 lanes, SM, LOCAL, NARROW, LF
    for name, (x, xo, y, yo), scoring, okw, desc in cases:
        kname = knames[name]
        d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_ops = torch.empty(n * stride, dtype=torch.uint8, device=dev)
        al = Aligner.with_scoring(scoring, ctx=ctx)

        def step():
            al.align_dev(3, n, x.data_ptr(), xo.data_ptr(), y.data_ptr(), yo.data_ptr(), L, L, d_out.data_ptr(),
                         d_ops.data_ptr(), stride, stream)

        t = timed_steps(step, args.steps, args.warmup, dev)
        tm = kernel_timing(ctx, step)
        fill_ms = tm["fill_ms"] / max(1, tm["fill_launches"])
        tb_ms = tm["traceback_ms"] / max(1, tm["traceback_launches"])
        ppl = n / (tm["fill_launches"] / 2)
        n_ops_mean = float(d_out.view(torch.int32).view(n, 16)[:, 7].to(torch.int64).sum().item()) / n
        leg = {"value": round(world * float(n) * L * L * args.steps / t / 1e9, 2), "unit": "GCUPS", "dtype": "int32",
               "ms_per_step": round(t / args.steps * 1e3, 3), "config": {"workload": desc},
               "roofline": sw_roofline(kname, fill_ms, tb_ms, ppl, L, n_ops_mean,
                                       "K1 (int32): VALU-bound", shape_key="k1_pairs_per_launch")}
        if do_cpu:
            go, ge = scoring.gap_open, scoring.gap_extend
            osc = orc.make_scoring(go, ge, **okw)
            n_chk = max(1, int(min(n, 100_000) * args.parity_frac))
            ok, t_par = sw_parity(orc, osc, "local", x, y, L, d_out, d_ops, stride, n_chk, threads)
            parity[f"k1_{name}_pairs_checked"] = n_chk
            parity[f"k1_{name}_bit_exact"] = ok
            leg["cpu_baseline"] = {"value": round(n_chk * L * L / t_par / 1e9, 4), "unit": "GCUPS", "cores": threads,
                                   "kind": "port", "sample": f"{n_chk} of the {n} pairs (the parity pass), oracle Aligner::local"}
        legs[name] = leg
        del x, y, d_out, d_ops
    torch.cuda.empty_cache()
    return legs


def build_index(args, ctx, dev, n_genome, seed, want_sa, want_host):
    """Genome (SplitMix64, generated in HBM) -> suffix array -> BWT -> FM index + rate-32 suffix-array samples, all on the
    device and on every rank's own GPU: bg_suffix_array_dev / bg_bwt_dev / bg_sa_sample_dev / bg_fm_build_dev (1 Gbp in
    well under a second, nothing text-sized crosses PCIe unless the oracle legs want host copies).  `--host-sa` runs
    rust-bio's host-side order of things instead (bg_suffix_array: SA-IS on host cores, ~100 s per Gbp; bg_bwt; bg_less;
    bg_fm_build laying the rank blocks out with host threads)."""
    from rust_bio_amd.suffix_array import SampledSuffixArray, bwt_dev, sample_dev, suffix_array_dev
    t = {}
    t0 = time.perf_counter()
    g_dev = synth_gpu.genome(n_genome, seed=seed, device=dev)
    torch.cuda.synchronize()
    g = b = sa = None
    t1 = time.perf_counter()
    if args.host_sa:
        g = g_dev.cpu().numpy()
        t1 = time.perf_counter()
        sa = suffix_array(g)
        t["suffix_array_s"] = time.perf_counter() - t1
        t2 = time.perf_counter()
        b = bwt(g, sa)
        ssa = SampledSuffixArray(sa, g, b, 32)
        ls = less(b, N_ALPHABET)
        fm = FMIndex(b, ls, Occ(b, 128, N_ALPHABET), ctx=ctx)
        if not want_sa:
            sa = None
    else:
        d_sa = suffix_array_dev(g_dev, ctx=ctx)
        torch.cuda.synchronize()
        t["suffix_array_s"] = time.perf_counter() - t1
        t2 = time.perf_counter()
        d_b = bwt_dev(g_dev, d_sa, ctx=ctx)
        ssa = sample_dev(d_sa, d_b, ord("$"), 32, ctx=ctx)
        fm = FMIndex.from_device(d_b, 128, N_ALPHABET, ctx=ctx)
        ls = fm._less
        torch.cuda.synchronize()
        t["bwt_samples_blocks_s"] = time.perf_counter() - t2
        if want_host:  # the oracle legs search the BWT / walk the suffix array / cut windows on the host
            g = g_dev.cpu().numpy()
            b = d_b.cpu().numpy()
            if want_sa:
                sa = d_sa.cpu().numpy().view(np.uint32)  # the oracle walks it as it is (orc_seed_extend_batch_sa32)
        fm._d_bwt = None
        del d_sa, d_b
    ssa.attach(fm)
    t.setdefault("bwt_samples_blocks_s", time.perf_counter() - t2)
    t["total_s"] = time.perf_counter() - t0
    t["where"] = "host (SA-IS, --host-sa)" if args.host_sa else "device (prefix doubling over radix sorts)"
    return g_dev, g, sa, b, ls, fm, t


def fm_packed_leg(args, ctx, dev, stream, world, fm, n_q, pat, off, bufs, byte_ms, parity, key):
    """A/B: the same patterns as a 2-bit stream (bg_pack2_dev with the index's codes, bg_fm_backward_search_packed_dev: a dword
    load per 16 steps, symbols are codes already) against the byte flavour whose results sit in `bufs`"""
    from rust_bio_amd import pack2
    d_tag, d_lo, d_hi, d_ml = bufs
    ref = (d_tag.clone(), d_lo.clone(), d_hi.clone(), d_ml.clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ppk, bad = pack2.pack_dev(pat, codes=fm.pattern_codes(), ctx=ctx, stream=stream)
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) * 1e3

    def step():
        fm.backward_search_packed_dev(n_q, ppk.data_ptr(), off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(),
                                      d_ml.data_ptr(), stream)

    tpk = timed_steps(step, args.steps, args.warmup, dev)
    tmk = kernel_timing(ctx, step)
    same = bad == 0 and all(bool(torch.equal(a_, b_)) for a_, b_ in zip(ref, (d_tag, d_lo, d_hi, d_ml)))
    if parity is not None:
        parity[f"{key}_packed2_equals_bytes_all_queries"] = same
    return {"value": round(world * float(n_q) * args.steps / tpk, 1), "unit": "queries/s",
            "ms_per_step": round(tpk / args.steps * 1e3, 3), "launch_ms": round(tmk["fm_ms"] / max(1, tmk["fm_launches"]), 4),
            "byte_flavour_launch_ms": round(byte_ms, 4), "pack_ms": round(pack_ms, 3), "results_equal_byte_run": same,
            "config": {"workload": "the same patterns as a 2-bit stream (bg_pack2_dev with bg_fm_pattern_codes, "
                                   "bg_fm_backward_search_packed_dev): A/B against the byte flavour"}}


def fm_legs(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity, result):
    L = args.read_len
    pipeline_here = not args.skip_pipeline and not args.fm_big_genome  # else the seed-and-extend leg runs on the big index
    g_dev, g, sa, b, ls, fm, bt = build_index(args, ctx, dev, args.genome, 3, want_sa=do_cpu and pipeline_here, want_host=do_cpu)
    n_q, P = args.queries, args.pattern_len
    pat, off = synth_gpu.fm_patterns(g_dev, n_q, P, seed=4 + 100003 * rank)
    d_tag = torch.empty(n_q, dtype=torch.uint8, device=dev)
    d_lo = torch.empty(n_q, dtype=torch.int64, device=dev)
    d_hi = torch.empty(n_q, dtype=torch.int64, device=dev)
    d_ml = torch.empty(n_q, dtype=torch.int32, device=dev)

    def fm_step():
        fm.backward_search_dev(n_q, pat.data_ptr(), off.data_ptr(), d_tag.data_ptr(),
                               d_lo.data_ptr(), d_hi.data_ptr(), d_ml.data_ptr(), stream)
        if world > 1:  # the single collective: intervals of every query
            GATHER.gather(torch.stack((d_lo, d_hi), dim=1), [n_q] * world)

    fm_t = timed_steps(fm_step, args.steps, args.warmup, dev)
    qps = world * float(n_q) * args.steps / fm_t
    tm = kernel_timing(ctx, fm_step)
    fm_ms = tm["fm_ms"] / max(1, tm["fm_launches"])
    ml = d_ml.to(torch.int64)
    steps_exec = int((ml + (d_tag == 1).to(torch.int64) + (d_tag == 2).to(torch.int64)).sum().item())
    # SURVEY.md §8(d)'s per-query figure (|P| + 24 + 128 x LF steps executed: two 64-byte lines per step) — informational only
    survey_bytes = float(n_q) * (P + 24) + 128.0 * steps_exec
    fm_res = {"value": round(qps, 1), "unit": "queries/s", "ms_per_step": round(fm_t / args.steps * 1e3, 3), "scaling": "weak",
              "config": {"workload": f"FMIndex over {args.genome} bp synthetic genome + '$' (n_alphabet, Occ k=128), "
                                     f"{n_q} x {P} bp backward_search per GPU (BASELINE configs[2])",
                         "index_bytes": fm.device_bytes(), "index_build_s": round(bt["total_s"], 2),
                         "suffix_array_s": round(bt["suffix_array_s"], 3), "suffix_array_on": bt["where"],
                         "index_build": "every rank builds its replica on its GPU: genome, suffix array, BWT, SA samples, rank "
                                        "blocks (bg_suffix_array_dev, bg_bwt_dev, bg_sa_sample_dev, bg_fm_build_dev)"},
              "tags": {"complete": int((d_tag == 0).sum().item()), "partial": int((d_tag == 1).sum().item()),
                       "absent": int((d_tag == 2).sum().item())},
              "roofline": {"bound": "hbm", "kernel": "fm_search_fast2x_kernel", "achieved": None,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                           "traffic": pmc_traffic("fm_search_fast2x_kernel<false, false, false", "fm_queries_per_launch", n_q),  # (the timed instantiation, not the counted one)
                           "launch_ms": round(fm_ms, 4), "queries_per_launch": n_q,
                           "alg_bytes_survey_per_query": round(survey_bytes / n_q, 1),
                           "note": "the 33 MB block index of a 100 Mbp text sits in the 256 MiB Infinity Cache: see fm_big "
                                   "for an index that cannot"}}

    block_bytes = (args.genome + 1 + 191) // 192 * 64
    fm_res["roofline"].update(fm_gather_fields(fm, n_q, pat, off, (d_tag, d_lo, d_hi, d_ml), stream, fm_ms, block_bytes,
                                               float(n_q) * P, with_ceiling=rank == 0))
    if not args.skip_packed:
        fm_res["packed2"] = fm_packed_leg(args, ctx, dev, stream, world, fm, n_q, pat, off, (d_tag, d_lo, d_hi, d_ml), fm_ms, parity, "fm")
    # strong scaling on configs[2]: the SAME n_q queries in total, split over the ranks, gathered inside the step
    q_lo, q_hi = shard.partition(n_q, rank, world)
    if world > 1:  # every rank must search the same global query set: regenerate it from rank 0's seed
        pat_s, off_s = synth_gpu.fm_patterns(g_dev, n_q, P, seed=4)
    else:
        pat_s, off_s = pat, off
    my = q_hi - q_lo
    counts = [shard.partition(n_q, r, world)[1] - shard.partition(n_q, r, world)[0] for r in range(world)]
    s_off = (off_s[q_lo:q_hi + 1] - off_s[q_lo]).contiguous()
    s_pat = pat_s[int(off_s[q_lo].item()):int(off_s[q_hi].item())].contiguous()
    holder = {}

    def fm_strong_step():
        fm.backward_search_dev(my, s_pat.data_ptr(), s_off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(),
                               d_hi.data_ptr(), d_ml.data_ptr(), stream)
        rec = torch.stack((d_lo[:my], d_hi[:my], d_ml[:my].to(torch.int64) | (d_tag[:my].to(torch.int64) << 32)), dim=1)
        holder["rec"] = rec
        holder["all"] = GATHER.gather(rec, counts)  # 24-byte records {lower, upper, tag|matched_len}

    st_t = timed_steps(fm_strong_step, args.steps, args.warmup, dev)
    fm_res["strong"] = {"value": round(float(n_q) * args.steps / st_t, 1), "unit": "queries/s", "scaling": "strong",
                        "ms_per_step": round(st_t / args.steps * 1e3, 3), "queries_total": n_q, "queries_per_gpu": my,
                        "collective": GATHER.kind, "record_bytes": 24,
                        "gathered_records": int(holder["all"].shape[0]), "collective_proof": GATHER.proof(holder["rec"], counts)}
    if world == 1 and n_q >= 8:
        # an eighth of the queries in one call: what one GPU of eight sees of configs[2] in the strong leg — bounds the 8-GPU
        # strong-scaling efficiency of this leg from one GPU (launch of 1.25 M queries against the steady rate of 10 M)
        qe = n_q // 8

        def eighth_step():
            fm.backward_search_dev(qe, pat.data_ptr(), off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), d_ml.data_ptr(), stream)

        te = timed_steps(eighth_step, args.steps, args.warmup, dev)
        fm_res["strong"]["eighth_of_the_batch"] = {"queries": qe, "ms": round(te / args.steps * 1e3, 3), "queries_per_s": round(qe * args.steps / te, 1),
                                                   "frac_of_full_batch_rate": round((qe * args.steps / te) / (float(n_q) * args.steps / st_t), 3)}
        fm_step()  # the full batch's results back for the checks below
    if world > 1:  # the gathered records of the sharded run must be the unsharded answer
        fm_res["strong"]["capi_gather_equals_torch_gather"] = GATHER.check(holder["rec"], counts)
        fm.backward_search_dev(n_q, pat_s.data_ptr(), off_s.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(),
                               d_hi.data_ptr(), d_ml.data_ptr(), stream)
        full = torch.stack((d_lo, d_hi, d_ml.to(torch.int64) | (d_tag.to(torch.int64) << 32)), dim=1)
        fm_res["strong"]["sharded_equals_unsharded"] = bool((full == holder["all"]).all().item())
        fm_step()  # restore this rank's own results for the checks below
        del pat_s, off_s
    del s_pat, s_off, holder

    if do_cpu:
        occ = orc.Occ(b, 128, N_ALPHABET)
        n_chk = max(1, int(n_q * args.parity_frac))
        hp = pat[:n_chk * P].cpu().numpy()
        hoff = np.arange(n_chk + 1, dtype=np.uint64) * np.uint64(P)
        t0 = time.perf_counter()
        otag, olo, ohi, oml = orc.backward_search_batch(b, ls, occ, hp, hoff, threads=threads)
        t_par = time.perf_counter() - t0
        ok = bool((d_tag[:n_chk].cpu().numpy() == otag).all() and
                  (d_lo[:n_chk].cpu().numpy().astype(np.uint64) == olo).all() and
                  (d_hi[:n_chk].cpu().numpy().astype(np.uint64) == ohi).all() and
                  (d_ml[:n_chk].cpu().numpy().astype(np.uint64) == oml).all())
        parity.update({"fm_queries_checked": n_chk, "fm_queries_total": n_q, "fm_bit_exact": ok})
        del otag, olo, ohi, oml
        nsq = min(n_q, 60_000 * threads)
        t_all = median_time(lambda: orc.backward_search_batch(b, ls, occ, hp[:nsq * P], hoff[:nsq + 1], threads=threads))
        n1 = max(1, nsq // 16)
        t_one = median_time(lambda: orc.backward_search_batch(b, ls, occ, hp[:n1 * P], hoff[:n1 + 1], threads=1))
        fm_res["cpu_baseline"] = {"value": round(nsq / t_all, 1), "unit": "queries/s", "cores": threads, "kind": "port",
                                  "sample": f"{nsq} of the {n_q} queries, median of 3 runs, C++ restatement of rust-bio 4.0.1 "
                                            "backward_search + Occ::get (oracle/), shared index",
                                  "single_thread_value": round(n1 / t_one, 1),
                                  "full_parity_pass_value": round(n_chk / t_par, 1)}
        # PCIe-inclusive figure of the host-buffer entry point (bg_fm_backward_search_batch)
        hp_all = pat.cpu().numpy()
        hoff_all = np.arange(n_q + 1, dtype=np.uint64) * np.uint64(P)
        h_out = fm.backward_search_arrays(hp_all, hoff_all)  # warm-up: sizes the staging sets, pages the result arrays in
        t_h = median_time(lambda: fm.backward_search_arrays(hp_all, hoff_all, out=h_out))
        fm_res["host_api"] = {"value": round(n_q / t_h, 1), "unit": "queries/s", "queries": n_q,
                              "caller_bytes_read_GB_per_s": round(n_q * P / t_h / 1e9, 1), "host_threads": threads,
                              "note": "bg_fm_backward_search_batch: pageable host buffers in and out (PCIe-inclusive), median of 3; "
                                      "stages of 2^20 queries through three pinned staging sets, the caller's result arrays reused. "
                                      "Bound by the host: its threads read the caller's pattern bytes once to pack them to 2 bits "
                                      "(BG_TRACE_HOST=1: pack 12.5 of 15.3 ms per 10 M queries of 100 bp on 16 threads = 80 GB/s "
                                      "of pageable memory) - caller_bytes_read_GB_per_s is what explains box-to-box spread"}
        del h_out
        del hp, hp_all
    result["fm"] = fm_res
    del pat, off, d_tag, d_lo, d_hi, d_ml

    # -------------------------------------------------------------- seed-and-extend leg (configs[4] shape)
    if pipeline_here:
        result["seed_extend"] = seed_extend_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity,
                                                fm, g_dev, g, sa, b, ls, args.genome)
    del sa, g_dev, fm
    torch.cuda.empty_cache()


def fm_big_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity, result):
    """backward_search on an index larger than the 256 MiB Infinity Cache — every rank is an HBM read — and the
    seed-and-extend leg on the same genome (BASELINE configs[4]: 3 Gbp with --fm-big-genome 3000000000)."""
    import resource
    n_g, n_q, P = args.fm_big_genome, args.queries, args.pattern_len
    want_sa = do_cpu and not args.skip_pipeline
    g_dev, g, sa, b, ls, fm, bt = build_index(args, ctx, dev, n_g, 33, want_sa=want_sa, want_host=do_cpu)
    pat, off = synth_gpu.fm_patterns(g_dev, n_q, P, seed=34 + 100003 * rank)
    d_tag = torch.empty(n_q, dtype=torch.uint8, device=dev)
    d_lo = torch.empty(n_q, dtype=torch.int64, device=dev)
    d_hi = torch.empty(n_q, dtype=torch.int64, device=dev)
    d_ml = torch.empty(n_q, dtype=torch.int32, device=dev)

    def step():
        fm.backward_search_dev(n_q, pat.data_ptr(), off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(),
                               d_ml.data_ptr(), stream)
        if world > 1:
            GATHER.gather(torch.stack((d_lo, d_hi), dim=1), [n_q] * world)

    t = timed_steps(step, args.steps, args.warmup, dev)
    tm = kernel_timing(ctx, step)
    ms = tm["fm_ms"] / max(1, tm["fm_launches"])
    steps_exec = int((d_ml.to(torch.int64) + (d_tag == 1).to(torch.int64) + (d_tag == 2).to(torch.int64)).sum().item())
    survey_bytes = float(n_q) * (P + 24) + 128.0 * steps_exec  # SURVEY.md §8(d)'s figure: informational (fm_gather_fields)
    leg = {"value": round(world * float(n_q) * args.steps / t, 1), "unit": "queries/s", "ms_per_step": round(t / args.steps * 1e3, 3),
           "scaling": "weak",
           "config": {"workload": f"FMIndex over {n_g} bp synthetic genome + '$', {n_q} x {P} bp backward_search per GPU, "
                                  "index beyond the Infinity Cache, jump table off",
                      "index_bytes": fm.device_bytes(), "index_build_s": round(bt["total_s"], 2),
                      "suffix_array_s": round(bt["suffix_array_s"], 3), "suffix_array_on": bt["where"],
                      "bwt_samples_blocks_s": round(bt["bwt_samples_blocks_s"], 2),
                      "host_peak_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)},
           "tags": {"complete": int((d_tag == 0).sum().item()), "partial": int((d_tag == 1).sum().item())},
           "roofline": {"bound": "hbm", "kernel": "fm_search_fast2x_kernel", "achieved": None,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                        "traffic": fm_big_traffic(n_q, fm.device_bytes()),
                        "launch_ms": round(ms, 4), "queries_per_launch": n_q,
                        "alg_bytes_survey_per_query": round(survey_bytes / n_q, 1)}}
    leg["roofline"].update(fm_gather_fields(fm, n_q, pat, off, (d_tag, d_lo, d_hi, d_ml), stream, ms, (n_g + 1 + 191) // 192 * 64,
                                            float(n_q) * P, with_ceiling=rank == 0))
    if not args.skip_packed:
        leg["packed2"] = fm_packed_leg(args, ctx, dev, stream, world, fm, n_q, pat, off, (d_tag, d_lo, d_hi, d_ml), ms, parity, "fm_big")
    if do_cpu:
        occ = orc.Occ(b, 128, N_ALPHABET)
        n_chk = max(1, int(min(n_q, 1_000_000) * args.parity_frac))
        hp = pat[:n_chk * P].cpu().numpy()
        hoff = np.arange(n_chk + 1, dtype=np.uint64) * np.uint64(P)
        t0 = time.perf_counter()
        otag, olo, ohi, oml = orc.backward_search_batch(b, ls, occ, hp, hoff, threads=threads)
        t_par = time.perf_counter() - t0
        ok = bool((d_tag[:n_chk].cpu().numpy() == otag).all() and (d_lo[:n_chk].cpu().numpy().astype(np.uint64) == olo).all() and
                  (d_hi[:n_chk].cpu().numpy().astype(np.uint64) == ohi).all() and
                  (d_ml[:n_chk].cpu().numpy().astype(np.uint64) == oml).all())
        parity.update({"fm_big_queries_checked": n_chk, "fm_big_queries_total": n_q, "fm_big_bit_exact": ok})
        leg["cpu_baseline"] = {"value": round(n_chk / t_par, 1), "unit": "queries/s", "cores": threads, "kind": "port",
                               "sample": f"{n_chk} of the {n_q} queries (the parity pass), oracle backward_search"}
        del occ
    result["fm_big"] = leg
    del pat, off, d_tag, d_lo, d_hi, d_ml
    if not args.skip_pipeline:
        result["seed_extend"] = seed_extend_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity,
                                                fm, g_dev, g, sa, b, ls, n_g)
    del sa, g_dev, fm
    torch.cuda.empty_cache()


def seed_extend_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, threads, parity, fm, g_dev, g, sa, b, ls, n_genome):
    from rust_bio_amd.pipeline import SeedParams, attach_text, seed_extend_dev
    L, Rp = args.read_len, args.pipeline_reads
    reads, r_starts = synth_gpu.reads_from_genome(g_dev, Rp, L, seed=5 + 100003 * rank)
    d_roff = torch.arange(Rp + 1, dtype=torch.int64, device=dev) * L
    attach_text(fm, d_text=g_dev)
    prm = SeedParams(20, 10, 16, 25)
    sc = Scoring.from_scores(-5, -1, 1, -1)
    stride = 2 * L + 2 * prm.pad + 4
    d_hits = torch.empty(Rp * 96, dtype=torch.uint8, device=dev)
    d_ops = torch.empty(Rp * stride, dtype=torch.uint8, device=dev)
    tot = np.zeros(2, dtype=np.uint64)

    def pipe_step():
        seed_extend_dev(fm, sc, Rp, reads.data_ptr(), d_roff.data_ptr(), L, d_hits.data_ptr(), d_ops.data_ptr(), stride, prm,
                        stream, tot)
        if world > 1:  # the single collective: score + reference span of every read (24-byte records)
            h64 = d_hits.view(torch.int64).view(Rp, 12)
            GATHER.gather(torch.stack((h64[:, 0] & 0xFFFFFFFF, h64[:, 9], h64[:, 10]), dim=1), [Rp] * world)

    pipe_t = timed_steps(pipe_step, args.steps, args.warmup, dev)
    tm = kernel_timing(ctx, pipe_step, reps=1)
    hv = d_hits.view(torch.int32).view(Rp, 24)
    score = hv[:, 0]
    mapped = score > -(1 << 29)
    ref_start = d_hits.view(torch.int64).view(Rp, 12)[:, 9]
    near = ((ref_start - r_starts).abs() <= 8) & mapped
    n_ops_total = int(hv[:, 7].to(torch.int64).sum().item())
    C = int(tot[1])
    # algorithmic bytes (SURVEY.md section 8d formulas of the three calls): seeds |P| + 24 + 128 x LF steps (20 per voting
    # seed at most: counted as executed), 8 B per located row, and per candidate the aligner's m + n + 24 + 2 (m+1)(n+1)
    n_seeds = Rp * ((L - prm.seed_len) // prm.stride + 1)
    alg = n_seeds * (prm.seed_len + 24 + 128.0 * prm.seed_len) + 8.0 * int(tot[0]) + \
        C * (L + (L + 2 * prm.pad) + 24 + 2.0 * (L + 1) * (L + 2 * prm.pad + 1)) + n_ops_total
    ms = pipe_t / args.steps * 1e3
    leg = {"value": round(world * Rp * args.steps / pipe_t, 1), "unit": "reads/s", "ms_per_step": round(ms, 3),
           "config": {"workload": f"{Rp} x {L} bp reads per GPU (5% sub, 1% ins, 1% del) vs the {n_genome} bp genome through "
                                  "bg_seed_extend_batch_dev: 20-bp seeds at stride 10 -> backward_search -> Interval::occ (sampled SA, "
                                  "rate 32, intervals <= 16 rows) -> Aligner::semiglobal on +-25 bp windows -> best hit + its operations "
                                  "(BASELINE configs[4]: 10 M reads vs 3 Gbp over 8 GPUs; --fm-big-genome sets the genome)",
                      "genome": n_genome},
           "seed_hits": int(tot[0]), "candidates": C,
           "mapped_frac": round(float(mapped.float().mean().item()), 4),
           "mapped_at_origin_frac": round(float(near.float().mean().item()), 4),
           "kernel_ms": {"seed_search": round(tm["fm_ms"], 3), "align_fill": round(tm["fill_ms"], 3),
                         "align_traceback": round(tm["traceback_ms"], 3)},
           "roofline": {"bound": "hbm", "kernel": "whole pipeline (K5 seeds, K6 locate, gather, K1p semiglobal, K2, best hit)",
                        "achieved": round(alg / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": seed_extend_traffic(Rp, n_genome),
                        "calls_in_run": args.warmup + args.steps + 1,  # (weak leg; tools/pmc_summary.py divides the PMC totals by it)
                        "alg_bytes_per_read": round(alg / Rp, 1),
                        "note": "dominated by the candidates' semiglobal fill (VALU-bound like the headline kernel)"}}
    # strong scaling on configs[4]: the SAME reads in total (10 M), split over the ranks, one all-gather of 24-byte
    # records {score, ref_start, ref_end} inside the step; the index is a replica on every GPU
    Rt = args.pipeline_reads_total
    if Rt:
        s_steps, s_warm = max(2, args.steps // 4), 1
        r_lo, r_hi = shard.partition(Rt, rank, world)
        my = r_hi - r_lo
        counts = [shard.partition(Rt, r, world)[1] - shard.partition(Rt, r, world)[0] for r in range(world)]
        # every rank draws the global read set from rank 0's seed (chunked generator) and keeps its contiguous share
        g_reads, _ = synth_gpu.reads_from_genome(g_dev, Rt, L, seed=5)
        s_reads = g_reads[r_lo * L:r_hi * L].contiguous() if world > 1 else g_reads
        s_roff = torch.arange(my + 1, dtype=torch.int64, device=dev) * L
        s_hits = torch.empty(max(my, 1) * 96, dtype=torch.uint8, device=dev)
        s_ops = torch.empty(max(my, 1) * stride, dtype=torch.uint8, device=dev)  # same work as the weak leg: operations included
        holder = {}

        def strong_step():
            seed_extend_dev(fm, sc, my, s_reads.data_ptr(), s_roff.data_ptr(), L, s_hits.data_ptr(), s_ops.data_ptr(), stride, prm, stream, None)
            h64 = s_hits.view(torch.int64).view(-1, 12)[:my]
            holder["rec"] = torch.stack((h64[:, 0] & 0xFFFFFFFF, h64[:, 9], h64[:, 10]), dim=1)
            holder["all"] = GATHER.gather(holder["rec"], counts)

        st_t = timed_steps(strong_step, s_steps, s_warm, dev)
        leg["strong"] = {"value": round(float(Rt) * s_steps / st_t, 1), "unit": "reads/s", "scaling": "strong",
                         "ms_per_step": round(st_t / s_steps * 1e3, 3), "steps": s_steps, "reads_total": Rt, "reads_per_gpu": my,
                         "collective": GATHER.kind, "record_bytes": 24,
                         "gathered_records": int(holder["all"].shape[0]), "collective_proof": GATHER.proof(holder["rec"], counts),
                         "note": "records only (score + reference span); the winners' operations stay on the rank that "
                                 "computed them (INTEGRATION.md section 3)"}
        if world == 1:
            # an eighth of the reads in one call: what one GPU of eight sees of configs[4] in the strong leg — bounds the 8-GPU
            # strong figure from one GPU (the weak leg above when it has Rt / 8 reads, else a call of its own)
            Re = Rt // 8
            if Re == Rp:
                t_e = pipe_t / args.steps
            else:
                e_hits = torch.empty(max(Re, 1) * 96, dtype=torch.uint8, device=dev)
                e_ops = torch.empty(max(Re, 1) * stride, dtype=torch.uint8, device=dev)

                def eighth_step():
                    seed_extend_dev(fm, sc, Re, s_reads.data_ptr(), s_roff.data_ptr(), L, e_hits.data_ptr(), e_ops.data_ptr(), stride, prm, stream, None)

                t_e = timed_steps(eighth_step, s_steps, s_warm, dev) / s_steps
                del e_hits, e_ops
            leg["strong"]["eighth_of_the_batch"] = {"reads": Re, "ms": round(t_e * 1e3, 3), "reads_per_s": round(Re / t_e, 1),
                                                    "frac_of_full_batch_rate": round((Re / t_e) / (float(Rt) * s_steps / st_t), 3)}
        if world > 1:  # the gathered records of the sharded run must be the unsharded answer
            leg["strong"]["capi_gather_equals_torch_gather"] = GATHER.check(holder["rec"], counts)
            f_roff = torch.arange(Rt + 1, dtype=torch.int64, device=dev) * L
            f_hits = torch.empty(Rt * 96, dtype=torch.uint8, device=dev)
            seed_extend_dev(fm, sc, Rt, g_reads.data_ptr(), f_roff.data_ptr(), L, f_hits.data_ptr(), 0, 0, prm, stream, None)
            f64 = f_hits.view(torch.int64).view(Rt, 12)
            full = torch.stack((f64[:, 0] & 0xFFFFFFFF, f64[:, 9], f64[:, 10]), dim=1)
            leg["strong"]["sharded_equals_unsharded"] = bool((full == holder["all"]).all().item())
            del f_roff, f_hits, f64, full
        del g_reads, s_reads, s_roff, s_hits, s_ops, holder
        torch.cuda.empty_cache()
    if do_cpu:
        occ = orc.Occ(b, 128, N_ALPHABET)
        osc = orc.make_scoring(-5, -1, 1, -1)
        n_chk = max(2, int(min(Rp, 100_000) * args.parity_frac))
        # the first and the LAST reads of the batch: the call walks the reads in passes, the tail belongs to the last one
        segs = [(0, n_chk // 2), (Rp - (n_chk - n_chk // 2), n_chk - n_chk // 2)]
        hv64 = d_hits.view(torch.int64).view(Rp, 12)
        # every read's operations end at the end of its own slot of the caller's buffer (biogpu.h)
        ok = bool((hv64[:, 4] == (torch.arange(Rp, device=dev) + 1) * stride - hv[:, 7].to(torch.int64)).all().item())
        t_par = 0.0
        for r0, k in segs:
            hr = reads[r0 * L:(r0 + k) * L].cpu().numpy()
            ho = np.arange(k + 1, dtype=np.uint64) * np.uint64(L)
            t0 = time.perf_counter()
            ohits, oops, ostride = orc.seed_extend_batch(b, ls, occ, sa, g, n_genome, osc, hr, ho, threads=threads)
            t_par += time.perf_counter() - t0
            hits = d_hits[r0 * 96:(r0 + k) * 96].cpu().numpy().view(_lib.SEED_HIT_DTYPE)
            ok = ok and all((hits[f] == ohits[f]).all() for f in ("n_candidates", "n_seed_hits", "window_start", "ref_start", "ref_end"))
            ok = ok and all((hits["aln"][f].astype(np.int64) == ohits["aln"][f].astype(np.int64)).all()
                            for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"))
            if ok:  # the winners' operations (right-aligned in their slot)
                hops = d_ops[r0 * stride:(r0 + k) * stride].cpu().numpy().reshape(k, stride)
                kq = hits["aln"]["n_ops"].astype(np.int64)
                dev_mask = np.arange(stride)[None, :] >= (stride - kq)[:, None]
                or_mask = np.arange(ostride)[None, :] < kq[:, None]
                kind = (oops.reshape(k, ostride) & np.uint64(0xFF)).astype(np.uint8)
                ok = bool((hops[dev_mask] == kind[or_mask]).all())
        parity.update({"seed_extend_reads_checked": n_chk, "seed_extend_reads_total": Rp, "seed_extend_bit_exact": bool(ok),
                       "seed_extend_sample": "first and last n/2 reads + ops_off of every read"})
        hr = reads[:n_chk * L].cpu().numpy()
        ho = np.arange(n_chk + 1, dtype=np.uint64) * np.uint64(L)
        ns = min(n_chk, 2_000 * threads)
        t_all = median_time(lambda: orc.seed_extend_batch(b, ls, occ, sa, g, n_genome, osc, hr[:ns * L], ho[:ns + 1],
                                                          threads=threads, want_ops=False))
        leg["cpu_baseline"] = {"value": round(ns / t_all, 1), "unit": "reads/s", "cores": threads, "kind": "port",
                               "sample": f"{ns} of the {Rp} reads, median of 3 runs: the same composition out of the oracle's "
                                         "backward_search, Interval::occ (raw suffix array) and Aligner::semiglobal (oracle/pipeline.cpp)",
                               "full_parity_pass_value": round(n_chk / t_par, 1)}
    return leg


def bcells_of(bal):
    return float(bal.last_cells.sum())


def banded_leg(args, ctx, dev, rank, world, do_cpu, orc, threads, parity):
    from rust_bio_amd.banded import Aligner as BandedAligner
    Pb, Lb, kb, wb = args.banded_pairs, 10_000, 16, 32
    if args.banded_chunk:
        ctx.set_option("chunk_pairs", args.banded_chunk)
    bx, bxo, by, byo = synth_gpu.sw_pairs_big(Pb, Lb, seed=4 + 100003 * rank, device=dev, sub=0.06, ins=0.02,
                                               dele=0.02, chunk=64)
    hx, hy = bx.cpu().numpy(), by.cpu().numpy()
    hoff = np.arange(Pb + 1, dtype=np.uint64) * np.uint64(Lb)
    bal = BandedAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), kb, wb, ctx=ctx)
    # warm-up at full size: sizes the pinned staging and device scratch, and touches the pages of the caller's result buffers
    bout, bops = bal.align_arrays(2, hx, hoff, hy, hoff)
    shard.barrier()
    t0 = time.perf_counter()
    bout, bops = bal.align_arrays(2, hx, hoff, hy, hoff, out=bout, ops=bops)
    bt = shard.max_over_ranks(time.perf_counter() - t0, dev)
    # device-resident flavour: sequences, records and operation slots stay in HBM
    d_boff = torch.arange(Pb + 1, dtype=torch.int64, device=dev) * Lb
    bstride = 2 * Lb + 8
    d_bout = torch.empty(Pb * 64, dtype=torch.uint8, device=dev)
    d_bops = torch.empty(Pb * bstride, dtype=torch.uint8, device=dev)
    bal.align_dev(2, Pb, bx.data_ptr(), d_boff.data_ptr(), by.data_ptr(), d_boff.data_ptr(), d_bout.data_ptr(),
                  d_bops.data_ptr(), bstride)
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bal.align_dev(2, Pb, bx.data_ptr(), d_boff.data_ptr(), by.data_ptr(), d_boff.data_ptr(), d_bout.data_ptr(),
                  d_bops.data_ptr(), bstride)
    if world > 1:  # the single collective: scores + coordinates of every pair
        GATHER.gather(d_bout.view(torch.int32).view(Pb, 16)[:, :5].contiguous(), [Pb] * world)
    torch.cuda.synchronize()
    bt_dev = shard.max_over_ranks(time.perf_counter() - t0, dev)
    dev_ok = bool((d_bout.view(torch.int32).view(Pb, 16)[:, 0].cpu().numpy() == bout["score"]).all())
    # an eighth of the batch in one call: what one GPU of eight sees of configs[3] in the strong leg (a single sub-batch:
    # band construction, fill and traceback back to back — nothing to overlap with)
    eighth = None
    if world == 1 and Pb >= 8:
        Pe = Pb // 8
        bal.align_dev(2, Pe, bx.data_ptr(), d_boff.data_ptr(), by.data_ptr(), d_boff.data_ptr(), d_bout.data_ptr(), d_bops.data_ptr(), bstride)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bal.align_dev(2, Pe, bx.data_ptr(), d_boff.data_ptr(), by.data_ptr(), d_boff.data_ptr(), d_bout.data_ptr(), d_bops.data_ptr(), bstride)
        torch.cuda.synchronize()
        te = time.perf_counter() - t0
        eighth = {"pairs": Pe, "ms": round(te * 1e3, 2), "pairs_per_s": round(Pe / te, 1),
                  "frac_of_full_batch_rate": round((Pe / te) / (Pb / bt_dev), 3),
                  "scores_equal_full_batch": bool((d_bout.view(torch.int32).view(Pb, 16)[:Pe, 0].cpu().numpy() == bout["score"][:Pe]).all())}
        # (the records of the full batch come back below: the timing pass rewrites them)
    # strong scaling on configs[3]: the SAME Pb pairs in total (100 k), split over the ranks, one all-gather of the
    # 20-byte records {score, xstart, xend, ystart, yend} inside the step.  With one GPU the split is the whole batch:
    # the device-resident run above is that measurement.
    strong = {"value": round(bcells_of(bal) / bt_dev / 1e9, 3), "unit": "GCUPS (band cells)", "scaling": "strong",
              "pairs_total": Pb, "pairs_per_gpu": Pb, "pairs_per_s": round(Pb / bt_dev, 1), "collective": "none (1 GPU)",
              "note": "1 GPU: the same run as device_resident"}
    if world > 1:
        gx, _, gy, _ = (bx, None, by, None) if rank == 0 else synth_gpu.sw_pairs_big(Pb, Lb, seed=4, device=dev, sub=0.06, ins=0.02,
                                                                                     dele=0.02, chunk=64)[:4]
        p_lo, p_hi = shard.partition(Pb, rank, world)
        my = p_hi - p_lo
        counts = [shard.partition(Pb, r, world)[1] - shard.partition(Pb, r, world)[0] for r in range(world)]
        sx, sy = gx[p_lo * Lb:p_hi * Lb].contiguous(), gy[p_lo * Lb:p_hi * Lb].contiguous()
        holder = {}

        def strong_step():
            holder["cells"] = bal.align_dev(2, my, sx.data_ptr(), d_boff.data_ptr(), sy.data_ptr(), d_boff.data_ptr(),
                                            d_bout.data_ptr(), d_bops.data_ptr(), bstride, want_cells=True)
            holder["rec"] = d_bout.view(torch.int32).view(Pb, 16)[:my, :5].contiguous()
            holder["all"] = GATHER.gather(holder["rec"], counts)

        strong_step()
        my_cells = torch.tensor([float(holder["cells"].sum())], dtype=torch.float64, device=dev)
        tot_cells = float(shard.gather_records(my_cells.view(1, 1)).sum().item())
        shard.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        strong_step()
        torch.cuda.synchronize()
        st_t = shard.max_over_ranks(time.perf_counter() - t0, dev)
        sharded = holder["all"].clone()
        bal.align_dev(2, Pb, gx.data_ptr(), d_boff.data_ptr(), gy.data_ptr(), d_boff.data_ptr(), d_bout.data_ptr(),
                      d_bops.data_ptr(), bstride)
        torch.cuda.synchronize()
        strong = {"value": round(tot_cells / st_t / 1e9, 3), "unit": "GCUPS (band cells)", "scaling": "strong",
                  "pairs_total": Pb, "pairs_per_gpu": my, "pairs_per_s": round(Pb / st_t, 1),
                  "collective": GATHER.kind, "record_bytes": 20, "collective_proof": GATHER.proof(holder["rec"], counts),
                  "capi_gather_equals_torch_gather": GATHER.check(holder["rec"], counts),
                  "sharded_equals_unsharded": bool((sharded == d_bout.view(torch.int32).view(Pb, 16)[:, :5]).all().item())}
        del gx, gy, sx, sy
    del d_bout, d_bops, bx, by
    # kernel durations from a second, event-timed pass (timing serialises the K3/K4/host pipeline)
    ctx.enable_timing(True)
    bal.align_arrays(2, hx, hoff, hy, hoff)
    tm = ctx.timing()
    ctx.enable_timing(False)
    redo_pairs = ctx.band_redo_pairs()  # pairs the packed-int16 fill (K3p) flagged and the int32 kernels recomputed
    bcells = float(bal.last_cells.sum())
    # algorithmic bytes per pair: m + n + 8(n+1) + 2 x band_cells + 24 + n_ops
    balg = float(Pb) * (2 * Lb + 8 * (Lb + 1) + 24) + 2.0 * bcells + float(bout["n_ops"].sum())
    bfill_s = tm["fill_ms"] * 1e-3
    # the fill runs in sub-batches of 16 384 pairs (one round of K3v2 blocks); a last, short one runs K3 (one pair per wavefront)
    Pb_launch = min(Pb, 16384)
    fill_launch_ms = tm["fill_ms"] / max(1.0, Pb / 16384.0)
    banded = {"value": round(world * bcells / bt / 1e9, 3), "unit": "GCUPS (band cells, host-buffer API: PCIe + band "
              "construction on the device + fill + traceback)",
              "pairs_per_s": round(world * Pb / bt, 1),
              "device_resident": {"value": round(world * bcells / bt_dev / 1e9, 3), "unit": "GCUPS (band cells)",
                                  "pairs_per_s": round(world * Pb / bt_dev, 1), "scores_equal_host_api": dev_ok,
                                  "eighth_of_the_batch": eighth},
              "config": {"workload": f"{Pb} x 10 kb read pairs per GPU (6% sub, 2% ins, 2% del), banded::Aligner::"
                                     f"semiglobal, k-mer {kb}, w {wb} (BASELINE configs[3]: 100k x 10 kb)",
                         "mean_band_cells": round(bcells / Pb, 1)},
              "strong": strong,
              "kernel_ms": {"fill": round(tm["fill_ms"], 2), "traceback": round(tm["traceback_ms"], 2)},
              "dp_only_gcups": round(bcells / ((tm["fill_ms"] + tm["traceback_ms"]) * 1e-3) / 1e9, 2),
              "packed_fill": {"kernel": "banded_fill2p_kernel<2, 16> (K3p: interior strips, two pairs per lane group, 16-bit keys "
                                        "relative to a per-strip base)", "pairs_recomputed_by_int32_kernels": redo_pairs},
              "host_threads": host_cores(),
              "roofline": {"bound": "hbm", "kernel": "banded_fill", "achieved": round(balg / bfill_s / 1e9, 2),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(balg / bfill_s / 1e9 / HBM_PEAK_GBS, 5),
                           "traffic": pmc_traffic("banded_fill", "banded_pairs_per_launch", Pb_launch),
                           "valu_frac": valu_frac("banded_fill", fill_launch_ms, "banded_pairs_per_launch", Pb_launch),
                           "launch_ms": round(fill_launch_ms, 3),
                           "alg_bytes_per_pair": round(balg / Pb, 1)},
              "pairs_per_launch": Pb_launch}
    if do_cpu:
        nsb = min(Pb, max(8, int(args.banded_parity_pairs * args.parity_frac)))  # >= 1 % of the 100 000 pairs
        osc = orc.make_scoring(-5, -1, 1, -1)
        t0 = time.perf_counter()
        oout, oops, ostride, ocells = orc.banded_align_batch(osc, "semiglobal", kb, wb, hx[:nsb * Lb], hoff[:nsb + 1],
                                                             hy[:nsb * Lb], hoff[:nsb + 1], threads=min(nsb, threads))
        t_par = time.perf_counter() - t0
        okb = bool(all((bout[f][:nsb].astype(np.int64) == oout[f].astype(np.int64)).all()
                       for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops")) and
                   (bal.last_cells[:nsb] == ocells).all())
        kind = (oops.reshape(nsb, ostride) & 0xFF).astype(np.uint8)
        for p in range(nsb):
            kq, oq = int(bout["n_ops"][p]), int(bout["ops_off"][p])
            okb = okb and bool((bops[oq:oq + kq] == kind[p, :kq]).all())
        parity.update({"banded_pairs_checked": nsb, "banded_pairs_total": Pb, "banded_bit_exact": okb})
        # >= 512 pairs: with 4 pairs per thread the 200 MB traceback matrix + memset of a 10 kb pair's first run on every
        # thread dominated (0.16 GCUPS on 64 pairs against 0.49 over the parity pass — the same code)
        # (2048 pairs: ~4 s per run on 16 threads; 512 still read 0.43 against the parity pass's 0.60 — every thread's first pair
        #  pays the page faults of its 200 MB matrix)
        nt = min(nsb, max(2048, 4 * threads))
        t_all = median_time(lambda: orc.banded_align_batch(osc, "semiglobal", kb, wb, hx[:nt * Lb], hoff[:nt + 1],
                                                           hy[:nt * Lb], hoff[:nt + 1], threads=min(nt, threads), want_ops=False))
        banded["cpu_baseline"] = {"value": round(float(ocells[:nt].sum()) / t_all / 1e9, 4), "unit": "GCUPS (band cells)",
                                  "cores": min(nt, threads), "kind": "port",
                                  "sample": f"{nt} of the {Pb} pairs, median of 3 runs, C++ restatement of rust-bio 4.0.1 "
                                            "banded::Aligner::semiglobal incl. band construction (oracle/)",
                                  "full_parity_pass_value": round(float(ocells.sum()) / t_par / 1e9, 4)}
    if args.banded_chunk:
        ctx.set_option("chunk_pairs", 0)
        banded["config"]["pairs_per_sub_batch"] = args.banded_chunk
    return banded


def ingest_leg(args, ctx, dev, stream, rank, world, do_cpu, orc, parity):
    from rust_bio_amd import fastq as bgfastq
    L = args.read_len
    n_fq = args.ingest_reads
    text = synth.fastq_text(n_fq, L, seed=6 + 100003 * rank)
    d_text = torch.from_numpy(text).to(dev)
    fq_bufs = bgfastq.alloc_dev(len(text), dev)  # the caller's buffers, reused by every call
    bgfastq.parse_dev(d_text, ctx=ctx, stream=stream, bufs=fq_bufs)  # warm-up: sizes the scratch
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        k, st, _, d_recs, d_seq, d_so, d_qual, d_qo = bgfastq.parse_dev(d_text, ctx=ctx, stream=stream, bufs=fq_bufs)
    torch.cuda.synchronize()
    it = shard.max_over_ranks((time.perf_counter() - t0) / 3, dev)
    seq_bytes = int(d_so[-1].item())
    ingest = {"value": round(world * len(text) / it / 1e9, 2), "unit": "GB/s of FASTQ text (device-resident text -> records, "
              "concatenated sequences/qualities + offsets, Record::check)", "reads_per_s": round(world * k / it, 1),
              "config": {"workload": f"{n_fq} four-line records of {L} bp per GPU ({len(text)} bytes)"},
              "status": st, "records": int(k),
              "roofline": {"bound": "hbm", "achieved": round((len(text) + 2 * seq_bytes + 56 * k + 16 * k) / it / 1e9, 2),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "traffic": ingest_traffic(len(text)),
                           "kernel": "fq_fused_kernel",
                           "note": "algorithmic bytes: text once + sequences, qualities, records and offsets written; one kernel "
                                   "(decoupled look-back over 16 KB tiles) and one host synchronisation per call"}}
    ingest["roofline"]["frac"] = round(ingest["roofline"]["achieved"] / HBM_PEAK_GBS, 5)
    if do_cpu:
        ns = min(n_fq, 200_000)
        sample = synth.fastq_text(ns, L, seed=6)
        t0 = time.perf_counter()
        want, wst, _ = orc.fastq_parse_raw(sample)
        ct = time.perf_counter() - t0
        got = bgfastq.parse_arrays(sample, ctx=ctx)
        okf = wst == 0 and len(got) == len(want[0]) and bytes(got.seq) == want[1] and bytes(got.qual) == want[2]
        parity["ingest_sample_records"] = ns
        parity["ingest_bit_exact"] = bool(okf)
        ingest["cpu_baseline"] = {"value": round(len(sample) / ct / 1e9, 3), "unit": "GB/s of FASTQ text", "cores": 1, "kind": "port",
                                  "sample": f"{ns} records, C++ restatement of bio::io::fastq::Reader::read + Record::check (oracle/)"}
    del d_text, fq_bufs
    return ingest


if __name__ == "__main__":
    main()
