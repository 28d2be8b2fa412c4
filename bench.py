#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: GCUPS (Smith-Waterman) + FM-index queries/s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM:
  * headline (`value`): BASELINE configs[1] — 1 M x 150 bp synthetic read pairs per GPU through
    `Aligner::local` (affine gaps, Scoring::from_scores(-5,-1,1,-1)): K1 fill + K2 traceback,
    score + coordinates + full operation list for every pair;
  * second leg (`fm`): BASELINE configs[2] — FMIndex over a 100 Mbp synthetic genome,
    10 M x 100 bp backward_search per GPU.
Units (pairs / queries) shard across ranks, the index is replicated, and each step ends with
the single all-gather of fixed-size result records (world_size > 1 only).  `scaling` is weak:
every rank processes its own full batch.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

from rust_bio_amd import _lib, shard, synth, synth_gpu
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.fmindex import FMIndex
from rust_bio_amd.pairwise import Aligner, Scoring
from rust_bio_amd.suffix_array import suffix_array

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
N_ALPHABET = b"ACGTNacgtn"


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


def pmc_traffic(kernel, shape_key, shape_val):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass (tools/collect_profiles.sh:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  None when no pass has
    been recorded for this launch shape — the counters cannot be read from inside the process."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    if (d.get("launch_shape") or {}).get(shape_key) != shape_val:
        return None
    hits = [int(sum(v["mean_bytes"] for v in c.values())) for name, c in d["kernels"].items() if kernel in name]
    return max(hits) if hits else None  # several instantiations of one kernel: the one that did the work


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


def main():
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
    ap.add_argument("--skip-banded", action="store_true")
    ap.add_argument("--banded-pairs", type=int, default=32768, help="banded leg: 10 kb pairs per GPU (configs[3] is 100k over 8 GPUs)")
    ap.add_argument("--skip-pipeline", action="store_true")
    ap.add_argument("--pipeline-reads", type=int, default=1_000_000, help="seed-and-extend leg: reads per GPU")
    ap.add_argument("--skip-ingest", action="store_true")
    ap.add_argument("--ingest-reads", type=int, default=1_000_000, help="FASTQ ingest leg: four-line records per GPU")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    rank, local_rank, world = shard.init_process_group()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = _lib.Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    L = args.read_len
    n_pairs = args.pairs

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
            shard.gather_records(rec, counts=[n_pairs] * world)

    sw_t = timed_steps(sw_step, args.steps, args.warmup, dev)
    cells_per_step = float(n_pairs) * L * L
    gcups = world * cells_per_step * args.steps / sw_t / 1e9

    # kernel-level timing with HIP events on the launch stream (outside the timed region)
    ctx.enable_timing(True)
    for _ in range(2):
        sw_step()
    torch.cuda.synchronize()
    tm = ctx.timing()
    ctx.enable_timing(False)
    fill_ms = tm["fill_ms"] / max(1, tm["fill_launches"])
    tb_ms = tm["traceback_ms"] / max(1, tm["traceback_launches"])
    launches_per_step = tm["fill_launches"] / 2
    rec = d_out.view(torch.int32).view(n_pairs, 16)
    n_ops_total = int(rec[:, 7].to(torch.int64).sum().item())
    pairs_per_launch = n_pairs / launches_per_step
    # algorithmic bytes per pair (traceback spilled to HBM):
    #   m + n + 24 + n_ops + 2 B x (m+1)(n+1) reference traceback cells
    alg_bytes_pair = L + L + 24 + n_ops_total / n_pairs + 2.0 * (L + 1) * (L + 1)
    achieved = alg_bytes_pair * pairs_per_launch / (fill_ms * 1e-3) / 1e9
    fill_kernel = "sw_fill_pk16_kernel" if L <= 192 else "sw_fill_kernel"  # K1p: two pairs per lane (short reads)
    roofline = {"bound": "hbm", "kernel": fill_kernel, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic(fill_kernel, "sw_pairs_per_launch", int(pairs_per_launch)),
                "launch_ms": round(fill_ms, 4),
                "traceback_launch_ms": round(tb_ms, 4),
                "alg_bytes_per_pair": round(alg_bytes_pair, 1),
                "pairs_per_launch": int(pairs_per_launch),
                "note": "VALU-bound integer DP; HBM fraction reported as required, see DESIGN.md"}

    # parity of a sample against the oracle + CPU baseline on the same sample (rank 0)
    # the CPU legs (oracle parity sample + cpu_baseline) run on rank 0 of the single-GPU run only
    do_cpu = rank == 0 and world == 1 and not args.skip_cpu
    parity = {} if do_cpu else None
    cpu_baseline = None
    if do_cpu:
        import oracle_py as orc
        threads = args.cpu_threads or host_cores()
        # bounded sample: ~10-60 CPU-seconds of work spread over all host cores
        ns = min(n_pairs, max(40_000, 1500 * threads))
        hx, hy = x[:ns * L].cpu().numpy(), y[:ns * L].cpu().numpy()
        ho = np.arange(ns + 1, dtype=np.uint64) * np.uint64(L)
        osc = orc.make_scoring(-5, -1, 1, -1)
        t0 = time.perf_counter()
        oout, oops, ostride = orc.align_batch(osc, "local", hx, ho, hy, ho, threads=threads)
        t_all = time.perf_counter() - t0
        n1 = min(ns, 5000)
        t0 = time.perf_counter()
        orc.align_batch(osc, "local", hx[:n1 * L], ho[:n1 + 1], hy[:n1 * L], ho[:n1 + 1], threads=1)
        t_one = time.perf_counter() - t0
        hrec = d_out[:ns * 64].cpu().numpy().view(_lib.ALN_DTYPE)
        hops = d_ops[:ns * stride].cpu().numpy().reshape(ns, stride)
        ok = all((hrec[f].astype(np.int64) == oout[f].astype(np.int64)).all()
                 for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"))
        if ok:  # every operation of every sampled pair (device ops are right-aligned in their slot)
            kq = hrec["n_ops"].astype(np.int64)
            ok = bool((hrec["ops_off"].astype(np.int64) == (np.arange(ns) + 1) * stride - kq).all())
            dev_mask = np.arange(stride)[None, :] >= (stride - kq)[:, None]
            or_mask = np.arange(ostride)[None, :] < kq[:, None]
            kind = (oops.reshape(ns, ostride) & np.uint64(0xFF)).astype(np.uint8)
            ok = ok and bool((hops[dev_mask] == kind[or_mask]).all())
        parity.update({"sw_sample_pairs": ns, "sw_bit_exact": bool(ok)})
        # PCIe-inclusive rate of the host-buffer entry point (bg_align_batch): never the headline value
        nh = n_pairs
        hxa, hya = x[:nh * L].cpu().numpy(), y[:nh * L].cpu().numpy()
        hoa = np.arange(nh + 1, dtype=np.uint64) * np.uint64(L)
        hout, hopsb = aligner.align_arrays(3, hxa, hoa, hya, hoa)  # warm-up: sizes the staging sets, touches the result pages
        t0 = time.perf_counter()
        aligner.align_arrays(3, hxa, hoa, hya, hoa, out=hout, ops=hopsb)
        host_api_gcups = nh * L * L / (time.perf_counter() - t0) / 1e9
        del hout, hopsb
        cpu_baseline = {"value": round(ns * L * L / t_all / 1e9, 4), "unit": "GCUPS", "cores": threads,
                        "kind": "port",
                        "sample": f"{ns} of the {n_pairs} pairs, C++ restatement of rust-bio 4.0.1 "
                                  "Aligner::local (oracle/), one Aligner per thread",
                        "single_thread_value": round(n1 * L * L / t_one / 1e9, 4)}

    result = {"metric": "GCUPS (SW) + FM-index queries/sec", "value": round(gcups, 3), "unit": "GCUPS",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(sw_t / args.steps * 1e3, 3), "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "int16" if L <= 192 else "int32", "data": "synthetic",
              "config": {"workload": f"{n_pairs} x {L} bp synthetic read pairs per GPU, Aligner::local "
                                     "affine-gap (-5,-1,+1,-1), score+coords+traceback ops (BASELINE configs[1])",
                         "pairs_per_gpu": n_pairs, "read_len": L, "parallelism": f"shard{world}"},
              "roofline": roofline}
    if do_cpu:
        result["host_api"] = {"value": round(host_api_gcups, 2), "unit": "GCUPS", "pairs": nh,
                              "note": "bg_align_batch with pageable host buffers in and out (PCIe-inclusive): stages of 131072 pairs through "
                                      "three pinned staging sets, upload / kernels / download / compaction overlapped"}
    del x, y, d_ops, d_out
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ FM leg
    if not args.skip_fm:
        t0 = time.perf_counter()
        g_dev = synth_gpu.genome(args.genome, seed=3, device=dev)
        g = g_dev.cpu().numpy()
        sa = suffix_array(g)
        b = bwt(g, sa)
        ls = less(b, N_ALPHABET)
        fm = FMIndex(b, ls, Occ(b, 128, N_ALPHABET), ctx=ctx)
        build_s = time.perf_counter() - t0
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
                shard.gather_records(torch.stack((d_lo, d_hi), dim=1), counts=[n_q] * world)

        fm_t = timed_steps(fm_step, args.steps, args.warmup, dev)
        qps = world * float(n_q) * args.steps / fm_t
        ctx.enable_timing(True)
        for _ in range(2):
            fm_step()
        torch.cuda.synchronize()
        tm = ctx.timing()
        ctx.enable_timing(False)
        fm_ms = tm["fm_ms"] / max(1, tm["fm_launches"])
        ml = d_ml.to(torch.int64)
        steps_exec = int((ml + (d_tag == 1).to(torch.int64) + (d_tag == 2).to(torch.int64)).sum().item())
        # algorithmic bytes per query: |P| + 24 + 128 x LF steps executed
        alg_bytes = float(n_q) * (P + 24) + 128.0 * steps_exec
        fm_ach = alg_bytes / (fm_ms * 1e-3) / 1e9
        fm_res = {"value": round(qps, 1), "unit": "queries/s", "ms_per_step": round(fm_t / args.steps * 1e3, 3),
                  "config": {"workload": f"FMIndex over {args.genome} bp synthetic genome + '$' (n_alphabet, Occ k=128), "
                                         f"{n_q} x {P} bp backward_search per GPU (BASELINE configs[2])",
                             "index_bytes": fm.device_bytes(), "index_build_s": round(build_s, 1)},
                  "tags": {"complete": int((d_tag == 0).sum().item()), "partial": int((d_tag == 1).sum().item()),
                           "absent": int((d_tag == 2).sum().item())},
                  "roofline": {"bound": "hbm", "kernel": "fm_backward_search_kernel<true>", "achieved": round(fm_ach, 2),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fm_ach / HBM_PEAK_GBS, 5),
                               "traffic": pmc_traffic("fm_backward_search_kernel<true>", "fm_queries_per_launch", n_q),
                               "launch_ms": round(fm_ms, 4), "queries_per_launch": n_q,
                               "alg_bytes_per_query": round(alg_bytes / n_q, 1)}}
        if do_cpu:
            import oracle_py as orc
            nsq = min(n_q, 400_000)
            threads = args.cpu_threads or host_cores()
            occ = orc.Occ(b, 128, N_ALPHABET)
            hp = pat[:nsq * P].cpu().numpy()
            hoff = np.arange(nsq + 1, dtype=np.uint64) * np.uint64(P)
            t0 = time.perf_counter()
            otag, olo, ohi, oml = orc.backward_search_batch(b, ls, occ, hp, hoff, threads=threads)
            t_all = time.perf_counter() - t0
            n1 = max(1, nsq // 8)
            t0 = time.perf_counter()
            orc.backward_search_batch(b, ls, occ, hp[:n1 * P], hoff[:n1 + 1], threads=1)
            t_one = time.perf_counter() - t0
            ok = bool((d_tag[:nsq].cpu().numpy() == otag).all() and
                      (d_lo[:nsq].cpu().numpy().astype(np.uint64) == olo).all() and
                      (d_hi[:nsq].cpu().numpy().astype(np.uint64) == ohi).all() and
                      (d_ml[:nsq].cpu().numpy().astype(np.uint64) == oml).all())
            parity["fm_sample_queries"] = nsq
            parity["fm_bit_exact"] = ok
            fm_res["cpu_baseline"] = {"value": round(nsq / t_all, 1), "unit": "queries/s", "cores": threads,
                                      "kind": "port",
                                      "sample": f"{nsq} of the {n_q} queries, C++ restatement of rust-bio 4.0.1 "
                                                "backward_search + Occ::get (oracle/), shared index",
                                      "single_thread_value": round(n1 / t_one, 1)}
        result["fm"] = fm_res
        del pat, off, d_tag, d_lo, d_hi, d_ml

        # -------------------------------------------------------------- seed-and-extend leg (configs[4] shape)
        if not args.skip_pipeline:
            from rust_bio_amd.pipeline import seed_and_extend
            from rust_bio_amd.suffix_array import SampledSuffixArray
            t0 = time.perf_counter()
            SampledSuffixArray(sa, g, b, 32, fmindex=fm)
            sa_s = time.perf_counter() - t0
            Rp = args.pipeline_reads
            reads, r_starts = synth_gpu.reads_from_genome(g_dev, Rp, L, seed=5 + 100003 * rank)
            al2 = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)
            holder = {}

            def pipe_step():
                holder["res"] = seed_and_extend(fm, al2, g_dev, args.genome, reads, Rp, L)

            pipe_t = timed_steps(pipe_step, max(1, args.steps // 2), 1, dev)
            res = holder["res"]
            mapped = res.score > -(1 << 29)
            near = ((res.ref_start - r_starts).abs() <= 8) & mapped
            result["seed_extend"] = {
                "value": round(world * Rp * max(1, args.steps // 2) / pipe_t, 1), "unit": "reads/s",
                "config": {"workload": f"{Rp} x {L} bp reads per GPU (5% sub, 1% ins, 1% del) vs the {args.genome} bp genome: "
                                       "20-bp seeds at stride 10 -> backward_search -> Interval::occ (sampled SA, rate 32, "
                                       "intervals <= 16 rows) -> Aligner::semiglobal on +-25 bp windows -> best hit "
                                       "(BASELINE configs[4] shape, genome scaled to the FM leg's)",
                           "sampled_sa_build_s": round(sa_s, 1)},
                "seed_hits": res.n_seed_hits, "candidates": res.n_candidates,
                "mapped_frac": round(float(mapped.float().mean().item()), 4),
                "mapped_at_origin_frac": round(float(near.float().mean().item()), 4)}
            del reads, res, holder
        del sa, g_dev
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ banded leg (configs[3] shape)
    if not args.skip_banded:
        from rust_bio_amd.banded import Aligner as BandedAligner
        Pb, Lb, kb, wb = args.banded_pairs, 10_000, 16, 32
        bx, bxo, by, byo = synth_gpu.sw_pairs_big(Pb, Lb, seed=4 + 100003 * rank, device=dev, sub=0.06, ins=0.02,
                                                   dele=0.02, chunk=64)
        hx, hy = bx.cpu().numpy(), by.cpu().numpy()
        hoff = np.arange(Pb + 1, dtype=np.uint64) * np.uint64(Lb)
        bal = BandedAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), kb, wb, ctx=ctx)
        bal.align_arrays(2, hx, hoff, hy, hoff)  # warm-up at full size: sizes the pinned staging and device scratch
        shard.barrier()
        t0 = time.perf_counter()
        bout, bops = bal.align_arrays(2, hx, hoff, hy, hoff)
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
        torch.cuda.synchronize()
        bt_dev = shard.max_over_ranks(time.perf_counter() - t0, dev)
        dev_ok = bool((d_bout.view(torch.int32).view(Pb, 16)[:, 0].cpu().numpy() == bout["score"]).all())
        del d_bout, d_bops, bx, by
        # kernel durations from a second, event-timed pass (timing serialises the K3/K4/host pipeline)
        ctx.enable_timing(True)
        bal.align_arrays(2, hx, hoff, hy, hoff)
        tm = ctx.timing()
        ctx.enable_timing(False)
        bcells = float(bal.last_cells.sum())
        # algorithmic bytes per pair: m + n + 8(n+1) + 2 x band_cells + 24 + n_ops
        balg = float(Pb) * (2 * Lb + 8 * (Lb + 1) + 24) + 2.0 * bcells + float(bout["n_ops"].sum())
        bfill_s = tm["fill_ms"] * 1e-3
        Pb_launch = int(Pb / max(1, tm["fill_launches"]))
        banded = {"value": round(world * bcells / bt / 1e9, 3), "unit": "GCUPS (band cells, host-buffer API: PCIe + band "
                  "construction on the device + K3 + K4)",
                  "pairs_per_s": round(world * Pb / bt, 1),
                  "device_resident": {"value": round(world * bcells / bt_dev / 1e9, 3), "unit": "GCUPS (band cells)",
                                      "pairs_per_s": round(world * Pb / bt_dev, 1), "scores_equal_host_api": dev_ok},
                  "config": {"workload": f"{Pb} x 10 kb read pairs per GPU (6% sub, 2% ins, 2% del), banded::Aligner::"
                                         f"semiglobal, k-mer {kb}, w {wb} (BASELINE configs[3] shape)",
                             "mean_band_cells": round(bcells / Pb, 1)},
                  "kernel_ms": {"fill": round(tm["fill_ms"], 2), "traceback": round(tm["traceback_ms"], 2)},
                  "dp_only_gcups": round(bcells / ((tm["fill_ms"] + tm["traceback_ms"]) * 1e-3) / 1e9, 2),
                  "host_threads": host_cores(),
                  "roofline": {"bound": "hbm", "kernel": "banded_fill2_kernel (+ banded_epilogue_kernel)", "achieved": round(balg / bfill_s / 1e9, 2),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(balg / bfill_s / 1e9 / HBM_PEAK_GBS, 5),
                               "traffic": pmc_traffic("banded_fill2_kernel", "banded_pairs_per_launch", Pb_launch),
                               "alg_bytes_per_pair": round(balg / Pb, 1)},
                  "pairs_per_launch": Pb_launch}
        if do_cpu:
            import oracle_py as orc
            nsb = min(Pb, max(8, (args.cpu_threads or host_cores()) // 2))
            threads = min(nsb, args.cpu_threads or host_cores())
            osc = orc.make_scoring(-5, -1, 1, -1)
            t0 = time.perf_counter()
            oout, oops, ostride, ocells = orc.banded_align_batch(osc, "semiglobal", kb, wb, hx[:nsb * Lb], hoff[:nsb + 1],
                                                                 hy[:nsb * Lb], hoff[:nsb + 1], threads=threads)
            t_all = time.perf_counter() - t0
            okb = bool((bout["score"][:nsb] == oout["score"]).all() and (bout["n_ops"][:nsb] == oout["n_ops"]).all() and
                       (bal.last_cells[:nsb] == ocells).all())
            kind = (oops.reshape(nsb, ostride) & 0xFF).astype(np.uint8)
            for p in range(nsb):
                kq, oq = int(bout["n_ops"][p]), int(bout["ops_off"][p])
                okb = okb and bool((bops[oq:oq + kq] == kind[p, :kq]).all())
            parity["banded_sample_pairs"] = nsb
            parity["banded_bit_exact"] = okb
            banded["cpu_baseline"] = {"value": round(float(ocells.sum()) / t_all / 1e9, 4), "unit": "GCUPS (band cells)",
                                      "cores": threads, "kind": "port",
                                      "sample": f"{nsb} of the {Pb} pairs, C++ restatement of rust-bio 4.0.1 "
                                                "banded::Aligner::semiglobal incl. band construction (oracle/)"}
        result["banded"] = banded

    # ------------------------------------------------------------------ ingest leg
    if not args.skip_ingest:
        from rust_bio_amd import fastq as bgfastq
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
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                               "note": "algorithmic bytes: text once + sequences, qualities, records and offsets written"}}
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
        result["ingest"] = ingest

    if rank == 0:
        if cpu_baseline is not None:
            result["cpu_baseline"] = cpu_baseline
        if parity is not None:
            result["parity"] = parity
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
