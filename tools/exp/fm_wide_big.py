"""An FM index over a text of MORE THAN 2^32 SYMBOLS on one MI355X ("a synthetic 4.4 G-symbol text indexes on
the 288 GB part, a seeded sample of intervals and positions equals the oracle's"):
    python tools/exp/fm_wide_big.py [symbols=4400000000] [queries=10000000] > gpurun_out/r05_fm_wide_4g4.json
  1. the text in HBM: random ACGT (SplitMix64, synth_gpu.genome) with a 2 Mbp segment copied three times further on (real work
     for the prefix doubling: groups that stay tied for 2 M symbols) and `n_N` N (argv[3], default 2: the 2-step rank blocks
     take a text with at most four positions outside its four letters; 100 reproduces round 5's single-step text), '$' at the end;
  2. bg_suffix_array_dev64 -> bg_bwt_dev64 -> bg_sa_sample_dev64 (rate 32) -> bg_fm_build_dev (the 64-bit layout, fm_wide.hip)
     -> bg_fm_set_sampled_suffix_array: nothing text-sized leaves the device except the samples;
  3. `queries` 100 bp patterns (half cut from the text — some from the copied segment: several occurrences — half random)
     through bg_fm_backward_search_batch_dev: the rate of the 64-bit kernel;
  4. parity, three ways:
     a. ORACLE BY DEFINITION on a STRATIFIED seeded sample — round 6: >= 1 000 Complete and >= 1 000 Partial results whose
        interval starts at or beyond row 2^32 (patterns that begin with T), 500 of each from anywhere, the copied segment's
        multi-row intervals, and every Complete pattern's position list — (oracle/fm.cpp: orc_intervals_by_scan — lower = #suffixes < P, upper = lower +
        #suffixes with prefix P, and the occurrence positions, by one pass over the 4.4 GB text on the host threads; no suffix
        array on the host): Complete intervals equal, Partial results equal the interval of the matched suffix and the suffix one
        symbol longer does not occur, located positions (K6 on 64-bit samples) equal the scan's;
     b. every position K6 returns for 200 000 Complete queries is an occurrence (the text is compared on the device) and a
        query cut from position p finds p;
     c. positions and interval bounds beyond 2^32 occur (counted), i.e. the run exercises what uint32 could not hold."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import oracle_py as orc  # noqa: E402
from csrc_hash import csrc_sha  # noqa: E402
from rust_bio_amd import _lib, synth_gpu  # noqa: E402
from rust_bio_amd.fmindex import FMIndex  # noqa: E402
from rust_bio_amd.suffix_array import bwt_dev, sample_dev, suffix_array_dev  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_400_000_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
N_N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
P = 100
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
if os.environ.get("BG_FORCE_WIDE"):  # dry run on a small text: the same code paths (64-bit layout, several passes of round 0)
    ctx.set_option("fm_wide_from", 1)
    ctx.set_option("fm_wide_sb_shift", 6)
    ctx.set_option("sa_chunk_symbols", max(1 << 16, N // 3))
threads = len(os.sched_getaffinity(0))
try:  # cgroup CPU quota: threads beyond it are only throttled (run A: 256 "cores" on a 16-CPU quota made the oracle 2.5 x slower)
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if quota != "max":
        threads = min(threads, max(1, -(-int(quota) // int(period))))
except (OSError, ValueError):
    pass
res = {"symbols": N + 1, "beyond_2_32": N + 1 > 1 << 32, "csrc_sha": csrc_sha(ROOT)}


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


# ---- 1. text
t0 = sync()
g = synth_gpu.genome(N, 91, dev)
SEG = min(2_000_000, N // 16)
for k in range(1, 4):  # the same segment four times in all
    dst = (N // 5) * k + 12345
    g[dst:dst + SEG] = g[777:777 + SEG]
npos = torch.arange(N_N, dtype=torch.int64, device=dev) * (N // (N_N + 1)) + 31
g[npos] = ord("N")
res["text_s"] = round(sync() - t0, 2)
torch.cuda.empty_cache()
free0, total = torch.cuda.mem_get_info()
res["hbm_total_gb"] = round(total / 1e9, 1)

# ---- 2. suffix array, BWT, samples, index
t0 = sync()
d_sa = suffix_array_dev(g, ctx=ctx, wide=True)
res["suffix_array_s"] = round(sync() - t0, 2)
t0 = sync()
d_b = bwt_dev(g, d_sa, ctx=ctx)
samples = sample_dev(d_sa, d_b, ord("$"), 32, ctx=ctx)
res["bwt_and_samples_s"] = round(sync() - t0, 2)
t0 = sync()
fm = FMIndex.from_device(d_b, 128, b"ACGTNacgtn", ctx=ctx)
samples.attach(fm)
res["index_s"] = round(sync() - t0, 2)
res["index_bytes"] = fm.device_bytes()
res["step2_bytes"] = fm.step2_bytes()
res["less"] = {chr(c): int(fm._less[c]) for c in b"$ACGNT"}
# the suffix array itself is not needed any more (35 GB): K6 locates through the samples
sa_probe = d_sa[torch.tensor([0, 1, N // 2, N], device=dev)].cpu().tolist()
res["sa_probe_rows_0_1_mid_last"] = sa_probe
del d_sa
torch.cuda.empty_cache()

# ---- 3. queries
rng = np.random.default_rng(5)
z = synth_gpu.splitmix64(1234, NQ, dev)
start = torch.remainder(z & ((1 << 62) - 1), N - P)  # (non-negative 62-bit draws)
in_seg = torch.arange(NQ, device=dev) % 50 == 0       # 2 %: inside the copied segment (four occurrences at least)
start = torch.where(in_seg, 777 + torch.remainder(start, SEG - P), start)
is_rand = torch.arange(NQ, device=dev) % 2 == 1
ar = torch.arange(P, dtype=torch.int64, device=dev)
pat = torch.empty((NQ, P), dtype=torch.uint8, device=dev)
CH = 1 << 20
for s in range(0, NQ, CH):
    e = min(NQ, s + CH)
    pat[s:e] = g[start[s:e].view(-1, 1) + ar.view(1, -1)]
for s in range(0, NQ, CH):  # every second pattern: random symbols instead (most of them end Partial after ~16 steps)
    e = min(NQ, s + CH)
    rnd = synth_gpu.random_dna((e - s) * P, 77, dev, start=s * P).view(e - s, P)
    m = is_rand[s:e]
    pat[s:e][m] = rnd[m]
    del rnd
pat = pat.view(-1)
off = torch.arange(NQ + 1, dtype=torch.int64, device=dev) * P
d_tag = torch.empty(NQ, dtype=torch.uint8, device=dev)
d_lo = torch.empty(NQ, dtype=torch.int64, device=dev)
d_hi = torch.empty(NQ, dtype=torch.int64, device=dev)
d_ml = torch.empty(NQ, dtype=torch.int32, device=dev)


def search():
    fm.backward_search_dev(NQ, pat.data_ptr(), off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), d_ml.data_ptr())


def timed(reps=3):
    search()
    t0 = sync()
    for _ in range(reps):
        search()
    return (sync() - t0) / reps


def snapshot():
    return d_tag.clone(), d_lo.clone(), d_hi.clone(), d_ml.clone()


def same_as(ref):
    ok = d_tag < 2
    return bool(torch.equal(ref[0], d_tag) and torch.equal(ref[3], d_ml) and torch.equal(ref[1][ok], d_lo[ok]) and torch.equal(ref[2][ok], d_hi[ok]))


# every kernel of the 64-bit search on the same arrays: one query per quad on single steps (round 5's first kernel), two per
# quad on single steps (round 5's default), and — where the text has 2-step blocks — the 2x fast kernel on them (round 6),
# with byte patterns and with the patterns packed to 2 bits
rates = {}
fm.set_option("ilp", 1)
rates["single_steps_one_query_per_quad"] = timed()
ref = snapshot()
fm.set_option("ilp", 2)
fm.set_option("no_step2", 1)
rates["single_steps_two_queries_per_quad"] = timed()
equal = {"single_steps_two_queries_per_quad": same_as(ref)}
fm.set_option("no_step2", 0)
kernel = "fmw_search2x_kernel (64-bit positions, 1-step blocks, byte patterns, two queries per quad)"
if fm.step2_bytes():
    rates["two_step_blocks_byte_patterns"] = timed()
    equal["two_step_blocks_byte_patterns"] = same_as(ref)
    kernel = "fm_search_fast2x_kernel<WIDE> (64-bit positions, 2-step blocks, byte patterns, two queries per quad) + the generic kernel for deferred queries"
    from rust_bio_amd import pack2
    pk, bad = pack2.pack_dev(pat, codes=fm.pattern_codes(), ctx=ctx)
    if bad == 0:
        byte_search = search

        def search():  # noqa: F811
            fm.backward_search_packed_dev(NQ, pk.data_ptr(), off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), d_ml.data_ptr())

        rates["two_step_blocks_packed_patterns"] = timed()
        equal["two_step_blocks_packed_patterns"] = same_as(ref)
        search = byte_search
    del pk
search()
torch.cuda.synchronize()
del ref
dt = rates.get("two_step_blocks_byte_patterns", rates["single_steps_two_queries_per_quad"])
res["search"] = {"queries": NQ, "pattern_len": P, "ms": round(dt * 1e3, 2), "queries_per_s": round(NQ / dt, 1),
                 "complete": int((d_tag == 0).sum()), "partial": int((d_tag == 1).sum()), "absent": int((d_tag == 2).sum()),
                 "kernel": kernel,
                 "all_kernels_M_queries_per_s": {k: round(NQ / v / 1e6, 1) for k, v in rates.items()},
                 "results_equal_to_the_one_query_kernel": equal}
res["intervals_with_a_bound_beyond_2_32"] = int(((d_hi > (1 << 32)) & (d_tag < 2)).sum())

# ---- 4b. located positions are occurrences; a query cut from p finds p
NL = min(NQ // 2, 200_000)
sel = torch.nonzero((d_tag == 0) & ~is_rand)[:NL].view(-1)
lo_s, hi_s = d_lo[sel], d_hi[sel]
cnt = (hi_s - lo_s).clamp(max=64)  # (the copied segment's patterns have four rows; cap anything odd)
out_off = torch.zeros(sel.numel() + 1, dtype=torch.int64, device=dev)
out_off[1:] = torch.cumsum(cnt, 0)
total = int(out_off[-1])
d_pos = torch.empty(total, dtype=torch.int64, device=dev)
t0 = sync()
fm.interval_occ_dev(sel.numel(), lo_s.data_ptr(), out_off.data_ptr(), total, d_pos.data_ptr())
res["locate"] = {"intervals": int(sel.numel()), "positions": total, "ms": round((sync() - t0) * 1e3, 2)}
owner = torch.repeat_interleave(torch.arange(sel.numel(), device=dev), cnt)
ok_occ = True
found_self = torch.zeros(sel.numel(), dtype=torch.bool, device=dev)
for s in range(0, total, CH):
    e = min(total, s + CH)
    q = sel[owner[s:e]]
    got = g[d_pos[s:e].view(-1, 1) + ar.view(1, -1)]
    ok_occ = ok_occ and bool((got == pat.view(NQ, P)[q]).all())
    hit = d_pos[s:e] == start[q]
    found_self[owner[s:e][hit]] = True
res["locate"]["every_position_is_an_occurrence"] = ok_occ
res["locate"]["every_query_finds_where_it_was_cut"] = bool(found_self[(hi_s - lo_s) <= 64].all())
res["locate"]["positions_beyond_2_32"] = int((d_pos >= (1 << 32)).sum())

# ---- 4d. seed -> locate -> extend on 64-bit positions (round 6): reads cut from the text — half of them beyond position 2^32 — with
# 2 % substitutions must come home (bg_seed_extend_batch_dev: SEEDS flavour of the 64-bit search, K6 on 64-bit samples, proposals
# sorted as uint64, windows cut at 64-bit offsets); small-scale parity of the same call against oracle/pipeline.cpp is
# tests/test_gpu_fm_wide.py::test_wide_seed_and_extend_equals_the_pipeline_oracle
from rust_bio_amd.pairwise import Scoring  # noqa: E402
from rust_bio_amd.pipeline import attach_text, seed_extend_dev  # noqa: E402
attach_text(fm, d_text=g)
RN, RL = 200_000, 150
zr = synth_gpu.splitmix64(555, RN, dev)
lo_half = torch.remainder(zr & ((1 << 62) - 1), min(N - RL - 1, (1 << 32) - RL - 1))
hi_half = (1 << 32) + torch.remainder(zr & ((1 << 62) - 1), max(1, N - RL - 1 - (1 << 32))) if N > (1 << 32) + RL + 1 else lo_half
r_start = torch.where(torch.arange(RN, device=dev) % 2 == 0, lo_half, hi_half)
arr = torch.arange(RL, dtype=torch.int64, device=dev)
rd = g[r_start.view(-1, 1) + arr.view(1, -1)].clone()
ur = synth_gpu.splitmix64(556, RN * RL, dev).view(RN, RL)
acgt_t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
rd = torch.where((ur & 0xFFFF) < int(0.02 * 65536), acgt_t[(ur >> 20) & 3], rd).contiguous()
clean = ((rd != ord("N")) & (rd != ord("$"))).all(dim=1)
r_off = torch.arange(RN + 1, dtype=torch.int64, device=dev) * RL
d_hits = torch.zeros(RN * 96, dtype=torch.uint8, device=dev)
sc = Scoring.from_scores(-5, -1, 1, -1)
seed_extend_dev(fm, sc, RN, rd.data_ptr(), r_off.data_ptr(), RL, d_hits.data_ptr())
t0 = sync()
seed_extend_dev(fm, sc, RN, rd.data_ptr(), r_off.data_ptr(), RL, d_hits.data_ptr())
dt_se = sync() - t0
h64 = d_hits.view(torch.int64).view(RN, 12)
score = (h64[:, 0] << 32) >> 32  # (sign-extended int32 score in the low word)
ref_start = h64[:, 9]
mapped = score > -858993459
near = (ref_start - r_start).abs() <= 8
# (a read from the copied segment comes home to the smallest of its identical occurrences: the text there must read the same)
rs = torch.where(mapped, ref_start, r_start).clamp(0, N - RL - 1)
same_text = torch.zeros(RN, dtype=torch.bool, device=dev)
for s0 in range(0, RN, 1 << 16):
    e0 = min(RN, s0 + (1 << 16))
    same_text[s0:e0] = (g[rs[s0:e0].view(-1, 1) + arr.view(1, -1)] == g[r_start[s0:e0].view(-1, 1) + arr.view(1, -1)]).all(dim=1)
home = mapped & (near | same_text)
res["seed_extend"] = {"reads": RN, "read_len": RL, "ms": round(dt_se * 1e3, 2), "reads_per_s": round(RN / dt_se, 1),
                      "mapped": int(mapped.sum()), "mapped_within_8_of_where_they_were_cut": int(home.sum()),
                      "reads_cut_beyond_2_32": int((r_start >= (1 << 32)).sum()),
                      "of_them_home": int((home & (r_start >= (1 << 32))).sum()),
                      "clean_reads_not_home": int((clean & ~home).sum())}
del rd, d_hits, ur

# ---- 4a. the oracle by definition on a stratified seeded sample
BEY = 1 << 32
NS_BEY, NS_ANY = 1000, 500


def first(mask, k):
    return torch.nonzero(mask)[:k].view(-1)


pick = torch.cat([first((d_tag == 0) & ~is_rand & (d_lo >= BEY), NS_BEY), first((d_tag == 0) & ~is_rand & (d_lo < BEY), NS_ANY),
                  first((d_tag == 0) & in_seg & ~is_rand, 64),
                  first((d_tag == 1) & (d_lo >= BEY), NS_BEY), first((d_tag == 1) & (d_lo < BEY), NS_ANY)])
pick = torch.unique(pick)
h_pat = pat.view(NQ, P)[pick].cpu().numpy()
h_tag, h_lo, h_hi, h_ml = d_tag[pick].cpu().numpy(), d_lo[pick].cpu().numpy().astype(np.uint64), d_hi[pick].cpu().numpy().astype(np.uint64), d_ml[pick].cpu().numpy()
scan_pats, kinds = [], []
for k in range(len(pick)):
    if h_tag[k] == 0:
        scan_pats.append(h_pat[k].tobytes()); kinds.append(("complete", k))
    else:  # Partial(pl, pr + 1, ml): the interval of the last ml symbols; one symbol more does not occur (fmindex.rs:160-182)
        ml = int(h_ml[k])
        scan_pats.append(h_pat[k][P - ml:].tobytes()); kinds.append(("partial", k))
        scan_pats.append(h_pat[k][P - ml - 1:].tobytes()); kinds.append(("longer", k))
buf = np.frombuffer(b"".join(scan_pats), dtype=np.uint8)
soff = np.zeros(len(scan_pats) + 1, dtype=np.uint64)
soff[1:] = np.cumsum([len(p) for p in scan_pats])
t0 = time.perf_counter()
h_text = g.cpu().numpy()
res["text_download_s"] = round(time.perf_counter() - t0, 2)
t0 = time.perf_counter()
slo, shi, spos = orc.intervals_by_scan(h_text, buf, soff, pos_cap=64, threads=threads)
res["oracle_scan_s"] = round(time.perf_counter() - t0, 1)
bad = 0
n_c = n_p = 0
loc_lo, loc_hi, loc_want = [], [], []
for j, (kind, k) in enumerate(kinds):
    if kind == "complete":
        n_c += 1
        bad += (int(slo[j]), int(shi[j])) != (int(h_lo[k]), int(h_hi[k]))
        loc_lo.append(int(h_lo[k])); loc_hi.append(int(h_hi[k])); loc_want.append(sorted(int(v) for v in spos[j]))
    elif kind == "partial":
        n_p += 1
        bad += (int(slo[j]), int(shi[j])) != (int(h_lo[k]), int(h_hi[k]))
    else:
        bad += int(shi[j]) != int(slo[j])
_, pos = fm.interval_occ_arrays(np.array(loc_lo, dtype=np.uint64), np.array(loc_hi, dtype=np.uint64))
o = 0
bad_pos = 0
for lo_, hi_, want in zip(loc_lo, loc_hi, loc_want):
    got = sorted(int(v) for v in pos[o:o + hi_ - lo_])
    o += hi_ - lo_
    bad_pos += got != want
res["oracle_sample"] = {"patterns": len(pick), "complete": n_c, "partial": n_p, "scans": len(scan_pats), "oracle_threads": threads,
                        "interval_mismatches": int(bad), "position_list_mismatches": int(bad_pos),
                        "max_interval_size": int(max(h - l for l, h in zip(loc_lo, loc_hi))),
                        "intervals_beyond_2_32": int(sum(1 for l in loc_lo if l >= (1 << 32))),
                        "complete_intervals_beyond_2_32": int(sum(1 for j, (kind, k) in enumerate(kinds) if kind == "complete" and int(h_lo[k]) >= BEY)),
                        "partial_intervals_beyond_2_32": int(sum(1 for j, (kind, k) in enumerate(kinds) if kind == "partial" and int(h_lo[k]) >= BEY)),
                        "position_lists_compared": len(loc_lo),
                        "position_lists_of_intervals_beyond_2_32": int(sum(1 for l in loc_lo if l >= BEY)),
                        "positions_beyond_2_32_in_those_lists": int(sum(1 for w in loc_want for v in w if v >= BEY))}
res["bit_exact"] = bool(bad == 0 and bad_pos == 0 and ok_occ and res["locate"]["every_query_finds_where_it_was_cut"])
res["seed_extend"]["ok"] = bool(res["seed_extend"]["mapped_within_8_of_where_they_were_cut"] >= 0.97 * RN)
res["hbm_free_before_the_suffix_array_gb"] = round(free0 / 1e9, 1)
print(json.dumps(res))
