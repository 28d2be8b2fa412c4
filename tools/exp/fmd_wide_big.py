"""An FMD index over T$R$ of a human-genome-sized text — 6.2 G symbols, the case the 64-bit layout exists for (VERDICT r5 item 1:
"a 6.2 G-symbol T$R$ index built in HBM, SMEMs of >= 10 k sampled reads equal to an oracle"):
    python tools/exp/fmd_wide_big.py [genome_bp=3100000000] [reads=12000] > gpurun_out/<tag>/fmd_wide_big.json
  1. the text in HBM: T = random ACGT (SplitMix64) with a 1 Mbp segment copied twice further on and one N (argv[3]), then '$',
     revcomp(T), '$' (FMDIndex::from's input, /root/reference/src/data_structures/fmindex.rs:311-340): 2 |T| + 2 symbols;
  2. bg_suffix_array_dev64 (round 0 in bucket-range passes: sa_build.hip) -> bg_bwt_dev64 -> bg_sa_sample_dev64 ->
     bg_fm_build_dev (64-bit layout + 2-step blocks) — everything stays on the device;
  3. `reads` 150 bp reads cut from T (3 % substitutions, half of them reverse-complemented, some from the copied segment)
     through bg_fmd_smems_batch64_dev (FMDIndex::all_smems, fmindex.rs:479-501, min length 20): the rate of K7 on 64-bit positions;
  4. parity, two independent ways:
     a. THE ALGORITHM: oracle/fm.cpp's line-by-line restatement of FMDIndex (fmindex.rs:363-564) run on the host over the BWT
        read back from the index (bg_fm_bwt_dev) with the oracle's own Occ / less: every read's list of
        (BiInterval{lower, lower_rev, size, match_size}, position, length) must be identical;
     b. THE DEFINITION, without the BWT (so that (a) does not rest on the device's suffix array): for every distinct SMEM
        string P of the sample, one scan of the 6.2 GB text on the host threads (orc_intervals_by_scan) gives
        lower = #suffixes < P, size = #suffixes with prefix P, lower_rev = #suffixes < revcomp(P) — compared with the record —
        and says that P extended by the read's next symbol on either side does not occur (the match is maximal on both sides);
     c. backward_search of 2 M x 100 bp patterns on the same index (the 2x fast kernel on 2-step blocks, 64-bit positions): rate,
        and a sample of its intervals against the same scan."""
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

G = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 12_000
L, MINLEN = 150, 20
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
if os.environ.get("BG_FORCE_WIDE"):  # dry run on a small text: the same code paths (64-bit layout, several passes of round 0)
    ctx.set_option("fm_wide_from", 1)
    ctx.set_option("fm_wide_sb_shift", 6)
    ctx.set_option("sa_chunk_symbols", max(1 << 16, (2 * G) // 5))
threads = len(os.sched_getaffinity(0))
try:
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if quota != "max":
        threads = min(threads, max(1, -(-int(quota) // int(period))))
except (OSError, ValueError):
    pass
N = 2 * G + 2
res = {"genome_bp": G, "symbols": N, "beyond_2_32": N > 1 << 32, "csrc_sha": csrc_sha(ROOT), "host_threads": threads}
ALPHA = b"ACGTNacgtn"


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


# ---- 1. T$R$
t0 = sync()
g = torch.empty(N, dtype=torch.uint8, device=dev)
step = 1 << 26
for s in range(0, G, step):
    e = min(G, s + step)
    g[s:e] = synth_gpu.random_dna(e - s, 92, dev, start=s)
SEG = min(1_000_000, G // 16)
for k in (1, 2):
    dst = (G // 3) * k + 4321
    g[dst:dst + SEG] = g[999:999 + SEG]
N_N = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # (every N of T is one of R too; with the two sentinels at most four listed
for k in range(N_N):                                  #  positions keep the 2-step rank blocks: one N by default, more = single steps)
    g[(G // (N_N + 1)) * (k + 1) + 5] = ord("N")
g[G] = ord("$")
comp = torch.zeros(256, dtype=torch.uint8, device=dev)
for a, b in zip(b"ACGTN", b"TGCAN"):
    comp[a] = b
for s in range(0, G, step):  # R[j] = comp(T[G - 1 - j])
    e = min(G, s + step)
    g[G + 1 + s:G + 1 + e] = comp[g[G - e:G - s].flip(0).to(torch.int64)]
g[N - 1] = ord("$")
res["text_s"] = round(sync() - t0, 2)
torch.cuda.empty_cache()
free0, total = torch.cuda.mem_get_info()
res["hbm_total_gb"], res["hbm_free_before_the_suffix_array_gb"] = round(total / 1e9, 1), round(free0 / 1e9, 1)

# ---- 2. suffix array, BWT, samples, index
t0 = sync()
d_sa = suffix_array_dev(g, ctx=ctx, wide=True)
res["suffix_array_s"] = round(sync() - t0, 2)
t0 = sync()
d_b = bwt_dev(g, d_sa, ctx=ctx)
samples = sample_dev(d_sa, d_b, ord("$"), 32, ctx=ctx)
res["bwt_and_samples_s"] = round(sync() - t0, 2)
res["sa_probe_rows_0_1_2_last"] = d_sa[torch.tensor([0, 1, 2, N - 1], device=dev)].cpu().tolist()  # rows 0, 1: the sentinels, last first
del d_sa
torch.cuda.empty_cache()
t0 = sync()
fm = FMIndex.from_device(d_b, 128, ALPHA, ctx=ctx)
samples.attach(fm)
res["index_s"] = round(sync() - t0, 2)
res["index_bytes"], res["step2_bytes"] = fm.device_bytes(), fm.step2_bytes()
res["less"] = {chr(c): int(fm._less[c]) for c in b"$ACGNT"}
del d_b
torch.cuda.empty_cache()

# ---- 3. reads and their SMEMs
z = synth_gpu.splitmix64(4321, NR, dev)
start = torch.remainder(z & ((1 << 62) - 1), G - L)
in_seg = torch.arange(NR, device=dev) % 25 == 0
start = torch.where(in_seg, 999 + torch.remainder(start, SEG - L), start)
ar = torch.arange(L, dtype=torch.int64, device=dev)
reads = g[start.view(-1, 1) + ar.view(1, -1)].clone()
u = synth_gpu.splitmix64(99, NR * L, dev).view(NR, L)
sub = (u & 0xFFFF) < int(0.03 * 65536)
acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
reads = torch.where(sub, acgt[((u >> 20) & 3)], reads)
rc = torch.arange(NR, device=dev) % 2 == 1
reads[rc] = comp[reads[rc].flip(1).to(torch.int64)]
reads = reads.contiguous()
r_off = torch.arange(NR + 1, dtype=torch.int64, device=dev) * L
CAP = 32
d_cnt = torch.zeros(NR, dtype=torch.int32, device=dev)
d_out = torch.zeros((NR, CAP, 6), dtype=torch.int64, device=dev)


def smems():
    _lib.check(_lib.lib().bg_fmd_smems_batch64_dev(fm.h, 1, NR, reads.data_ptr(), r_off.data_ptr(), None, MINLEN, L, CAP, d_cnt.data_ptr(),
                                                   d_out.data_ptr(), None), "bg_fmd_smems_batch64_dev")


smems()
t0 = sync()
for _ in range(3):
    smems()
dt = (sync() - t0) / 3
cnt = d_cnt.cpu().numpy().astype(np.int64)
out = d_out.cpu().numpy().astype(np.uint64)
assert cnt.max() <= CAP and cnt.min() >= 0
n_smem = int(cnt.sum())
BEY = 1 << 32
recs = np.concatenate([out[q, :cnt[q]] for q in range(NR)]) if n_smem else np.zeros((0, 6), dtype=np.uint64)
res["smems"] = {"reads": NR, "read_len": L, "min_len": MINLEN, "ms": round(dt * 1e3, 2), "reads_per_s": round(NR / dt, 1), "smems": n_smem,
                "kernel": "fmd_smems_kernel<WIDE, OUT64> (K7 on 64-bit positions, uint64 records)",
                "records_with_lower_beyond_2_32": int((recs[:, 0] >= BEY).sum()), "records_with_lower_rev_beyond_2_32": int((recs[:, 1] >= BEY).sum()),
                "records_with_size_above_1": int((recs[:, 2] > 1).sum()), "longest": int(recs[:, 5].max()) if n_smem else 0}

# ---- 4a. the algorithm: the oracle's FMDIndex over the BWT read back from the index
t0 = time.perf_counter()
d_bw = torch.empty(N, dtype=torch.uint8, device=dev)
_lib.check(_lib.lib().bg_fm_bwt_dev(fm.h, d_bw.data_ptr(), None), "bg_fm_bwt_dev")
torch.cuda.synchronize()
h_bwt = d_bw.cpu().numpy()
del d_bw
h_less = orc.less(h_bwt, ALPHA)
assert (np.asarray(h_less, dtype=np.uint64) == fm._less).all()
occ = orc.Occ(h_bwt, 128, ALPHA)
ofmd = orc.FMDIndex(h_bwt, h_less, occ)
res["oracle_tables_s"] = round(time.perf_counter() - t0, 1)
h_reads = reads.cpu().numpy()
t0 = time.perf_counter()
bad_alg = 0
first_bad = None
for q in range(NR):
    want = ofmd.all_smems(h_reads[q].tobytes(), MINLEN)
    got = [((int(r[0]), int(r[1]), int(r[2]), int(r[3])), int(r[4]), int(r[5])) for r in out[q, :cnt[q]]]
    if got != want:
        bad_alg += 1
        if first_bad is None:
            first_bad = {"read": q, "got": got[:3], "want": want[:3]}
res["oracle_algorithm"] = {"reads_compared": NR, "reads_with_a_difference": bad_alg, "first": first_bad, "s": round(time.perf_counter() - t0, 1),
                           "what": "oracle/fm.cpp FMDIndex::all_smems (fmindex.rs:363-564 restated) over the BWT read back from the handle"}
del ofmd, occ, h_bwt

# ---- 4b. the definition: one scan of the text for every distinct SMEM string, its reverse complement and its two extensions
h_text = g.cpu().numpy()
ctab = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    ctab[a] = b
pats, what = [], []
seen = {}
for q in range(NR):
    rd = h_reads[q]
    for r in out[q, :cnt[q]]:
        p0, ln = int(r[4]), int(r[5])
        P_ = rd[p0:p0 + ln].tobytes()
        key = (q, p0, ln)
        pats.append(P_); what.append(("fwd", key, r))
        pats.append(ctab[rd[p0:p0 + ln]][::-1].tobytes()); what.append(("rev", key, r))
        if p0 > 0:
            pats.append(rd[p0 - 1:p0 + ln].tobytes()); what.append(("left", key, r))
        if p0 + ln < L:
            pats.append(rd[p0:p0 + ln + 1].tobytes()); what.append(("right", key, r))
buf = np.frombuffer(b"".join(pats), dtype=np.uint8)
soff = np.zeros(len(pats) + 1, dtype=np.uint64)
soff[1:] = np.cumsum([len(p) for p in pats])
t0 = time.perf_counter()
slo, shi, _ = orc.intervals_by_scan(h_text, buf, soff, pos_cap=1, threads=threads)
res["oracle_scan_s"] = round(time.perf_counter() - t0, 1)
bad_def = {"lower": 0, "size": 0, "lower_rev": 0, "extends_left": 0, "extends_right": 0}
for j, (kind, key, r) in enumerate(what):
    lo_, sz_ = int(slo[j]), int(shi[j]) - int(slo[j])
    if kind == "fwd":
        bad_def["lower"] += lo_ != int(r[0])
        bad_def["size"] += sz_ != int(r[2])
    elif kind == "rev":
        bad_def["lower_rev"] += lo_ != int(r[1])
        bad_def["size"] += sz_ != int(r[2])
    elif kind == "left":
        bad_def["extends_left"] += sz_ != 0
    else:
        bad_def["extends_right"] += sz_ != 0
res["oracle_definition"] = {"smems_checked": n_smem, "patterns_scanned": len(pats), "mismatches": bad_def,
                            "what": "orc_intervals_by_scan over the 6.2 GB text: lower / size / lower_rev of every SMEM by definition, and "
                                    "that neither one-symbol extension inside the read occurs"}

# ---- 4c. backward_search on the same index
NQ, P = 2_000_000, 100
zq = synth_gpu.splitmix64(777, NQ, dev)
qs = torch.remainder(zq & ((1 << 62) - 1), N - P - 2)
arp = torch.arange(P, dtype=torch.int64, device=dev)
pat = torch.empty((NQ, P), dtype=torch.uint8, device=dev)
for s in range(0, NQ, 1 << 20):
    e = min(NQ, s + (1 << 20))
    pat[s:e] = g[qs[s:e].view(-1, 1) + arp.view(1, -1)]
is_rand = torch.arange(NQ, device=dev) % 2 == 1
rnd = synth_gpu.random_dna(NQ * P, 78, dev).view(NQ, P)
pat[is_rand] = rnd[is_rand]
del rnd
pat = pat.view(-1)
poff = torch.arange(NQ + 1, dtype=torch.int64, device=dev) * P
d_tag = torch.empty(NQ, dtype=torch.uint8, device=dev)
d_lo = torch.empty(NQ, dtype=torch.int64, device=dev)
d_hi = torch.empty(NQ, dtype=torch.int64, device=dev)
d_ml = torch.empty(NQ, dtype=torch.int32, device=dev)


def search():
    fm.backward_search_dev(NQ, pat.data_ptr(), poff.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), d_ml.data_ptr())


search()
t0 = sync()
for _ in range(3):
    search()
dts = (sync() - t0) / 3
pick = torch.cat([torch.nonzero((d_tag == 0) & (d_lo >= BEY))[:300].view(-1), torch.nonzero((d_tag == 0) & (d_lo < BEY))[:100].view(-1),
                  torch.nonzero((d_tag == 1) & (d_lo >= BEY))[:300].view(-1), torch.nonzero((d_tag == 1) & (d_lo < BEY))[:100].view(-1)])
h_pat = pat.view(NQ, P)[pick].cpu().numpy()
h_tag, h_lo, h_hi, h_ml = d_tag[pick].cpu().numpy(), d_lo[pick].cpu().numpy(), d_hi[pick].cpu().numpy(), d_ml[pick].cpu().numpy()
sp, kinds = [], []
for k in range(len(pick)):
    if h_tag[k] == 0:
        sp.append(h_pat[k].tobytes()); kinds.append(("complete", k))
    else:
        ml = int(h_ml[k])
        sp.append(h_pat[k][P - ml:].tobytes()); kinds.append(("partial", k))
        sp.append(h_pat[k][P - ml - 1:].tobytes()); kinds.append(("longer", k))
buf = np.frombuffer(b"".join(sp), dtype=np.uint8)
soff = np.zeros(len(sp) + 1, dtype=np.uint64)
soff[1:] = np.cumsum([len(p) for p in sp])
slo, shi, _ = orc.intervals_by_scan(h_text, buf, soff, pos_cap=1, threads=threads)
bad = 0
for j, (kind, k) in enumerate(kinds):
    if kind == "longer":
        bad += int(shi[j]) != int(slo[j])
    else:
        bad += (int(slo[j]), int(shi[j])) != (int(h_lo[k]), int(h_hi[k]))
res["search"] = {"queries": NQ, "pattern_len": P, "ms": round(dts * 1e3, 2), "queries_per_s": round(NQ / dts, 1),
                 "complete": int((d_tag == 0).sum()), "partial": int((d_tag == 1).sum()), "absent": int((d_tag == 2).sum()),
                 "oracle_sample": {"patterns": int(len(pick)), "scans": len(sp), "interval_mismatches": int(bad),
                                   "intervals_beyond_2_32": int((h_lo >= BEY).sum())}}
res["bit_exact"] = bool(bad_alg == 0 and sum(bad_def.values()) == 0 and bad == 0)
print(json.dumps(res))
