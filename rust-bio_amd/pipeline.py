"""Seed-and-extend read mapping composed from the engine's batched calls (BASELINE configs[4]):
FM-index seeds (`backward_search`) -> `Interval::occ` -> `Aligner::semiglobal` on the candidate
windows -> best hit per read.  The reference has no such function; callers compose it from the
same three calls (/root/reference/src/lib.rs:129-165, benches/fmindex.rs:20-38).  Everything
between the calls is index arithmetic on device tensors (torch), nothing leaves HBM.

Definition used here (tests/test_gpu_pipeline.py restates it on the CPU):
  * seeds: the windows read[o : o + seed_len] for o = 0, stride, 2*stride, ... (whole windows only);
  * a seed votes when its search is Complete and its interval holds at most `max_occ` rows;
  * each hit position p of a seed at offset o proposes the read start s = p - o; proposals that
    would start before the text are dropped, duplicates (same read, same s) are merged;
  * the candidate window is text[max(0, s - pad) : min(n_text, s + read_len + pad)], n_text being
    the text without its final sentinel; the read (x) is aligned semiglobally against it (y);
  * per read the candidate with the highest score wins, the smallest s among equals; reads without
    candidates get score MIN_SCORE and position -1.
"""
import numpy as np
import torch

from . import _lib
from .pairwise import MIN_SCORE, MODE_SEMIGLOBAL


class SeedExtendResult:
    def __init__(self, score, ref_start, ref_end, n_candidates, n_seed_hits):
        self.score = score              # int32[R]   best semiglobal score (MIN_SCORE: unmapped)
        self.ref_start = ref_start      # int64[R]   text position of the alignment's first y base (-1: unmapped)
        self.ref_end = ref_end          # int64[R]
        self.n_candidates = n_candidates
        self.n_seed_hits = n_seed_hits


def seed_and_extend(fm, aligner, d_text, n_text, d_reads, n_reads, read_len, seed_len=20, stride=10, max_occ=16,
                    pad=25, stream=None):
    """fm: FMIndex with an attached suffix array; aligner: pairwise.Aligner; d_text: uint8 cuda tensor of
    the text (sentinel excluded from n_text); d_reads: uint8 cuda tensor [n_reads * read_len]."""
    dev = d_reads.device
    st = stream if stream is not None else torch.cuda.current_stream(dev)
    sp = st.cuda_stream
    with torch.cuda.stream(st):
        offs = list(range(0, read_len - seed_len + 1, stride))
        S = len(offs)
        seeds = d_reads.view(n_reads, read_len).unfold(1, seed_len, stride)[:, :S, :].contiguous()
        n_q = n_reads * S
        pat_off = torch.arange(n_q + 1, dtype=torch.int64, device=dev) * seed_len
        tag = torch.empty(n_q, dtype=torch.uint8, device=dev)
        lo = torch.empty(n_q, dtype=torch.int64, device=dev)
        hi = torch.empty(n_q, dtype=torch.int64, device=dev)
        ml = torch.empty(n_q, dtype=torch.int32, device=dev)
        fm.backward_search_dev(n_q, seeds.data_ptr(), pat_off.data_ptr(), tag.data_ptr(), lo.data_ptr(),
                               hi.data_ptr(), ml.data_ptr(), sp)
        size = hi - lo
        vote = (tag == 0) & (size <= max_occ) & (size > 0)
        qidx = torch.nonzero(vote).flatten()
        n_iv = int(qidx.numel())
        if n_iv == 0:
            return _empty(n_reads, dev)
        iv_lo = lo[qidx].contiguous()
        iv_sz = size[qidx]
        out_off = torch.zeros(n_iv + 1, dtype=torch.int64, device=dev)
        out_off[1:] = torch.cumsum(iv_sz, 0)
        total = int(out_off[-1].item())
        pos = torch.empty(total, dtype=torch.int64, device=dev)
        fm.interval_occ_dev(n_iv, iv_lo.data_ptr(), out_off.data_ptr(), total, pos.data_ptr(), sp)
        # hit -> (read, proposed read start)
        hit_q = torch.repeat_interleave(qidx, iv_sz)
        read_id = hit_q // S
        seed_off = torch.tensor(offs, dtype=torch.int64, device=dev)[hit_q % S]
        start = pos - seed_off
        ok = (start >= 0) & (start < n_text)
        key = torch.unique(read_id[ok] * (n_text + 1) + start[ok])  # sorted: by read, then by start
        c_read = key // (n_text + 1)
        c_start = key % (n_text + 1)
        C = int(key.numel())
        if C == 0:
            return _empty(n_reads, dev, n_seed_hits=total)
        w_lo = torch.clamp(c_start - pad, min=0)
        w_hi = torch.clamp(c_start + read_len + pad, max=n_text)
        w_len = w_hi - w_lo
        y_off = torch.zeros(C + 1, dtype=torch.int64, device=dev)
        y_off[1:] = torch.cumsum(w_len, 0)
        ytot = int(y_off[-1].item())
        max_y = int(w_len.max().item())
        # gather the windows: element e of candidate c is text[w_lo[c] + e]
        cand_of = torch.repeat_interleave(torch.arange(C, device=dev), w_len)
        src = w_lo[cand_of] + (torch.arange(ytot, device=dev) - y_off[cand_of])
        y = d_text[src]
        x = d_reads.view(n_reads, read_len)[c_read].contiguous().view(-1)
        x_off = torch.arange(C + 1, dtype=torch.int64, device=dev) * read_len
        out = torch.empty(C * 64, dtype=torch.uint8, device=dev)
        aligner.align_dev(MODE_SEMIGLOBAL, C, x.data_ptr(), x_off.data_ptr(), y.data_ptr(), y_off.data_ptr(),
                          read_len, max_y, out.data_ptr(), 0, 0, sp)
        rec = out.view(torch.int32).view(C, 16)
        score = rec[:, 0].to(torch.int64)
        ystart = rec[:, 3].to(torch.int64)
        yend = rec[:, 4].to(torch.int64)
        # best candidate per read: highest score, then smallest start (candidates are sorted by start)
        best = torch.full((n_reads,), int(MIN_SCORE), dtype=torch.int64, device=dev)
        best.scatter_reduce_(0, c_read, score, reduce="amax", include_self=True)
        is_best = score == best[c_read]
        first = torch.full((n_reads,), C, dtype=torch.int64, device=dev)
        first.scatter_reduce_(0, c_read[is_best], torch.arange(C, device=dev)[is_best], reduce="amin", include_self=True)
        mapped = first < C
        pick = torch.where(mapped, first, torch.zeros_like(first))
        ref_start = torch.where(mapped, w_lo[pick] + ystart[pick], torch.full_like(first, -1))
        ref_end = torch.where(mapped, w_lo[pick] + yend[pick], torch.full_like(first, -1))
        return SeedExtendResult(best.to(torch.int32), ref_start, ref_end, C, total)


def _empty(n_reads, dev, n_seed_hits=0):
    return SeedExtendResult(torch.full((n_reads,), int(MIN_SCORE), dtype=torch.int32, device=dev),
                            torch.full((n_reads,), -1, dtype=torch.int64, device=dev),
                            torch.full((n_reads,), -1, dtype=torch.int64, device=dev), 0, n_seed_hits)
