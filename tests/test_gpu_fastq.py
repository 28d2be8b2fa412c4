"""GPU parity of the FASTQ ingest (csrc/fastq_ingest.hip: bio::io::fastq::Reader::read / Records, Record::check)
and of the CIGAR emission against the reference's own tests and the CPU oracle."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import load
from rust_bio_amd import _lib, fastq
from rust_bio_amd.pairwise import Aligner, Alignment, Scoring, cigar_batch

pytestmark = pytest.mark.gpu
K = load("fastq_kats.json")


def same_as_oracle(text):
    want, wst, wpos = orc.fastq_parse(text)
    p = fastq.parse_arrays(text)
    assert p.status == wst, (text[:80], p.status, wst)
    assert len(p) == len(want)
    if wst != "ok":
        assert p.err_pos == wpos
    for k, w in enumerate(want):
        r = p.record(k)
        assert (r._id, r._desc, r._seq, r._qual) == (w["id"], w["desc"], w["seq"], w["qual"]), (k, r, w)
        assert fastq.CHECK[r._check] == w["check"], (k, r, w)
    return p


def test_reference_reader_kats():
    for c in K["reader"]:
        p = same_as_oracle(c["text"].encode())
        assert p.status == c["status"], c["name"]
        if "records" in c:
            got = [{"id": r.id(), "desc": r.desc(), "seq": r.seq().decode(), "qual": r.qual().decode(),
                    "check": fastq.CHECK[r._check]} for r in (p.record(k) for k in range(len(p)))]
            assert got == c["records"], c["name"]
        else:
            assert len(p) == c["n_records"], c["name"]
    # the iterator: records, then the error (fastq.rs:859-866)
    it = fastq.Reader(b"@a\nAC\n+\nII\n@id description\nACGT\n+\n").records()
    assert next(it).id() == "a"
    with pytest.raises(fastq.ReadError) as e:
        next(it)
    assert e.value.kind == "IncompleteRecord"


def test_check_kats():
    for c in K["check"]:
        p = fastq.parse_arrays(("@%s\n%s\n+\n%s\n" % (c["id"], c["seq"], c["qual"])).encode())
        r = p.record(0)
        if c["check"] == "ok":
            r.check()
        else:
            with pytest.raises(fastq.CheckError) as e:
                r.check()
            assert e.value.kind == c["check"], c["name"]


def test_edge_cases_like_the_oracle():
    for t in (b"", b"\n", b"@a\n+\nII\n", b"@a\nAC\n+\n\n", b"@a\nAC\n+\nII", b"@r1\r\nACGT \t\r\n+r1\r\nIIII\r\n@r2  two  spaces \nAC\n+\nII",
              "@x\nAC \n+\nII 　\n".encode(), b"@a\nAC\n+\nII\n@b\n\xff\xfe\n+\nII\n", b"@a\nAC\nGT\n+\n@I\n+I\n@b\nA\n+\nI\n",
              b"@a\nAC\n+\nII\nACGT\n", b"@a\nAC\n+\nII\n@b\nAC\n", b"@\nAC\n+\nII\n", "@é ü\nAC\n+\nII\n".encode(), b"@a\n+AC\n", b"@a"):
        same_as_oracle(t)


def make_fastq(rng, n, wrapped=0.0, crlf=False, bad_at=None):
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    out = []
    for k in range(n):
        ln = int(rng.integers(1, 160))
        seq = alpha[rng.integers(0, 5, size=ln)].tobytes()
        qual = (rng.integers(33, 75, size=ln).astype(np.uint8)).tobytes()
        nl = b"\r\n" if crlf else b"\n"
        hdr = b"@read%d" % k + (b" desc %d x" % k if rng.random() < 0.5 else b"")
        if rng.random() < wrapped:
            w = int(rng.integers(1, ln + 1))
            sl = [seq[i:i + w] for i in range(0, ln, w)]
            ql = [qual[i:i + w] for i in range(0, ln, w)]
            out.append(hdr + nl + nl.join(sl) + nl + b"+" + nl + nl.join(ql) + nl)
        else:
            out.append(hdr + nl + seq + nl + b"+" + (hdr[1:] if rng.random() < 0.2 else b"") + nl + qual + nl)
        if bad_at is not None and k == bad_at:
            out.append(rng.choice([b"garbage line\n", b"@trunc\nACGT\n", b"@x\nAC\n+\n\n", b"\n"]))
    return b"".join(out)


def test_random_files_four_line_wrapped_and_broken():
    rng = np.random.default_rng(3)
    for trial in range(40):
        n = int(rng.integers(1, 400))
        t = make_fastq(rng, n, wrapped=float(rng.choice([0.0, 0.0, 0.1, 1.0])), crlf=rng.random() < 0.3,
                       bad_at=int(rng.integers(0, n)) if rng.random() < 0.4 else None)
        if rng.random() < 0.3 and t.endswith(b"\n"):
            t = t[:-1]  # no newline at the end of the file
        same_as_oracle(t)


def test_large_four_line_file_and_truncated_tail():
    rng = np.random.default_rng(4)
    t = make_fastq(rng, 60_000)
    p = same_as_oracle(t)
    assert len(p) == 60_000 and p.status == "ok"
    p = same_as_oracle(t[:-40])  # the last record loses its tail: every record before it is still returned
    assert len(p) >= 59_998


def test_parse_dev_feeds_the_aligner():
    import torch
    rng = np.random.default_rng(5)
    t = make_fastq(rng, 3000)
    d = torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda()
    n, st, _, d_recs, d_seq, d_so, d_qual, d_qo = fastq.parse_dev(d)
    want, _, _ = orc.fastq_parse(t)
    assert n == len(want) and st == "ok"
    so = d_so.cpu().numpy()
    seq = d_seq.cpu().numpy()
    for k in (0, 1, 17, n - 1):
        assert seq[so[k]:so[k + 1]].tobytes() == want[k]["seq"]
    # read k against read k (local): perfect alignments, straight from the device buffers
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    d_out = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
    al.align_dev(3, n, d_seq.data_ptr(), d_so.data_ptr(), d_seq.data_ptr(), d_so.data_ptr(), 160, 160, d_out.data_ptr(), 0, 0,
                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    score = d_out.view(torch.int32).view(n, 16)[:, 0].cpu().numpy()
    assert (score == np.diff(so).astype(np.int64)).all()


def test_cigar_kats_and_batches_against_the_oracle():
    OPK = {"Match": "M", "Subst": "S", "Ins": "I", "Del": "D"}
    for c in K["cigar"]:
        a = Alignment(0, 0, c["xstart"], 0, c["xend"], 0, c["xlen"], [OPK[o] for o in c["ops"]], c["mode"].capitalize())
        assert a.cigar(False) == c["soft"] and a.cigar(True) == c["hard"]
    with pytest.raises(AssertionError):
        Alignment(0, 0, 0, 0, 1, 1, 1, ["M"], "Custom").cigar(False)
    assert Alignment(0, 0, 0, 0, 0, 0, 4, [], "Local").cigar(False) == ""
    # alignments of random reads in the three supported modes
    from rust_bio_amd import synth
    xs, ys = synth.ragged_pairs(500, 120, seed=9, min_len=1)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    for mode in (1, 2, 3):
        out, ops = al.align_arrays(mode, x, xo, y, yo)
        for hard in (False, True):
            got = cigar_batch(out, ops, hard)
            for p in range(0, len(out), 7):
                o = ops[int(out["ops_off"][p]):int(out["ops_off"][p]) + int(out["n_ops"][p])].astype(np.uint64)
                want = orc.cigar({"xstart": int(out["xstart"][p]), "xend": int(out["xend"][p]), "xlen": int(out["xlen"][p]), "mode": mode}, o, hard)
                assert got[p] == want, (mode, p, got[p], want)


def test_long_reads_and_header_spaces_at_every_offset():
    """Lines beyond 256 bytes on average take the wavefront-per-record gather, short ones 16 lanes per record; both copy in
    16-byte pieces whose source and destination alignments take every value; the header's first space is found eight bytes
    at a time from the first 8-byte boundary on."""
    rng = np.random.default_rng(9)
    alpha = np.frombuffer(b"ACGTNacgtn-.*", dtype=np.uint8)
    for lo, hi in ((257, 900), (1, 40), (120, 330)):
        out = []
        for k in range(300):
            ln = int(rng.integers(lo, hi))
            seq = alpha[rng.integers(0, len(alpha), size=ln)].tobytes()
            qual = rng.integers(33, 75, size=ln).astype(np.uint8).tobytes()
            idl = int(rng.integers(1, 30))
            hdr = b"@" + bytes(rng.integers(97, 123, size=idl).astype(np.uint8))
            if k % 3:
                hdr += b" " + bytes(rng.integers(97, 123, size=int(rng.integers(0, 25))).astype(np.uint8))
            if k % 7 == 0:
                hdr += b"  second space"
            out.append(hdr + b"\n" + seq + b"\n+\n" + qual + b"\n")
        t = b"".join(out)
        for cut in (0, 1, 2, 3, 5):  # shift every alignment by dropping leading records' bytes... a prefix record of odd length
            same_as_oracle((b"@p\n" + b"A" * cut + b"\n+\n" + b"I" * cut + b"\n" if cut else b"") + t)


def both_paths_like_the_oracle(text):
    """the one-pass kernel (or, where it raises its flag, the general kernels behind it) and the general kernels alone"""
    p = same_as_oracle(text)
    ctx = _lib.Context(0)
    ctx.set_option("fq_no_fused", 1)
    q = fastq.parse_arrays(text, ctx=ctx)
    assert (p.status, p.err_pos, len(p)) == (q.status, q.err_pos, len(q))
    assert (p.recs == q.recs).all() and (p.seq == q.seq).all() and (p.qual == q.qual).all()
    assert (p.seq_off == q.seq_off).all() and (p.qual_off == q.qual_off).all()
    ctx.close()
    return p


def test_one_pass_reader_on_tiles_halos_and_what_it_leaves_to_the_general_kernels():
    """Round 6, fq_fused_kernel: records that straddle its 16 KB tiles (every phase of the line index at a tile boundary),
    lines longer than a tile (the window search in front of a tile runs over several windows), CRLF, a text without a final
    newline ending exactly on / next to a tile boundary, empty ids, invalid and unequal records (Record::check), and the
    inputs it must hand over: a non-ASCII byte, a wrapped record in the middle, a line count that is no multiple of four,
    more than 512 records in a tile (reads of a few bases), more than 4096 newlines in a tile."""
    rng = np.random.default_rng(21)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)

    def rec(k, ln, nl=b"\n", idl=None, desc=True):
        seq = alpha[rng.integers(0, 5, size=ln)].tobytes()
        qual = rng.integers(33, 75, size=ln).astype(np.uint8).tobytes()
        hdr = b"@" + (b"r%d" % k if idl is None else b"x" * idl) + (b" d %d" % k if desc else b"")
        return hdr + nl + seq + nl + b"+" + nl + qual + nl

    # 150 bp reads, 3000 records (~1 MB: 60 tiles), shifted byte by byte so that tile boundaries fall everywhere in a record
    body = b"".join(rec(k, 150, desc=k % 2 == 0) for k in range(3000))
    for shift in (0, 1, 7, 33, 150, 151, 152, 153, 154, 160, 300, 305):
        pre = (b"@s\n" + b"A" * shift + b"\n+\n" + b"I" * shift + b"\n") if shift else b""
        p = both_paths_like_the_oracle(pre + body)
        assert p.status == "ok" and len(p) == 3000 + (1 if shift else 0)
    # the text ends without a newline, exactly at a tile boundary and one byte either side of it
    for total in (16384, 16383, 16385, 32768, 32769):
        t = b"".join(rec(k, 100) for k in range(total // 100))[:0]
        parts, size, k = [], 0, 0
        while True:
            r = rec(k, int(rng.integers(20, 200)))
            if size + len(r) > total - 300:
                break
            parts.append(r)
            size += len(r)
            k += 1
        tail_len = (total - size - len(b"@t\n\n+\n")) // 2
        odd = total - size - len(b"@t\n\n+\n") - 2 * tail_len
        last = b"@t" + b"y" * odd + b"\n" + b"A" * tail_len + b"\n+\n" + b"I" * tail_len  # no newline at the end
        t = b"".join(parts) + last
        assert len(t) == total
        p = both_paths_like_the_oracle(t)
        assert p.status == "ok" and len(p) == len(parts) + 1
        both_paths_like_the_oracle(t + b"\n")
    # long reads: lines of 20 - 70 kb (several tiles per line; the four newlines in front of a tile are far away)
    both_paths_like_the_oracle(b"".join(rec(k, int(rng.integers(20_000, 70_000))) for k in range(12)))
    both_paths_like_the_oracle(b"".join(rec(k, int(rng.integers(2_000, 9_000)), nl=b"\r\n") for k in range(60)))
    # records of 20 - 30 KB: they stay on the one-pass path (all four lines begin within 31 KB of the tile their last line ends
    # in), with more bytes to copy than the tile's piece tables hold; mixed with short reads so that tiles have both
    both_paths_like_the_oracle(b"".join(rec(k, int(rng.integers(9_000, 14_500)) if k % 3 else int(rng.integers(30, 300))) for k in range(90)))
    both_paths_like_the_oracle(b"".join(rec(k, 15_000 + k) for k in range(40)))
    # Record::check on the one-pass path: empty id, invalid sequence byte, unequal lengths
    t = rec(0, 50) + b"@\nACGT\n+\nIIII\n" + b"@bad\nAC#T\n+\nIIII\n" + b"@uneq\nACGT\n+\nIII\n" + b"@ lead space\nAC\n+\nII\n" + b"@ok  two\nA\n+\nI\n"
    p = both_paths_like_the_oracle(t * 300)
    assert len(p) == 1800
    # handed over to the general kernels: the result must still be the reference's
    base = b"".join(rec(k, 80) for k in range(800))
    both_paths_like_the_oracle(base + "@é\nAC\n+\nII\n".encode() + base)                      # a non-ASCII byte
    both_paths_like_the_oracle(base + b"@w\nAC\nGT\n+\nII\nII\n" + base)                       # a wrapped record: six lines
    both_paths_like_the_oracle(base + b"@trunc\nACGT\n")                                       # lines % 4 != 0
    both_paths_like_the_oracle(base[:-1] + b"\n\n")                                            # an empty line at the end
    both_paths_like_the_oracle(b"".join(rec(k, 3, idl=1, desc=False) for k in range(5000)))    # ~1300 records per tile
    both_paths_like_the_oracle(base + b"@n\n" + b"\n" * 9000 + base)                           # > 4096 newlines in a tile
    both_paths_like_the_oracle(b"@a\nAC\n+\nII\n" * 3 + b"garbage\nAC\n+\nII\n" + b"@a\nAC\n+\nII\n" * 3)  # MissingAt in the middle
    both_paths_like_the_oracle(b"@a\nAC\n+\n\n" + b"@a\nAC\n+\nII\n" * 3)                      # an empty quality line: IncompleteRecord
    both_paths_like_the_oracle(b"@a\n+C\n+\nII\n" * 5)                                         # a sequence line that starts with '+'
