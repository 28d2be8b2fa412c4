"""The oracle's FASTQ reader / Record::check against the reference's own tests (io/fastq.rs) and its cigar
restatement against the documented example."""
import numpy as np

import oracle_py as orc
from kat_util import load

K = load("fastq_kats.json")
OPK = {"Match": 0, "Subst": 1, "Del": 2, "Ins": 3}
MODES = {"custom": 0, "global": 1, "semiglobal": 2, "local": 3}


def test_reader_kats():
    for c in K["reader"]:
        recs, status, _ = orc.fastq_parse(c["text"].encode())
        assert status == c["status"], c["name"]
        if "records" in c:
            got = [{"id": r["id"].decode(), "desc": None if r["desc"] is None else r["desc"].decode(),
                    "seq": r["seq"].decode(), "qual": r["qual"].decode(), "check": r["check"]} for r in recs]
            assert got == c["records"], c["name"]
        else:
            assert len(recs) == c["n_records"], c["name"]


def test_check_kats():
    for c in K["check"]:
        text = ("@%s\n%s\n+\n%s\n" % (c["id"], c["seq"], c["qual"])).encode()
        recs, status, _ = orc.fastq_parse(text)
        assert status == "ok" and len(recs) == 1
        assert recs[0]["check"] == c["check"], c["name"]
    # non-ASCII sequence / qualities (fastq.rs:741-751, 764-771)
    recs, _, _ = orc.fastq_parse("@id\nATéC\n+\nQQQQ\n".encode())
    assert recs[0]["check"] == "NonAsciiSequence"
    recs, _, _ = orc.fastq_parse("@id\nATGC\n+\nQQéQ\n".encode())
    assert recs[0]["check"] == "NonAsciiQualities"


def test_reader_edge_cases():
    # trim_end is Unicode-aware, ids end at the first space, a header may have no description
    recs, st, _ = orc.fastq_parse(b"@r1\r\nACGT \t\r\n+r1\r\nIIII\r\n@r2  two  spaces \nAC\n+\nII")
    assert st == "ok" and [r["id"] for r in recs] == [b"r1", b"r2"]
    assert recs[0]["desc"] is None and recs[0]["seq"] == b"ACGT" and recs[0]["qual"] == b"IIII"
    assert recs[1]["desc"] == b" two  spaces" and recs[1]["qual"] == b"II"
    recs, st, _ = orc.fastq_parse("@x\nAC \n+\nII 　\n".encode())
    assert st == "ok" and recs[0]["seq"] == b"AC" and recs[0]["qual"] == b"II"
    assert orc.fastq_parse(b"")[1] == "ok" and orc.fastq_parse(b"\n")[1] == "MissingAt"
    assert orc.fastq_parse(b"@a\n+\nII\n")[1] == "IncompleteRecord"      # no sequence line: no quality line is read
    assert orc.fastq_parse(b"@a\nAC\n+\n\n")[1] == "IncompleteRecord"    # quality trims to nothing
    recs, st, ep = orc.fastq_parse(b"@a\nAC\n+\nII\n@b\n\xff\xfe\n+\nII\n")
    assert st == "Io" and len(recs) == 1 and ep == 14
    # the quality lines are counted, not recognised: they may start with '@' or '+'
    recs, st, _ = orc.fastq_parse(b"@a\nAC\nGT\n+\n@I\n+I\n@b\nA\n+\nI\n")
    assert st == "ok" and recs[0]["qual"] == b"@I+I" and recs[1]["id"] == b"b"


def test_cigar_kat_and_rules():
    for c in K["cigar"]:
        aln = {"xstart": c["xstart"], "xend": c["xend"], "xlen": c["xlen"], "mode": MODES[c["mode"]]}
        ops = np.array([OPK[o] for o in c["ops"]], dtype=np.uint64)
        assert orc.cigar(aln, ops, False) == c["soft"]
        assert orc.cigar(aln, ops, True) == c["hard"]
    aln = {"xstart": 0, "xend": 4, "xlen": 4, "mode": 3}
    assert orc.cigar(aln, np.array([0, 0, 0, 0], dtype=np.uint64), False) == "4="
    assert orc.cigar(aln, np.zeros(0, dtype=np.uint64), False) == ""
    assert orc.cigar(dict(aln, mode=0), np.array([0], dtype=np.uint64), False) is None
