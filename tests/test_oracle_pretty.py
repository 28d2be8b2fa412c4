"""bio-types `Alignment::pretty(x, y, ncol)` restated in the oracle (PARITY UNPINNED: the crate is not in the
reference tree, rust-bio only prints the result — pairwise/banded.rs:1805,1820,1833,1877).  These checks pin the
restatement to the layout the crate documents for its example pair: three rows (x, marks, y), '|' match, '\\'
mismatch, '+' insertion, 'x' deletion, ' ' for what the standard modes clip implicitly, blocks of `ncol` columns
each followed by two empty lines."""
import oracle_py as orc

X, Y = b"CCGTCCGGCAAGGG", b"AAAAACCGTTGACGGCCAA"
KIND = {"M": 0, "S": 1, "D": 2, "I": 3, "X": 4, "Y": 5}


def ops_u64(tokens):
    return [KIND[t[0]] | ((int(t[1:]) if len(t) > 1 else 0) << 8) for t in tokens]


def run(mode, ncol=100, **kw):
    sc = orc.make_scoring(-5, -1, 1, -1, **kw)
    a = orc.align(sc, mode, X, Y)
    return a, orc.pretty(dict(a, mode=mode), ops_u64(a["ops"]), X, Y, ncol)


def test_local_pads_both_flanks_with_blanks():
    a, s = run("local")
    assert s == ("     CCGTCCGGCAAGGG          \n"
                 "     ||||                    \n"
                 "AAAAACCGT          TGACGGCCAA\n\n\n")


def test_global_shows_gaps_and_mismatches():
    a, s = run("global")
    assert s == ("-----CCGTCCGGCAAGGG\n"
                 "xxxxx||||\\\\\\\\\\\\\\\\\\\\\n"
                 "AAAAACCGTTGACGGCCAA\n\n\n")


def test_blocks_of_ncol_columns():
    a, s = run("local", ncol=10)
    blocks = s.split("\n\n\n")
    assert blocks[-1] == "" and len(blocks) == 4  # 29 columns -> 10 + 10 + 9
    rows = [b.split("\n") for b in blocks[:-1]]
    assert [len(r[0]) for r in rows] == [10, 10, 9] and all(len(r) == 3 and len(r[0]) == len(r[1]) == len(r[2]) for r in rows)
    assert "".join(r[2] for r in rows) == "AAAAACCGT          TGACGGCCAA"


def test_custom_mode_walks_its_clip_operations():
    # AlignmentMode::Custom: no implicit flanks; Xclip(n) / Yclip(n) print the first n symbols of the sequence
    a, s = run("custom", xclip_prefix=-1, xclip_suffix=-1, yclip_prefix=0, yclip_suffix=0)
    assert any(t[0] in "XY" for t in a["ops"])
    rows = s.split("\n")
    assert len(rows[0]) == len(rows[1]) == len(rows[2])
    n_cols = sum(int(t[1:]) if t[0] in "XY" else 1 for t in a["ops"])
    assert len(rows[0]) == n_cols


def test_empty_alignment_is_empty_string():
    assert orc.pretty({"mode": "local", "xstart": 0, "ystart": 0}, [], b"", b"", 80) == ""
