"""Deterministic synthetic workloads: SplitMix64-seeded, base = "ACGT"[next()>>62].
Shared by the tests, bench.py and the oracle's CPU-baseline leg."""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(seed, count, start=0):
    """`count` outputs of SplitMix64(seed) starting at draw index `start` (counter form)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def random_dna(n, seed, start=0):
    return ACGT[(splitmix64(seed, n, start) >> np.uint64(62)).astype(np.intp)]


def _unit(u64):
    return (u64 >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def mutate_fixed(refs, seed, sub, ins, dele):
    """refs: uint8[n, L].  One SplitMix64 draw per base, split into bit fields:
    bits 40-63 -> u in [0,1): deletion if u < dele, substitution if u < dele+sub;
    bits 38-39 -> substitution shift 1..3 (mod 3 + 1); bits 12-35 -> insertion if < ins;
    bits 10-11 -> inserted base.  Result padded with random bases / truncated to L."""
    n, L = refs.shape
    r = splitmix64(seed, n * L).reshape(n, L)
    u = (r >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    deleted = u < dele
    subst = (u >= dele) & (u < dele + sub)
    code = np.searchsorted(ACGT, refs).astype(np.uint8)  # A C G T -> 0..3
    shift = (1 + ((r >> np.uint64(38)) & np.uint64(3)) % np.uint64(3)).astype(np.uint8)
    code = np.where(subst, (code + shift) & 3, code)
    inserted = ((r >> np.uint64(12)) & np.uint64(0xFFFFFF)).astype(np.float64) * (1.0 / (1 << 24)) < ins
    ins_code = ((r >> np.uint64(10)) & np.uint64(3)).astype(np.uint8)
    cand = np.empty((n, 2 * L), dtype=np.uint8)
    keep = np.empty((n, 2 * L), dtype=bool)
    cand[:, 0::2], cand[:, 1::2] = code, ins_code
    keep[:, 0::2], keep[:, 1::2] = ~deleted, inserted
    dest = np.cumsum(keep, axis=1) - 1
    lens = dest[:, -1] + 1
    pad = (splitmix64(seed ^ 0x5bd1e995, n * L) >> np.uint64(62)).astype(np.uint8).reshape(n, L)
    out = pad.copy()
    rows = np.broadcast_to(np.arange(n)[:, None], keep.shape)
    sel = keep & (dest < L)
    out[rows[sel], dest[sel]] = cand[sel]
    return ACGT[out], np.minimum(lens, L)


def sw_pairs(n_pairs, length, seed, sub=0.05, ins=0.01, dele=0.01):
    """cfg 1/2: refs y uniform ACGT, x = mutated y padded/truncated to `length`.
    Returns (x uint8[n*L], x_off, y, y_off)."""
    y = random_dna(n_pairs * length, seed).reshape(n_pairs, length)
    x, _ = mutate_fixed(y, seed + 1000003, sub, ins, dele)
    off = np.arange(n_pairs + 1, dtype=np.uint64) * np.uint64(length)
    return x.reshape(-1), off, y.reshape(-1), off.copy()


def ragged_pairs(n_pairs, max_len, seed, alphabet=b"ACGT", min_len=0):
    """Pairs of independent lengths in [min_len, max_len]; x is a noisy copy of a slice of y
    half of the time, unrelated otherwise (exercises every traceback move)."""
    rng = np.random.default_rng(seed)
    al = np.frombuffer(alphabet, dtype=np.uint8)
    xs, ys = [], []
    for _ in range(n_pairs):
        ly = int(rng.integers(min_len, max_len + 1))
        y = al[rng.integers(0, len(al), size=ly)]
        if rng.random() < 0.6 and ly > 0:
            a = int(rng.integers(0, ly))
            b = int(rng.integers(a, ly + 1))
            x = y[a:b].copy()
            if len(x):
                mut = rng.random(len(x)) < 0.15
                x[mut] = al[rng.integers(0, len(al), size=int(mut.sum()))]
                if rng.random() < 0.5 and len(x) > 2:
                    cut = int(rng.integers(0, len(x)))
                    x = np.concatenate([x[:cut], x[cut + int(rng.integers(1, 4)):]])
                if rng.random() < 0.5:
                    cut = int(rng.integers(0, len(x) + 1))
                    x = np.concatenate([x[:cut], al[rng.integers(0, len(al), size=int(rng.integers(1, 4)))], x[cut:]])
            x = x[:max_len]
        else:
            x = al[rng.integers(0, len(al), size=int(rng.integers(min_len, max_len + 1)))]
        xs.append(x.astype(np.uint8).tobytes())
        ys.append(y.astype(np.uint8).tobytes())
    return xs, ys


def genome(n, seed):
    """cfg 3: n uniform ACGT bases + '$'."""
    g = np.empty(n + 1, dtype=np.uint8)
    g[:n] = random_dna(n, seed)
    g[n] = ord("$")
    return g


def fm_patterns(text, n_q, plen, seed, frac_exact=0.799, frac_mut=0.2):
    """cfg 3 patterns over ACGT: frac_exact exact substrings, frac_mut substrings with 1-3
    substitutions, the rest uniform random.  Returns (pat uint8[n_q*plen], off)."""
    n = len(text) - 1  # without the sentinel
    r = splitmix64(seed, n_q * 6).reshape(n_q, 6)
    pos = (r[:, 0] % np.uint64(n - plen + 1)).astype(np.int64)
    kind = _unit(r[:, 1])
    out = np.empty((n_q, plen), dtype=np.uint8)
    step = 1 << 18
    ar = np.arange(plen, dtype=np.int64)
    for s in range(0, n_q, step):
        e = min(n_q, s + step)
        out[s:e] = text[pos[s:e, None] + ar[None, :]]
    is_mut = (kind >= frac_exact) & (kind < frac_exact + frac_mut)
    is_rand = kind >= frac_exact + frac_mut
    nsub = 1 + (r[:, 2] % np.uint64(3)).astype(np.int64)
    code = np.searchsorted(ACGT, out).astype(np.uint8)
    for t in range(3):
        rows = np.nonzero(is_mut & (nsub > t))[0]
        cols = (r[rows, 3 + t] % np.uint64(plen)).astype(np.int64)
        sh = (1 + ((r[rows, 3 + t] >> np.uint64(32)) % np.uint64(3))).astype(np.uint8)
        code[rows, cols] = (code[rows, cols] + sh) & 3
    nr = int(is_rand.sum())
    if nr:
        code[is_rand] = (splitmix64(seed + 77, nr * plen) >> np.uint64(62)).astype(np.uint8).reshape(nr, plen)
    off = np.arange(n_q + 1, dtype=np.uint64) * np.uint64(plen)
    return ACGT[code].reshape(-1), off


def fastq_text(n_reads, length, seed):
    """A four-line FASTQ of `n_reads` records as a uint8 array: '@r<10 digits> s<seed>', ACGT read, '+', Phred+33
    qualities in [33, 74)."""
    rng = np.random.default_rng(seed)
    hdr = np.frombuffer(b"@r0000000000 s%04d\n" % (seed % 10000), dtype=np.uint8)
    rec_len = len(hdr) + length + 1 + 2 + length + 1
    out = np.empty((n_reads, rec_len), dtype=np.uint8)
    out[:, :len(hdr)] = hdr
    idx = np.arange(n_reads, dtype=np.int64)
    for d in range(10):
        out[:, 2 + 9 - d] = 48 + (idx // 10 ** d) % 10
    o = len(hdr)
    out[:, o:o + length] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n_reads, length), dtype=np.uint8)]
    out[:, o + length] = 10
    out[:, o + length + 1] = ord("+")
    out[:, o + length + 2] = 10
    o2 = o + length + 3
    out[:, o2:o2 + length] = rng.integers(33, 74, size=(n_reads, length), dtype=np.uint8)
    out[:, o2 + length] = 10
    return out.reshape(-1)
