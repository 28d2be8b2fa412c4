"""torch mirror of synth.py (same SplitMix64 streams, same bit fields) so that bench.py can
create the full-size workloads directly in HBM.  tests/test_gpu_synth.py pins it to synth.py."""
import torch

_GOLD = 0x9E3779B97F4A7C15 - (1 << 64)
_M1 = 0xBF58476D1CE4E5B9 - (1 << 64)
_M2 = 0x94D049BB133111EB - (1 << 64)


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _wrap(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def splitmix64(seed, count, device, start=0):
    idx = torch.arange(start + 1, start + count + 1, dtype=torch.int64, device=device)
    z = idx * _GOLD + _wrap(seed)
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _acgt(device):
    return torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)


def random_dna(n, seed, device, start=0):
    return _acgt(device)[_lsr(splitmix64(seed, n, device, start), 62)]


def _code_of(bases):
    # A C G T -> 0 1 2 3  ((b >> 1) & 3 gives A0 C1 T2 G3; swap the last two)
    c = ((bases >> 1) & 3).to(torch.int64)
    return torch.where(c >= 2, 5 - c, c)


def mutate_fixed(refs, seed, sub, ins, dele):
    n, L = refs.shape
    dev = refs.device
    r = splitmix64(seed, n * L, dev).view(n, L)
    u = _lsr(r, 40).to(torch.float64) * (1.0 / (1 << 24))
    deleted = u < dele
    subst = (u >= dele) & (u < dele + sub)
    code = _code_of(refs)
    shift = 1 + (_lsr(r, 38) & 3) % 3
    code = torch.where(subst, (code + shift) & 3, code)
    inserted = (_lsr(r, 12) & 0xFFFFFF).to(torch.float64) * (1.0 / (1 << 24)) < ins
    ins_code = _lsr(r, 10) & 3
    cand = torch.empty((n, 2 * L), dtype=torch.int64, device=dev)
    keep = torch.empty((n, 2 * L), dtype=torch.bool, device=dev)
    cand[:, 0::2], cand[:, 1::2] = code, ins_code
    keep[:, 0::2], keep[:, 1::2] = ~deleted, inserted
    dest = torch.cumsum(keep.to(torch.int32), dim=1) - 1
    lens = dest[:, -1] + 1
    out = _lsr(splitmix64(seed ^ 0x5bd1e995, n * L, dev), 62).view(n, L)
    sel = keep & (dest < L)
    rows = torch.arange(n, device=dev).view(n, 1).expand(n, 2 * L)
    out[rows[sel], dest[sel].to(torch.int64)] = cand[sel]
    return _acgt(dev)[out], torch.clamp(lens, max=L)


def sw_pairs(n_pairs, length, seed, device, sub=0.05, ins=0.01, dele=0.01):
    """Mirror of synth.sw_pairs (whole batch in one go; see sw_pairs_big for bench sizes)."""
    y = random_dna(n_pairs * length, seed, device).view(n_pairs, length)
    x, _ = mutate_fixed(y, seed + 1000003, sub, ins, dele)
    off = torch.arange(n_pairs + 1, dtype=torch.int64, device=device) * length
    return x.reshape(-1), off, y.reshape(-1), off.clone()


def sw_pairs_big(n_pairs, length, seed, device, sub=0.05, ins=0.01, dele=0.01, chunk=1 << 16):
    """Full-size batches in bounded memory: independent sub-seeds per chunk of pairs (this is the
    bench workload; tests compare sw_pairs with synth.sw_pairs)."""
    xs, ys = [], []
    for c0 in range(0, n_pairs, chunk):
        k = min(chunk, n_pairs - c0)
        x, _, y, _ = sw_pairs(k, length, seed + 7919 * (c0 // chunk), device, sub, ins, dele)
        xs.append(x)
        ys.append(y)
    off = torch.arange(n_pairs + 1, dtype=torch.int64, device=device) * length
    return torch.cat(xs), off, torch.cat(ys), off.clone()


def genome(n, seed, device):
    g = torch.empty(n + 1, dtype=torch.uint8, device=device)
    step = 1 << 24
    for s in range(0, n, step):
        e = min(n, s + step)
        g[s:e] = random_dna(e - s, seed, device, start=s)
    g[n] = ord("$")
    return g


def fm_patterns(text, n_q, plen, seed, frac_exact=0.799, frac_mut=0.2, chunk=1 << 20):
    """Mirror of synth.fm_patterns; `text` is a uint8 device tensor ending in '$'."""
    dev = text.device
    n = text.numel() - 1
    out = torch.empty((n_q, plen), dtype=torch.uint8, device=dev)
    ar = torch.arange(plen, dtype=torch.int64, device=dev)
    acgt = _acgt(dev)
    r_all = splitmix64(seed, n_q * 6, dev).view(n_q, 6)
    kind_all = _lsr(r_all[:, 1], 11).to(torch.float64) * (1.0 / (1 << 53))
    is_rand_all = kind_all >= frac_exact + frac_mut
    rand_rank = torch.cumsum(is_rand_all.to(torch.int64), 0) - 1
    for s in range(0, n_q, chunk):
        e = min(n_q, s + chunk)
        r = r_all[s:e]
        # unsigned modulo of a 64-bit draw by (n - plen + 1)
        mod = n - plen + 1
        hi = _lsr(r[:, 0], 32) % mod
        lo = (r[:, 0] & 0xFFFFFFFF) % mod
        pos = ((hi * ((1 << 32) % mod)) % mod + lo) % mod
        code = _code_of(text[pos.view(-1, 1) + ar.view(1, -1)])
        kind = kind_all[s:e]
        is_mut = (kind >= frac_exact) & (kind < frac_exact + frac_mut)
        is_rand = is_rand_all[s:e]
        nsub = 1 + _umod(r[:, 2], 3)
        for t in range(3):
            rows = torch.nonzero(is_mut & (nsub > t)).view(-1)
            d = r[rows, 3 + t]
            cols = _umod(d, plen)
            sh = 1 + _umod(_lsr(d, 32), 3)
            code[rows, cols] = (code[rows, cols] + sh) & 3
        rows = torch.nonzero(is_rand).view(-1)
        if rows.numel():
            k0 = rand_rank[s:e][rows]
            # draw index of (rank k, column c) in the random stream = k*plen + c
            idx = (k0.view(-1, 1) * plen + ar.view(1, -1)).view(-1)
            z = (idx + 1) * _GOLD + _wrap(seed + 77)
            z = (z ^ _lsr(z, 30)) * _M1
            z = (z ^ _lsr(z, 27)) * _M2
            z = z ^ _lsr(z, 31)
            code[rows] = _lsr(z, 62).view(-1, plen)
        out[s:e] = acgt[code]
    off = torch.arange(n_q + 1, dtype=torch.int64, device=dev) * plen
    return out.view(-1), off


def _umod(z, m):
    """unsigned (z mod m) for int64 tensors holding uint64 bit patterns, m < 2^31"""
    hi = _lsr(z, 32) % m
    lo = (z & 0xFFFFFFFF) % m
    return ((hi * ((1 << 32) % m)) % m + lo) % m


def reads_from_genome(text, n_reads, length, seed, sub=0.05, ins=0.01, dele=0.01, chunk=1 << 16):
    """cfg 5 reads: windows of the genome at SplitMix64 positions, mutated like cfg 2 (mutate_fixed).
    Returns (reads uint8[n_reads * length], starts int64[n_reads])."""
    dev = text.device
    n = text.numel() - 1
    starts = (_lsr(splitmix64(seed, n_reads, dev), 1) % (n - length)).to(torch.int64)
    ar = torch.arange(length, dtype=torch.int64, device=dev)
    out = []
    for c0 in range(0, n_reads, chunk):
        s = starts[c0:c0 + chunk]
        refs = text[(s[:, None] + ar[None, :]).reshape(-1)].view(-1, length)
        x, _ = mutate_fixed(refs, seed + 1000003 + 7919 * (c0 // chunk), sub, ins, dele)
        out.append(x.reshape(-1))
    return torch.cat(out), starts


AMINO = b"ARNDCQEGHILKMFPSTWYV"


def protein_pairs(n_pairs, length, seed, device, sub=0.15, chunk=1 << 16):
    """Protein pairs for the tabulated-scoring (BLOSUM62) bench leg: y uniform over the 20 amino acids,
    x = y with per-residue substitutions, one residue deleted and one inserted per pair (so the alignment
    has gaps), same length.  Returns (x, off, y, off)."""
    aa = torch.tensor(list(AMINO), dtype=torch.uint8, device=device)
    xs, ys = [], []
    ar = torch.arange(length, dtype=torch.int64, device=device)
    for c0 in range(0, n_pairs, chunk):
        k = min(chunk, n_pairs - c0)
        sd = seed + 7919 * (c0 // chunk)
        y = _umod(splitmix64(sd, k * length, device), 20).view(k, length)
        r = splitmix64(sd + 1000003, k * length, device).view(k, length)
        u = _lsr(r, 40).to(torch.float64) * (1.0 / (1 << 24))
        x = torch.where(u < sub, _umod(_lsr(r, 8), 20), y)
        p = splitmix64(sd + 2000003, k * 2, device).view(k, 2)
        dpos = _umod(p[:, 0], length).view(k, 1)      # residue dropped from x ...
        ipos = _umod(p[:, 1], length).view(k, 1)      # ... and a residue inserted elsewhere
        src = ar.view(1, -1) + (ar.view(1, -1) >= dpos).to(torch.int64)
        x = torch.gather(x, 1, torch.clamp(src, max=length - 1))
        src2 = ar.view(1, -1) - (ar.view(1, -1) > ipos).to(torch.int64)
        x = torch.gather(x, 1, src2)
        xs.append(aa[x].reshape(-1))
        ys.append(aa[y].reshape(-1))
    off = torch.arange(n_pairs + 1, dtype=torch.int64, device=device) * length
    return torch.cat(xs), off, torch.cat(ys), off.clone()
