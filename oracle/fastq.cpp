// ORACLE (test infrastructure, not product code) — see oracle.h.
// bio::io::fastq::Reader::read / Records (io/fastq.rs:266-303, 508-527), Record::check (388-410) restated on a
// byte buffer, and bio-types 1.0 `Alignment::cigar` (third-party crate, absent from /root/reference; restated
// from its published documentation — PARITY UNPINNED: the reference tree holds no test that asserts a cigar).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "oracle.h"

namespace {
// std::io::BufRead::read_line: bytes up to and including '\n' (or to the end of the input)
inline uint64_t line_end(const uint8_t* t, uint64_t len, uint64_t pos) {
    while (pos < len && t[pos] != '\n') pos++;
    return pos < len ? pos + 1 : len;
}
// read_line appends to a String: the bytes must be UTF-8 or the read fails (io::ErrorKind::InvalidData)
bool valid_utf8(const uint8_t* s, uint64_t n) {
    uint64_t i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        if (c < 0x80) { i++; continue; }
        int k;
        uint32_t cp, lo;
        if (c >= 0xC2 && c <= 0xDF) { k = 1; cp = c & 0x1F; lo = 0x80; }
        else if (c >= 0xE0 && c <= 0xEF) { k = 2; cp = c & 0x0F; lo = 0x800; }
        else if (c >= 0xF0 && c <= 0xF4) { k = 3; cp = c & 0x07; lo = 0x10000; }
        else return false;
        for (int j = 1; j <= k; j++) {
            if (i + j >= n) return false;
            if ((s[i + j] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (s[i + j] & 0x3F);
        }
        if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += k + 1;
    }
    return true;
}
// char::is_whitespace (Unicode White_Space)
inline bool is_ws(uint32_t cp) {
    return (cp >= 9 && cp <= 13) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
           cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
// str::trim_end on valid UTF-8: length of the prefix that remains
uint64_t trim_end(const uint8_t* s, uint64_t n) {
    while (n) {
        uint64_t b = n - 1;
        while (b > 0 && (s[b] & 0xC0) == 0x80) b--;  // start of the last character
        uint32_t cp;
        const uint8_t c = s[b];
        if (c < 0x80) cp = c;
        else if (c < 0xE0) cp = c & 0x1F;
        else if (c < 0xF0) cp = c & 0x0F;
        else cp = c & 0x07;
        for (uint64_t j = b + 1; j < n; j++) cp = (cp << 6) | (s[j] & 0x3F);
        if (!is_ws(cp)) break;
        n = b;
    }
    return n;
}
}  // namespace

extern "C" int orc_fastq_parse(const uint8_t* t, uint64_t len, orc_fastq_rec_t* recs, uint64_t rec_cap, uint8_t* seq,
                               uint8_t* qual, uint64_t* n_records, int32_t* status, uint64_t* err_pos) {
    uint64_t pos = 0, nrec = 0, so = 0, qo = 0;
    *status = ORC_FASTQ_OK;
    *err_pos = 0;
    auto fail = [&](int32_t st, uint64_t at) {
        *status = st;
        *err_pos = at;
        *n_records = nrec;
        return 0;
    };
    while (true) {
        // fastq.rs:266-270
        const uint64_t h0 = pos, h1 = line_end(t, len, pos);
        if (h1 == h0) break;  // empty line buffer: no more records (fastq.rs:228, 517-524)
        if (!valid_utf8(t + h0, h1 - h0)) return fail(ORC_FASTQ_IO, h0);
        if (t[h0] != '@') return fail(ORC_FASTQ_MISSING_AT, h0);  // fastq.rs:272-274
        orc_fastq_rec_t r;
        memset(&r, 0, sizeof r);
        {  // fastq.rs:275-277: line[1..].trim_end().splitn(2, ' ')
            const uint64_t a = h0 + 1, n = trim_end(t + a, h1 - a);
            uint64_t sp = 0;
            while (sp < n && t[a + sp] != ' ') sp++;
            r.id_off = a;
            r.id_len = sp;
            if (sp < n) {
                r.desc_off = a + sp + 1;
                r.desc_len = n - sp - 1;
                r.has_desc = 1;
            }
        }
        pos = h1;
        r.seq_off = so;
        r.qual_off = qo;
        // fastq.rs:280-288
        uint64_t lines_read = 0;
        uint64_t l0 = pos, l1 = line_end(t, len, pos);
        if (l1 > l0 && !valid_utf8(t + l0, l1 - l0)) return fail(ORC_FASTQ_IO, l0);
        while (l1 > l0 && t[l0] != '+') {
            const uint64_t n = trim_end(t + l0, l1 - l0);
            if (seq) memcpy(seq + so, t + l0, n);
            so += n;
            lines_read++;
            pos = l1;
            l0 = pos;
            l1 = line_end(t, len, pos);
            if (l1 > l0 && !valid_utf8(t + l0, l1 - l0)) return fail(ORC_FASTQ_IO, l0);
        }
        pos = l1;  // the '+' line (or nothing at the end of the input)
        // fastq.rs:290-296
        for (uint64_t q = 0; q < lines_read; q++) {
            l0 = pos;
            l1 = line_end(t, len, pos);
            if (l1 > l0 && !valid_utf8(t + l0, l1 - l0)) return fail(ORC_FASTQ_IO, l0);
            const uint64_t n = trim_end(t + l0, l1 - l0);
            if (qual) memcpy(qual + qo, t + l0, n);
            qo += n;
            pos = l1;
        }
        r.seq_len = so - r.seq_off;
        r.qual_len = qo - r.qual_off;
        if (r.qual_len == 0) return fail(ORC_FASTQ_INCOMPLETE, h0);  // fastq.rs:298-300
        // Record::check, fastq.rs:388-410 (first failing rule)
        r.check = ORC_FQCHECK_OK;
        {
            bool seq_ascii = true, seq_ok = true, qual_ascii = true;
            if (seq)
                for (uint64_t i = 0; i < r.seq_len; i++) {
                    const uint8_t b = seq[r.seq_off + i];
                    if (b >= 0x80) seq_ascii = false;
                    if (!((b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '-' || b == '.' || b == '*')) seq_ok = false;
                }
            if (qual)
                for (uint64_t i = 0; i < r.qual_len; i++)
                    if (qual[r.qual_off + i] >= 0x80) qual_ascii = false;
            if (r.id_len == 0) r.check = ORC_FQCHECK_EMPTY_ID;
            else if (!seq_ascii) r.check = ORC_FQCHECK_NONASCII_SEQ;
            else if (!seq_ok) r.check = ORC_FQCHECK_INVALID_SEQ;
            else if (!qual_ascii) r.check = ORC_FQCHECK_NONASCII_QUAL;
            else if (r.seq_len != r.qual_len) r.check = ORC_FQCHECK_UNEQUAL;
        }
        if (nrec < rec_cap && recs) recs[nrec] = r;
        nrec++;
    }
    *n_records = nrec;
    return 0;
}

// bio_types::alignment::Alignment::cigar(hard_clip) — bio-types 1.0 (PARITY UNPINNED, see the file header).
// x is the query: soft/hard clips are xstart and xlen - xend; runs of Match '=', Subst 'X', Del 'D', Ins 'I';
// clip operations inside `operations` emit nothing; AlignmentMode::Custom is not supported there (panic).
// bio_types::alignment::Alignment::pretty(x, y, ncol) — bio-types 1.0 (PARITY UNPINNED: the crate is not in the
// reference tree and rust-bio only prints the result, e.g. pairwise/banded.rs:1805; restated from the crate's source).
// Returns the length written, -1 if cap is too small, -2 where the crate panics.
extern "C" int64_t orc_pretty(const orc_alignment_t* a, const uint64_t* ops, const uint8_t* x, uint64_t xl, const uint8_t* y,
                              uint64_t yl, uint64_t ncol, char* out, uint64_t cap) {
    std::string xp, ip, yp;
    bool panic = false;
    auto ch = [&](uint8_t c) -> std::string {  // format!("{}", String::from_utf8_lossy(&[c]))
        if (c >= 0x80) return "\xEF\xBF\xBD";
        return std::string(1, (char)c);
    };
    if (a->n_ops) {
        uint64_t xi = 0, yi = 0;
        if (a->mode != ORC_MODE_CUSTOM) {
            xi = a->xstart;
            yi = a->ystart;
            for (uint64_t k = 0; k < a->xstart && k < xl; k++) { xp += ch(x[k]); ip += ' '; yp += ' '; }
            for (uint64_t k = 0; k < a->ystart && k < yl; k++) { yp += ch(y[k]); ip += ' '; xp += ' '; }
        }
        for (uint64_t i = 0; i < a->n_ops && !panic; i++) {
            const uint64_t kind = ops[i] & 0xFF, len = ops[i] >> 8;
            switch (kind) {
                case ORC_OP_MATCH:
                case ORC_OP_SUBST:
                    if (xi >= xl || yi >= yl) { panic = true; break; }
                    xp += ch(x[xi++]); ip += kind == ORC_OP_MATCH ? '|' : '\\'; yp += ch(y[yi++]);
                    break;
                case ORC_OP_DEL:
                    if (yi >= yl) { panic = true; break; }
                    xp += '-'; ip += 'x'; yp += ch(y[yi++]);
                    break;
                case ORC_OP_INS:
                    if (xi >= xl) { panic = true; break; }
                    xp += ch(x[xi++]); ip += '+'; yp += '-';
                    break;
                case ORC_OP_XCLIP:  // `for k in x.iter().take(len)`: the first len symbols, wherever the clip sits
                    for (uint64_t k = 0; k < len && k < xl; k++) { xp += ch(x[k]); xi++; ip += ' '; yp += ' '; }
                    break;
                default:
                    for (uint64_t k = 0; k < len && k < yl; k++) { yp += ch(y[k]); yi++; ip += ' '; xp += ' '; }
                    break;
            }
        }
        if (a->mode != ORC_MODE_CUSTOM) {
            for (uint64_t k = xi; k < xl; k++) { xp += ch(x[k]); ip += ' '; yp += ' '; }
            for (uint64_t k = yi; k < yl; k++) { yp += ch(y[k]); ip += ' '; xp += ' '; }
        }
    }
    if (panic || xp.size() != ip.size() || yp.size() != ip.size()) return -2;  // assert_eq!(x_pretty.len(), inb_pretty.len())
    std::string s;
    for (uint64_t idx = 0; idx < xp.size(); idx += ncol) {
        const uint64_t e = std::min<uint64_t>(idx + ncol, xp.size());
        s += xp.substr(idx, e - idx) + "\n" + ip.substr(idx, e - idx) + "\n" + yp.substr(idx, e - idx) + "\n" + "\n\n";
    }
    if (s.size() > cap) return -1;
    memcpy(out, s.data(), s.size());
    return (int64_t)s.size();
}

extern "C" int64_t orc_cigar(const orc_alignment_t* a, const uint64_t* ops, int hard_clip, char* out, uint64_t cap) {
    if (a->mode == ORC_MODE_CUSTOM) return -2;
    std::string c;
    const char clip = hard_clip ? 'H' : 'S';
    auto add = [&](uint64_t kind, uint64_t k) {
        const char* s = kind == ORC_OP_MATCH ? "=" : kind == ORC_OP_SUBST ? "X" : kind == ORC_OP_DEL ? "D" : kind == ORC_OP_INS ? "I" : nullptr;
        if (s) c += std::to_string(k) + s;
    };
    if (a->n_ops) {
        uint64_t last = ops[0] & 0xFF, k = 1;
        if (a->xstart > 0) c += std::to_string(a->xstart) + clip;
        for (uint64_t i = 1; i < a->n_ops; i++) {
            const uint64_t op = ops[i] & 0xFF;
            if (op == last) {
                k++;
            } else {
                add(last, k);
                k = 1;
            }
            last = op;
        }
        add(last, k);
        if (a->xlen > a->xend) c += std::to_string(a->xlen - a->xend) + clip;
    }
    if (c.size() > cap) return -1;
    memcpy(out, c.data(), c.size());
    return (int64_t)c.size();
}
