//! `FMIndex::backward_search` / `Interval::occ` on the GPU (reference: src/data_structures/fmindex.rs:69-248,
//! suffix_array.rs:86-184).  `suffix_array`, `bwt`, `less` stay rust-bio's host code; `FMIndex::new(bwt, less, occ)`
//! (fmindex.rs:245) becomes `GpuFMIndex::new(ctx, bwt, less, k, alphabet)`: the reference's third argument, the
//! host-built `Occ` table, is NOT uploaded — the engine ranks on its own packed blocks built from `bwt`
//! (include/biogpu.h, bg_fm_build), so a caller may skip `Occ::new` altogether; `k` is only validated.
use crate::{concat, strerror, sys, Context};
use bio::alphabets::Alphabet;
use bio::data_structures::bwt::{Less, BWT};
use bio::data_structures::fmindex::{BackwardSearchResult, Interval};
use bio::data_structures::suffix_array::RawSuffixArray;

/// The handle keeps the pointer of the `Context` it was built with (its stream and staging serve the host-buffer entry
/// points below), so the borrow is part of the type: the index cannot outlive its context.
pub struct GpuFMIndex<'ctx> {
    pub(crate) h: *mut sys::bg_fm,
    _ctx: std::marker::PhantomData<&'ctx Context>,
}
// May move to another thread; NOT `Sync`: `backward_search_batch` / `occ_batch` go through the host-buffer entry
// points, which use the context's stream and pinned staging and share its one-thread-at-a-time rule
// (include/biogpu.h, "Streams and threads").  Threads that want to share one index wrap it in a `Mutex`, or call
// the `*_dev` entry points of biogpu-sys on their own streams (those use no context state).
unsafe impl Send for GpuFMIndex<'_> {}

impl<'ctx> GpuFMIndex<'ctx> {
    pub fn new(ctx: &'ctx Context, bwt: &BWT, less: &Less, occ_k: u32, alphabet: &Alphabet) -> Self {
        let less64: Vec<u64> = less.iter().map(|&v| v as u64).collect();
        let syms: Vec<u8> = alphabet.symbols.iter().map(|s| s as u8).collect();
        let mut h = std::ptr::null_mut();
        let rc = unsafe {
            sys::bg_fm_build(ctx.raw, bwt.as_ptr(), bwt.len() as u64, less64.as_ptr(), less64.len() as u32, occ_k,
                             syms.as_ptr(), syms.len() as u32, &mut h)
        };
        assert!(rc == 0, "{}", strerror(rc)); // BG_ERR_OUT_OF_ALPHABET == Occ::new's index panic (bwt.rs:114)
        GpuFMIndex { h, _ctx: std::marker::PhantomData }
    }

    /// `Serialize` (the reference derives it for `FMIndex`, `Occ`, `SampledSuffixArray`: fmindex.rs:214, bwt.rs:76,
    /// suffix_array.rs:124): BWT, less, alphabet, k, the attached suffix array and an owned text, in the engine's file format.
    pub fn save(&self, path: &std::path::Path) -> std::io::Result<()> {
        let c = std::ffi::CString::new(path.to_str().expect("path")).expect("path");
        let rc = unsafe { sys::bg_fm_save(self.h, c.as_ptr()) };
        if rc == 0 { Ok(()) } else { Err(std::io::Error::new(std::io::ErrorKind::Other, strerror(rc))) }
    }

    /// `Deserialize`: the loaded index answers every call like the saved one.
    pub fn load(ctx: &'ctx Context, path: &std::path::Path) -> std::io::Result<Self> {
        let c = std::ffi::CString::new(path.to_str().expect("path")).expect("path");
        let mut h = std::ptr::null_mut();
        let rc = unsafe { sys::bg_fm_load(ctx.raw, c.as_ptr(), &mut h) };
        if rc == 0 { Ok(GpuFMIndex { h, _ctx: std::marker::PhantomData }) } else { Err(std::io::Error::new(std::io::ErrorKind::Other, strerror(rc))) }
    }

    /// `backward_search` (fmindex.rs:144-208) for many patterns.
    pub fn backward_search_batch(&self, patterns: &[&[u8]]) -> Vec<BackwardSearchResult> {
        let (pat, off) = concat(patterns);
        let n = patterns.len();
        let (mut tag, mut lo, mut hi, mut ml) = (vec![0u8; n], vec![0u64; n], vec![0u64; n], vec![0u32; n]);
        let rc = unsafe {
            sys::bg_fm_backward_search_batch(self.h, n as u64, pat.as_ptr(), off.as_ptr(), tag.as_mut_ptr(), lo.as_mut_ptr(),
                                             hi.as_mut_ptr(), ml.as_mut_ptr())
        };
        // BG_ERR_OUT_OF_ALPHABET: some query reached a byte outside the alphabet — the reference panics with an
        // index out of bounds there (fmindex.rs:229, bwt.rs:158)
        assert!(rc == 0, "{}", strerror(rc));
        (0..n)
            .map(|q| {
                let iv = Interval { lower: lo[q] as usize, upper: hi[q] as usize };
                match tag[q] as i32 {
                    sys::BG_FM_COMPLETE => BackwardSearchResult::Complete(iv),
                    sys::BG_FM_PARTIAL => BackwardSearchResult::Partial(iv, ml[q] as usize),
                    _ => BackwardSearchResult::Absent,
                }
            })
            .collect()
    }

    pub fn backward_search<'b, P: Iterator<Item = &'b u8> + DoubleEndedIterator>(&self, pattern: P) -> BackwardSearchResult {
        let p: Vec<u8> = pattern.copied().collect();
        self.backward_search_batch(&[&p]).pop().unwrap()
    }

    /// attach the suffix array the index was built from (`RawSuffixArray`, suffix_array.rs:25)
    pub fn attach_sa(&mut self, sa: &RawSuffixArray) {
        let v: Vec<u64> = sa.iter().map(|&p| p as u64).collect();
        let rc = unsafe { sys::bg_fm_set_suffix_array(self.h, v.as_ptr(), v.len() as u64) };
        assert!(rc == 0, "{}", strerror(rc));
    }

    /// `Interval::occ` (fmindex.rs:75-79) for many intervals: positions of interval v are pos[off[v]..off[v+1]]
    pub fn occ_batch(&self, ivs: &[Interval]) -> (Vec<u64>, Vec<u64>) {
        let lo: Vec<u64> = ivs.iter().map(|i| i.lower as u64).collect();
        let hi: Vec<u64> = ivs.iter().map(|i| i.upper as u64).collect();
        let total: u64 = ivs.iter().map(|i| (i.upper - i.lower) as u64).sum();
        let (mut off, mut pos) = (vec![0u64; ivs.len() + 1], vec![0u64; total as usize]);
        let rc = unsafe {
            sys::bg_interval_occ_batch(self.h, ivs.len() as u64, lo.as_ptr(), hi.as_ptr(), off.as_mut_ptr(), pos.as_mut_ptr(), total)
        };
        assert!(rc == 0, "Interval out of range of suffix array"); // fmindex.rs:77
        (off, pos)
    }
}

impl Drop for GpuFMIndex<'_> {
    fn drop(&mut self) {
        unsafe { sys::bg_fm_free(self.h) };
    }
}

/// `FMDIndex::from(fmindex)` (fmindex.rs:311-329) over a `GpuFMIndex` built on `T$R$`: the reference asserts that the BWT is
/// a word over `dna::n_alphabet()` + `$`; here the BWT is the handle's own, read back out of its rank blocks
/// (`bg_fm_bwt`) — so an index that came from `GpuFMIndex::load` works like a deserialized one does in the reference.
/// `BiInterval` is `usize` throughout (fmindex.rs:254-259): the `*64` entry points carry uint64 records on any index
/// size — `T$R$` of a human genome is 6.2 G symbols, beyond `u32`.
pub struct GpuFMDIndex<'a, 'ctx> {
    fm: &'a GpuFMIndex<'ctx>,
}

impl<'a, 'ctx> GpuFMDIndex<'a, 'ctx> {
    pub fn from(fm: &'a GpuFMIndex<'ctx>) -> Self {
        let mut n = 0u64;
        assert!(unsafe { sys::bg_fm_len(fm.h, &mut n) } == 0);
        let mut bwt = vec![0u8; n as usize];
        let rc = unsafe { sys::bg_fm_bwt(fm.h, bwt.as_mut_ptr()) };
        assert!(rc == 0, "{}", strerror(rc));
        let alphabet = bio::alphabets::dna::n_alphabet();
        assert!(bwt.iter().all(|&c| c == b'$' || alphabet.is_word(&[c])),
                "Expecting BWT over the DNA alphabet (including N) with the sentinel $.");
        GpuFMDIndex { fm }
    }

    /// `all_smems(pattern, l)` (fmindex.rs:479-501) for many patterns: (interval, position on the pattern, length) triples
    /// in the order the reference pushes them.
    pub fn all_smems_batch(&self, patterns: &[&[u8]], min_len: usize) -> Vec<Vec<(bio::data_structures::fmindex::BiInterval, usize, usize)>> {
        self.run(patterns, None, min_len)
    }

    /// `smems(pattern, i, l)` (fmindex.rs:363-434) for many (pattern, i) pairs.
    pub fn smems_batch(&self, patterns: &[&[u8]], positions: &[u32], min_len: usize) -> Vec<Vec<(bio::data_structures::fmindex::BiInterval, usize, usize)>> {
        self.run(patterns, Some(positions), min_len)
    }

    fn run(&self, patterns: &[&[u8]], positions: Option<&[u32]>, min_len: usize) -> Vec<Vec<(bio::data_structures::fmindex::BiInterval, usize, usize)>> {
        let (pat, off) = concat(patterns);
        let n = patterns.len();
        let cap = patterns.iter().map(|p| p.len()).max().unwrap_or(0) + 1;
        let (mut count, mut out) = (vec![0u32; n], vec![0u64; n * cap * 6]);
        let rc = unsafe {
            sys::bg_fmd_smems_batch64(self.fm.h, positions.is_none() as i32, n as u64, pat.as_ptr(), off.as_ptr(),
                                      positions.map_or(std::ptr::null(), |p| p.as_ptr()), min_len as u32, cap as u32,
                                      count.as_mut_ptr(), out.as_mut_ptr())
        };
        assert!(rc == 0, "{}", strerror(rc)); // BG_ERR_OUT_OF_ALPHABET: the reference's index panic (fmindex.rs:229)
        (0..n).map(|q| (0..count[q] as usize).map(|t| {
            let r = &out[(q * cap + t) * 6..];
            (bio::data_structures::fmindex::BiInterval { lower: r[0] as usize, lower_rev: r[1] as usize, size: r[2] as usize, match_size: r[3] as usize },
             r[4] as usize, r[5] as usize)
        }).collect()).collect()
    }
}
