//! `bio::alignment::pairwise::banded::Aligner` on the GPU (reference: src/alignment/pairwise/banded.rs:122-1004).
use crate::pairwise::{scoring_to_c, tabulate};
use crate::{concat, mode_to_c, strerror, sys, to_alignment, zero_alignment, Context};
use bio::alignment::pairwise::{MatchFunc, Scoring};
use bio::utils::TextSlice;
use bio_types::alignment::{Alignment, AlignmentMode};

pub struct Aligner<F: MatchFunc> {
    scoring: Scoring<F>,
    table: Option<Vec<i32>>,
    k: usize,
    w: usize,
    ctx: Context,
}

impl<F: MatchFunc> Aligner<F> {
    /// banded.rs:150 — `k`: k-mer length, `w`: window
    pub fn new(gap_open: i32, gap_extend: i32, match_fn: F, k: usize, w: usize) -> Self {
        Self::with_scoring(Scoring::new(gap_open, gap_extend, match_fn), k, w)
    }
    /// banded.rs:259 (asserts of 215-232)
    pub fn with_scoring(scoring: Scoring<F>, k: usize, w: usize) -> Self {
        assert!(scoring.gap_open <= 0, "gap_open can't be positive");
        assert!(scoring.gap_extend <= 0, "gap_extend can't be positive");
        assert!(scoring.xclip_prefix <= 0 && scoring.xclip_suffix <= 0, "Clipping penalty can't be positive");
        assert!(scoring.yclip_prefix <= 0 && scoring.yclip_suffix <= 0, "Clipping penalty can't be positive");
        let table = tabulate(&scoring);
        Aligner { scoring, table, k, w, ctx: Context::new(0) }
    }
    /// banded.rs:272
    pub fn get_mut_scoring(&mut self) -> &mut Scoring<F> {
        self.table = None; // re-tabulated on the next call
        &mut self.scoring
    }

    /// New: n pairs in one call; the band (k-mer matches, sparse DP chain, `Band::create`, banded.rs:1278-1367) is
    /// built on the device.  A band above MAX_CELLS yields the reference's sentinel alignment (banded.rs:407-420).
    pub fn align_batch(&mut self, mode: AlignmentMode, xs: &[&[u8]], ys: &[&[u8]]) -> Vec<Alignment> {
        if self.table.is_none() {
            self.table = tabulate(&self.scoring);
        }
        let (x, x_off) = concat(xs);
        let (y, y_off) = concat(ys);
        let sc = scoring_to_c(&self.scoring, &self.table);
        let mut out = vec![zero_alignment(); xs.len()];
        let mut ops = vec![0u8; x.len() + y.len() + 4 * xs.len() + 8];
        let mut used = 0u64;
        let rc = unsafe {
            sys::bg_align_banded_batch(self.ctx.raw, &sc, mode_to_c(mode), self.k as u32, self.w as u32, xs.len() as u64,
                                       x.as_ptr(), x_off.as_ptr(), y.as_ptr(), y_off.as_ptr(), out.as_mut_ptr(),
                                       ops.as_mut_ptr(), ops.len() as u64, &mut used, std::ptr::null_mut())
        };
        assert!(rc == 0, "{}", strerror(rc));
        out.iter().map(|r| to_alignment(r, &ops)).collect()
    }

    /// compute_alignment over a band the caller's matches define — the common tail of custom_with_matches /
    /// custom_with_match_path / custom_with_expanded_matches / *_with_prehash (banded.rs:294-401, 938-970)
    pub fn custom_with_matches(&mut self, x: TextSlice<'_>, y: TextSlice<'_>, matches: &[(u32, u32)]) -> Alignment {
        self.with_band(AlignmentMode::Custom, x, y, matches, None)
    }
    pub fn custom_with_match_path(&mut self, x: TextSlice<'_>, y: TextSlice<'_>, matches: &[(u32, u32)], path: &[usize]) -> Alignment {
        self.with_band(AlignmentMode::Custom, x, y, matches, Some(path))
    }

    fn with_band(&mut self, mode: AlignmentMode, x: &[u8], y: &[u8], matches: &[(u32, u32)], path: Option<&[usize]>) -> Alignment {
        if self.table.is_none() {
            self.table = tabulate(&self.scoring);
        }
        let sc = scoring_to_c(&self.scoring, &self.table);
        let (x_off, y_off) = ([0u64, x.len() as u64], [0u64, y.len() as u64]);
        let xy: Vec<u32> = matches.iter().flat_map(|&(a, b)| [a, b]).collect();
        let m_off = [0u64, matches.len() as u64];
        let p32: Option<Vec<u32>> = path.map(|p| p.iter().map(|&v| v as u32).collect());
        let p_off = [0u64, path.map_or(0, |p| p.len()) as u64];
        let band_off = [0u64, y.len() as u64 + 1];
        let (mut start, mut end) = (vec![0u32; y.len() + 1], vec![0u32; y.len() + 1]);
        let rc = unsafe {
            sys::bg_band_from_matches_batch(&sc, mode_to_c(mode), self.k as u32, self.w as u32, 1, x_off.as_ptr(), y_off.as_ptr(),
                                            xy.as_ptr(), m_off.as_ptr(), p32.as_ref().map_or(std::ptr::null(), |p| p.as_ptr()),
                                            if p32.is_some() { p_off.as_ptr() } else { std::ptr::null() }, band_off.as_ptr(),
                                            start.as_mut_ptr(), end.as_mut_ptr(), std::ptr::null_mut())
        };
        assert!(rc == 0, "incoming matches must be sorted"); // sparse.rs:212-217
        let mut out = [zero_alignment()];
        let mut ops = vec![0u8; x.len() + y.len() + 12];
        let mut used = 0u64;
        let rc = unsafe {
            sys::bg_align_banded_bands_batch(self.ctx.raw, &sc, mode_to_c(mode), 1, x.as_ptr(), x_off.as_ptr(), y.as_ptr(),
                                             y_off.as_ptr(), band_off.as_ptr(), start.as_ptr(), end.as_ptr(), out.as_mut_ptr(),
                                             ops.as_mut_ptr(), ops.len() as u64, &mut used, std::ptr::null_mut())
        };
        assert!(rc == 0, "{}", strerror(rc));
        to_alignment(&out[0], &ops)
    }

    /// banded.rs:282
    pub fn custom(&mut self, x: TextSlice<'_>, y: TextSlice<'_>) -> Alignment {
        self.align_batch(AlignmentMode::Custom, &[x], &[y]).pop().unwrap()
    }
    /// banded.rs:872
    pub fn global(&mut self, x: TextSlice<'_>, y: TextSlice<'_>) -> Alignment {
        self.align_batch(AlignmentMode::Global, &[x], &[y]).pop().unwrap()
    }
    /// banded.rs:901
    pub fn semiglobal(&mut self, x: TextSlice<'_>, y: TextSlice<'_>) -> Alignment {
        self.align_batch(AlignmentMode::Semiglobal, &[x], &[y]).pop().unwrap()
    }
    /// banded.rs:972
    pub fn local(&mut self, x: TextSlice<'_>, y: TextSlice<'_>) -> Alignment {
        self.align_batch(AlignmentMode::Local, &[x], &[y]).pop().unwrap()
    }
}
