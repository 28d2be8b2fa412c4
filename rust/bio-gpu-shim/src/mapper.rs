//! Seed-and-extend in one call (`bg_seed_extend_batch`): the composition rust-bio's callers write by hand from
//! `backward_search`, `Interval::occ` and `Aligner::semiglobal` (src/lib.rs:129-165, benches/fmindex.rs:20-38).
use crate::fmindex::GpuFMIndex;
use crate::pairwise::{scoring_to_c, tabulate};
use crate::{concat, strerror, sys, to_alignment, zero_alignment};
use bio::alignment::pairwise::{MatchFunc, Scoring};
use bio_types::alignment::Alignment;

pub struct Hit {
    /// `Aligner::semiglobal(read, window)` of the best candidate; `None`: no seed voted
    pub alignment: Option<Alignment>,
    pub ref_start: usize,
    pub ref_end: usize,
    pub n_candidates: u32,
}

impl GpuFMIndex<'_> {
    /// the text the index was built from, final sentinel included (windows are cut from it)
    pub fn attach_text(&mut self, text: &[u8]) {
        let rc = unsafe { sys::bg_fm_set_text(self.h, text.as_ptr(), text.len() as u64) };
        assert!(rc == 0, "{}", strerror(rc));
    }

    pub fn seed_extend_batch<F: MatchFunc>(&self, scoring: &Scoring<F>, reads: &[&[u8]], seed_len: u32, stride: u32,
                                           max_occ: u32, pad: u32) -> Vec<Hit> {
        let table = tabulate(scoring);
        let sc = scoring_to_c(scoring, &table);
        let prm = sys::bg_seed_params_t { seed_len, stride, max_occ, pad };
        let (buf, off) = concat(reads);
        let zero = sys::bg_seed_hit_t { aln: zero_alignment(), window_start: 0, ref_start: 0, ref_end: 0, n_candidates: 0, n_seed_hits: 0 };
        let mut hits = vec![zero; reads.len()];
        let mut ops = vec![0u8; 2 * buf.len() + (2 * pad as usize + 4) * reads.len() + 8];
        let mut used = 0u64;
        let rc = unsafe {
            sys::bg_seed_extend_batch(self.h, &sc, &prm, reads.len() as u64, buf.as_ptr(), off.as_ptr(), hits.as_mut_ptr(),
                                      ops.as_mut_ptr(), ops.len() as u64, &mut used)
        };
        assert!(rc == 0, "{}", strerror(rc));
        hits.iter()
            .map(|h| Hit {
                alignment: if h.aln.score == sys::BG_MIN_SCORE { None } else { Some(to_alignment(&h.aln, &ops)) },
                ref_start: h.ref_start as usize,
                ref_end: h.ref_end as usize,
                n_candidates: h.n_candidates,
            })
            .collect()
    }
}
