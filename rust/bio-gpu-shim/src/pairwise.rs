//! `bio::alignment::pairwise::Aligner` on the GPU (reference: src/alignment/pairwise/mod.rs:472-1016).
use crate::{concat, mode_to_c, strerror, sys, to_alignment, zero_alignment, Context};
use bio::alignment::pairwise::{MatchFunc, Scoring};
use bio::utils::TextSlice;
use bio_types::alignment::{Alignment, AlignmentMode};

/// A closure cannot cross the FFI: `F(a, b)` over all 65 536 byte pairs, once per aligner.
/// `MatchParams` takes the two-integer fast path (`match_scores` is `Some` only through `Scoring::from_scores`,
/// mod.rs:272), so no table is built for it.
pub(crate) fn tabulate<F: MatchFunc>(scoring: &Scoring<F>) -> Option<Vec<i32>> {
    if scoring.match_scores.is_some() {
        return None;
    }
    let mut t = vec![0i32; 65536];
    for a in 0..256usize {
        for b in 0..256usize {
            t[a * 256 + b] = scoring.match_fn.score(a as u8, b as u8);
        }
    }
    Some(t)
}

pub(crate) fn scoring_to_c<F: MatchFunc>(s: &Scoring<F>, table: &Option<Vec<i32>>) -> sys::bg_scoring_t {
    sys::bg_scoring_t {
        gap_open: s.gap_open,
        gap_extend: s.gap_extend,
        xclip_prefix: s.xclip_prefix,
        xclip_suffix: s.xclip_suffix,
        yclip_prefix: s.yclip_prefix,
        yclip_suffix: s.yclip_suffix,
        match_score: s.match_scores.map_or(0, |m| m.0),
        mismatch_score: s.match_scores.map_or(0, |m| m.1),
        match_scores_some: s.match_scores.is_some() as i32,
        matrix: table.as_ref().map_or(std::ptr::null(), |t| t.as_ptr()),
    }
}

pub struct Aligner<F: MatchFunc> {
    scoring: Scoring<F>,
    table: Option<Vec<i32>>,
    ctx: Context,
}

impl<F: MatchFunc> Aligner<F> {
    /// mod.rs:495-503
    pub fn new(gap_open: i32, gap_extend: i32, match_fn: F) -> Self {
        Self::with_scoring(Scoring::new(gap_open, gap_extend, match_fn))
    }
    /// mod.rs:516-530 — capacities are the engine's business (persistent device scratch in the context)
    pub fn with_capacity(_m: usize, _n: usize, gap_open: i32, gap_extend: i32, match_fn: F) -> Self {
        Self::new(gap_open, gap_extend, match_fn)
    }
    /// mod.rs:537-543, with the asserts of with_capacity_and_scoring (554-571)
    pub fn with_scoring(scoring: Scoring<F>) -> Self {
        assert!(scoring.gap_open <= 0, "gap_open can't be positive");
        assert!(scoring.gap_extend <= 0, "gap_extend can't be positive");
        assert!(scoring.xclip_prefix <= 0, "Clipping penalty (x prefix) can't be positive");
        assert!(scoring.xclip_suffix <= 0, "Clipping penalty (x suffix) can't be positive");
        assert!(scoring.yclip_prefix <= 0, "Clipping penalty (y prefix) can't be positive");
        assert!(scoring.yclip_suffix <= 0, "Clipping penalty (y suffix) can't be positive");
        let table = tabulate(&scoring);
        Aligner { scoring, table, ctx: Context::new(0) }
    }
    pub fn with_capacity_and_scoring(_m: usize, _n: usize, scoring: Scoring<F>) -> Self {
        Self::with_scoring(scoring)
    }

    /// New: n independent pairs in one call (`bg_align_batch`).
    pub fn align_batch(&mut self, mode: AlignmentMode, xs: &[&[u8]], ys: &[&[u8]]) -> Vec<Alignment> {
        assert_eq!(xs.len(), ys.len());
        let (x, x_off) = concat(xs);
        let (y, y_off) = concat(ys);
        let sc = scoring_to_c(&self.scoring, &self.table);
        let mut out = vec![zero_alignment(); xs.len()];
        let mut ops = vec![0u8; x.len() + y.len() + 4 * xs.len() + 8];
        let mut used = 0u64;
        let rc = unsafe {
            sys::bg_align_batch(self.ctx.raw, &sc, mode_to_c(mode), xs.len() as u64, x.as_ptr(), x_off.as_ptr(),
                                y.as_ptr(), y_off.as_ptr(), out.as_mut_ptr(), ops.as_mut_ptr(), ops.len() as u64, &mut used)
        };
        assert!(rc == 0, "{}", strerror(rc)); // BG_ERR_POSITIVE_PENALTY == the reference's asserts
        out.iter().map(|r| to_alignment(r, &ops)).collect()
    }

    /// New: the batch sharded over the ranks of `comm` (north_star: "query batches shard embarrassingly across the 8
    /// GPUs ... a single RCCL all-gather ... only to collect per-query scores"): this rank aligns its slice (returned as
    /// full `Alignment`s, operations included) and every rank receives `{score, xstart, xend, ystart, yend}` of ALL pairs.
    pub fn align_batch_sharded(&mut self, comm: &crate::Comm, mode: AlignmentMode, xs: &[&[u8]], ys: &[&[u8]])
                               -> (Vec<Alignment>, Vec<[i32; 5]>) {
        let (lo, hi) = crate::shard_range(xs.len() as u64, comm.rank, comm.world);
        let mine = self.align_batch(mode, &xs[lo as usize..hi as usize], &ys[lo as usize..hi as usize]);
        let mut local = Vec::with_capacity(mine.len() * 20);
        for a in &mine {
            let c = |v: usize| i32::try_from(v).expect("coordinate");
            for v in [a.score, c(a.xstart), c(a.xend), c(a.ystart), c(a.yend)] {
                local.extend_from_slice(&v.to_ne_bytes());
            }
        }
        let all = comm.gather_records_host(&local, 20, xs.len() as u64);
        let recs = all.chunks_exact(20)
            .map(|c| {
                let mut r = [0i32; 5];
                for (k, w) in c.chunks_exact(4).enumerate() {
                    r[k] = i32::from_ne_bytes([w[0], w[1], w[2], w[3]]);
                }
                r
            })
            .collect();
        (mine, recs)
    }

    /// mod.rs:591
    pub fn custom(&mut self, x: TextSlice<'_>, y: TextSlice<'_>) -> Alignment {
        self.align_batch(AlignmentMode::Custom, &[x], &[y]).pop().unwrap()
    }
    /// mod.rs:925
    pub fn global(&mut self, x: TextSlice<'_>, y: TextSlice<'_>) -> Alignment {
        self.align_batch(AlignmentMode::Global, &[x], &[y]).pop().unwrap()
    }
    /// mod.rs:954
    pub fn semiglobal(&mut self, x: TextSlice<'_>, y: TextSlice<'_>) -> Alignment {
        self.align_batch(AlignmentMode::Semiglobal, &[x], &[y]).pop().unwrap()
    }
    /// mod.rs:986
    pub fn local(&mut self, x: TextSlice<'_>, y: TextSlice<'_>) -> Alignment {
        self.align_batch(AlignmentMode::Local, &[x], &[y]).pop().unwrap()
    }
}
