//! Drop-in delegation of rust-bio's hot path to the MI355X engine (libbiogpu.so, include/biogpu.h).
//!
//! * [`pairwise::Aligner`]  — `bio::alignment::pairwise::Aligner` (src/alignment/pairwise/mod.rs:472-1016)
//! * [`banded::Aligner`]    — `bio::alignment::pairwise::banded::Aligner` (banded.rs:122-1004)
//! * [`fmindex::GpuFMIndex`] — `FMIndex::backward_search`, `Interval::occ` (fmindex.rs:75-79,144-208)
//! * [`mapper`]             — the seed-and-extend composition (src/lib.rs:129-165) in one call
//!
//! Same method names, argument meaning and panics as the reference; every single-pair method is a batch of one
//! of the new `*_batch` siblings.  `Scoring`, `MatchFunc`, the alphabet and the BWT / Less / suffix-array builders
//! stay rust-bio's own host code (north_star).  No rustc exists in the environment this repository is built in:
//! the crate is source only; the same entry points are exercised through ctypes (rust-bio_amd/_lib.py) and the
//! C++ mirror (include/biogpu.hpp), and tests/test_shim_matches_header.py pins biogpu-sys to the header.
pub mod banded;
pub mod fmindex;
pub mod mapper;
pub mod pairwise;

use biogpu_sys as sys;
use bio_types::alignment::{Alignment, AlignmentMode, AlignmentOperation};
use std::ffi::CStr;

/// One engine context per device; not `Sync` (like the `&mut self` workspace of `Aligner`, mod.rs:472-481).
pub struct Context {
    pub(crate) raw: *mut sys::bg_ctx,
}

impl Context {
    pub fn new(device: i32) -> Self {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::bg_init(device, &mut raw) };
        assert!(rc == 0, "bg_init: {}", strerror(rc)); // no CPU fallback: without a gfx950 device this fails
        Context { raw }
    }
}

impl Context {
    /// Pairs of the last banded call on this context that the packed 16-bit fill handed to the int32 kernels
    /// (`bg_band_redo_pairs`): a statistic — the alignments are the same either way.
    pub fn band_redo_pairs(&self) -> u64 {
        let mut n = 0u64;
        let rc = unsafe { sys::bg_band_redo_pairs(self.raw, &mut n) };
        assert!(rc == 0, "bg_band_redo_pairs: {}", strerror(rc));
        n
    }
}

/// `bytes` as the engine's 2-bit stream (16 symbols per little-endian dword, `codes[c]` = the byte of code c): the host-side
/// twin of `bg_pack2_dev`.  `None` if a byte is none of the four codes (such input takes the byte entry points).
pub fn pack2(bytes: &[u8], codes: &[u8; 4]) -> Option<Vec<u32>> {
    let mut out = vec![0u32; (bytes.len() + 15) / 16];
    match unsafe { sys::bg_pack2_host(bytes.as_ptr(), bytes.len() as u64, codes.as_ptr(), out.as_mut_ptr()) } {
        1 => Some(out),
        0 => None,
        rc => panic!("bg_pack2_host: {}", rc),
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::bg_free(self.raw) };
    }
}

/// Several GPUs (include/biogpu.h "several GPUs", csrc/comm.hip): one process and one [`Context`] per device, the batch
/// cut with [`shard_range`], and the one collective of a sharded batch — the all-gather of fixed-size result records —
/// over RCCL ([`Comm::rccl`]; the id comes from [`Comm::unique_id`] on one rank and travels however the job talks) or
/// through shared memory for ranks RCCL cannot serve ([`Comm::host`]).
pub struct Comm {
    raw: *mut sys::bg_comm,
    pub rank: i32,
    pub world: i32,
}

/// rank's contiguous slice `[rank * n / world, (rank + 1) * n / world)`
pub fn shard_range(n_units: u64, rank: i32, world: i32) -> (u64, u64) {
    let (mut lo, mut hi) = (0u64, 0u64);
    let rc = unsafe { sys::bg_shard_range(n_units, rank, world, &mut lo, &mut hi) };
    assert!(rc == 0, "bg_shard_range: {}", strerror(rc));
    (lo, hi)
}

impl Comm {
    pub fn unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        let rc = unsafe { sys::bg_comm_unique_id(id.as_mut_ptr()) };
        assert!(rc == 0, "bg_comm_unique_id: {}", strerror(rc));
        id
    }
    pub fn rccl(ctx: &Context, rank: i32, world: i32, id: &[u8; 128]) -> Self {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::bg_comm_init(ctx.raw, rank, world, id.as_ptr(), &mut raw) };
        assert!(rc == 0, "bg_comm_init: {}", strerror(rc));
        Comm { raw, rank, world }
    }
    pub fn host(ctx: Option<&Context>, rank: i32, world: i32, name: &str) -> Self {
        let mut raw = std::ptr::null_mut();
        let cname = std::ffi::CString::new(name).unwrap();
        let rc = unsafe {
            sys::bg_comm_init_host(ctx.map_or(std::ptr::null_mut(), |c| c.raw), rank, world, cname.as_ptr(), &mut raw)
        };
        assert!(rc == 0, "bg_comm_init_host: {}", strerror(rc));
        Comm { raw, rank, world }
    }
    /// every rank's `local` records (`rec_bytes` each, host memory) on every rank, in rank order
    pub fn gather_records_host(&self, local: &[u8], rec_bytes: u32, total_records: u64) -> Vec<u8> {
        let mut all = vec![0u8; (total_records as usize) * rec_bytes as usize];
        let rc = unsafe {
            sys::bg_gather_records_host(self.raw, local.as_ptr() as *const _, (local.len() / rec_bytes as usize) as u64, rec_bytes,
                                        all.as_mut_ptr() as *mut _, total_records, std::ptr::null_mut())
        };
        assert!(rc == 0, "bg_gather_records_host: {}", strerror(rc));
        all
    }
}

impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { sys::bg_comm_free(self.raw) };
    }
}

pub(crate) fn strerror(rc: i32) -> String {
    unsafe { CStr::from_ptr(sys::bg_strerror(rc)) }.to_string_lossy().into_owned()
}

/// `AlignmentMode` -> `BG_MODE_*`, explicitly (bio-types' discriminant order is not part of its API).
pub(crate) fn mode_to_c(mode: AlignmentMode) -> i32 {
    match mode {
        AlignmentMode::Custom => sys::BG_MODE_CUSTOM,
        AlignmentMode::Global => sys::BG_MODE_GLOBAL,
        AlignmentMode::Semiglobal => sys::BG_MODE_SEMIGLOBAL,
        AlignmentMode::Local => sys::BG_MODE_LOCAL,
    }
}

pub(crate) fn mode_from_c(mode: u8) -> AlignmentMode {
    match mode as i32 {
        sys::BG_MODE_GLOBAL => AlignmentMode::Global,
        sys::BG_MODE_SEMIGLOBAL => AlignmentMode::Semiglobal,
        sys::BG_MODE_LOCAL => AlignmentMode::Local,
        _ => AlignmentMode::Custom,
    }
}

/// concatenated sequences + n + 1 offsets, the layout every batched entry point takes
pub(crate) fn concat(seqs: &[&[u8]]) -> (Vec<u8>, Vec<u64>) {
    let mut buf = Vec::with_capacity(seqs.iter().map(|s| s.len()).sum());
    let mut off = Vec::with_capacity(seqs.len() + 1);
    off.push(0u64);
    for s in seqs {
        buf.extend_from_slice(s);
        off.push(buf.len() as u64);
    }
    (buf, off)
}

/// `bg_alignment_t` + operation bytes -> `bio_types::alignment::Alignment` (fields as constructed at mod.rs:911-921)
pub(crate) fn to_alignment(r: &sys::bg_alignment_t, ops: &[u8]) -> Alignment {
    let mut clip = r.clip_len.iter();
    let lo = r.ops_off as usize;
    let operations = ops[lo..lo + r.n_ops as usize]
        .iter()
        .map(|&o| match o as i32 {
            sys::BG_OP_MATCH => AlignmentOperation::Match,
            sys::BG_OP_SUBST => AlignmentOperation::Subst,
            sys::BG_OP_DEL => AlignmentOperation::Del,
            sys::BG_OP_INS => AlignmentOperation::Ins,
            sys::BG_OP_XCLIP => AlignmentOperation::Xclip(*clip.next().unwrap() as usize),
            _ => AlignmentOperation::Yclip(*clip.next().unwrap() as usize),
        })
        .collect();
    Alignment {
        score: r.score,
        ystart: r.ystart as usize,
        xstart: r.xstart as usize,
        yend: r.yend as usize,
        xend: r.xend as usize,
        ylen: r.ylen as usize,
        xlen: r.xlen as usize,
        operations,
        mode: mode_from_c(r.mode),
    }
}

pub(crate) fn zero_alignment() -> sys::bg_alignment_t {
    sys::bg_alignment_t {
        score: 0, xstart: 0, xend: 0, ystart: 0, yend: 0, xlen: 0, ylen: 0, n_ops: 0, ops_off: 0,
        clip_len: [0; 4], n_clips: 0, mode: 0, status: 0, _pad: 0, _reserved: 0,
    }
}
